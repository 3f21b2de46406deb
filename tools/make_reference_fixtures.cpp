// PIN KIT -- run by a maintainer who HAS the reference's SeqAn fork (JensUweUlrich/seqan, branch "SeqAn") + sdsl-lite v2.1.1
// (src/seqan/CMakeLists.txt.in:20-30); neither exists in this repository's build environment, so this file is never
// compiled here.  It calls only what the reference calls -- TIbf(bins,h,k,bits) IBFBuild.cpp:465, insertKmer :190, store :505,
// count on both strands IBFClassify.cpp:149-150 -- and writes test.ibf, test1.ibf (the fixtures missing from the reference
// checkout) and reference_counts.json.  Drop all three into tests/golden/reference_data/: tests/test_oracle_kat.py then
// pins the hash function, the .ibf layout and the reverse-complement-of-N rule.  See tools/README.md for the command line.
#include <seqan/binning_directory.h>
#include <seqan/modifier.h>
#include <seqan/seq_io.h>

#include <cmath>
#include <cstdio>
#include <fstream>
#include <stdexcept>
#include <string>
#include <vector>
using namespace seqan;
typedef BinningDirectory<InterleavedBloomFilter, BDConfig<Dna5, Normal, Uncompressed>> TIbf;   // src/IBF/IBF.hpp:92-94
typedef ModifiedString<ModifiedString<Dna5String, ModComplementDna>, ModReverse> TSeqRevComp;  // src/IBF/IBF.hpp:96-97
static const uint64_t F = 100000, K = 13, H = 3;  // IBFConfig defaults, src/IBF/IBFConfig.hpp:52-82
static const double FP = 0.01;

static std::string cut(const std::string &seq)  // cutOutNNNs + concatenation, IBFBuild.cpp:81-88,112-132
{
    std::string out;
    size_t start = 0, end = 0, len = seq.size();
    while ((start = seq.find_first_not_of("N", end)) != std::string::npos) {
        end = seq.find("N", start);
        if (end > len) { out += seq.substr(start, len - start - 1); break; }
        out += seq.substr(start, end - start);
    }
    return out;
}

static void build(TIbf &filter, const char *fasta, const std::string &out_path)  // IBF::create_filter, IBFBuild.cpp:421-521
{
    SeqFileIn in;
    if (!open(in, fasta)) throw std::runtime_error(std::string("cannot open ") + fasta);
    StringSet<CharString> ids, seqs;
    readRecords(ids, seqs, in);
    std::vector<std::string> clean;
    uint64_t bins = 0;
    for (unsigned i = 0; i < length(seqs); ++i) {
        if (length(seqs[i]) < K) continue;                 // IBFBuild.cpp:70-74
        clean.push_back(cut(toCString(seqs[i])));
        bins += clean.back().size() / F + 1;               // :90
    }
    const uint64_t columns = (uint64_t)std::floor((double)bins / 64.0 + 1) * 64;  // calculate_filter_size_bits, :404-413
    const uint64_t per_bin = (uint64_t)std::ceil(-1 / (std::pow(1 - std::pow(FP, 1.0 / (double)H), 1.0 / ((double)(H * (F - K + 1)))) - 1));
    filter = TIbf(bins, H, K, per_bin * columns);          // :465
    uint64_t binid = 0;
    for (const std::string &s : clean) {                   // fragment loop, :165-204
        Dna5String seq = s;
        int64_t len = (int64_t)length(seq), idx = 0, fs = 0;
        while (fs < len - 1) {
            const uint64_t fe = std::min<uint64_t>((uint64_t)(idx + 1) * F, (uint64_t)len);
            Infix<Dna5String>::Type fragment = infix(seq, fs, fe);
            insertKmer(filter, fragment, binid++);         // :190
            ++idx;
            fs = idx * (int64_t)F - (int64_t)K + 1;
        }
    }
    store(filter, toCString(out_path));                    // :505
}

static void dump(std::ostream &o, const char *key, const std::vector<uint16_t> &v)
{
    o << "\"" << key << "\": [";
    for (size_t i = 0; i < v.size(); ++i) o << (i ? ", " : "") << v[i];
    o << "]";
}

int main(int argc, char **argv)
{
    if (argc != 4) { std::fprintf(stderr, "usage: %s libIBFTests/data/test.fasta libIBFTests/data/test1.fasta out_dir\n", argv[0]); return 2; }
    const std::string dir = argv[3], names[2] = {"test.ibf", "test1.ibf"};
    TIbf filters[2];
    for (int i = 0; i < 2; ++i) build(filters[i], argv[1 + i], dir + "/" + names[i]);
    std::string r354;  // src/test/libIBFTests/read.hpp:22
    for (int i = 0; i < 6; ++i) r354 += "AAAAAAAACCCCCCCCCGAGAGAGGAGAGAGGAGAGAGAGAGCCCCAAAAGAGAGGAGA";
    std::string n_fwd = r354.substr(0, 120);
    n_fwd[60] = 'N';  // an N on the strand that is hashed as it stands (Dna5 ordinal 4)
    // reverse complement of bases 30..71 of test.fasta's N-free sequence, with the A that mirrors a T of the reference
    // replaced by N: 30 shared 13-mers on the reverse strand if that strand sees T there (ModComplementDna), 18 if it sees N
    const std::pair<const char *, std::string> reads[4] = {{"mer35", "AAAAAAACCCCCCCCCGAGAGAGGAGAGAGGAGAG"}, {"read354", r354}, {"n_forward", n_fwd},
                                                            {"n_reverse", "ATAATATATAANATCTCCTCTCTTTTGGGGCTCTCTCTCTCC"}};
    std::ofstream js(dir + "/reference_counts.json");
    js << "{\"kmer_size\": " << K << ", \"hash_functions\": " << H << ", \"reads\": [\n";
    for (int r = 0; r < 4; ++r) {
        Dna5String seq = reads[r].second;  // (seqan::Dna5String) conversion, src/main/classify.hpp:272
        js << " {\"name\": \"" << reads[r].first << "\", \"seq\": \"" << reads[r].second << "\"";
        for (int i = 0; i < 2; ++i) {
            js << ",\n  \"" << names[i] << "\": {\"bins\": " << getNumberOfBins(filters[i]) << ", ";
            dump(js, "fwd", count(filters[i], seq));               // IBFClassify.cpp:149
            js << ", ";
            dump(js, "rev", count(filters[i], TSeqRevComp(seq)));  // IBFClassify.cpp:150
            js << "}";
        }
        js << "}" << (r < 3 ? ",\n" : "\n");
    }
    js << "]}\n";
    return 0;
}

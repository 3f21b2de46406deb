// test_mirror.cpp -- the reference's libIBFTests (src/test/libIBFTests/read.hpp, createfilter.hpp) re-expressed
// against the C++ mirror include/readbouncer_amd.hpp.  Same objects, same calls, same expected values; the filters
// test.ibf / test1.ibf are built from the reference's own FASTA fixtures instead of being loaded from its
// (missing) binary fixtures.  Needs a GPU.  usage: test_mirror <dir with libIBFTests_test.fasta, libIBFTests_test1.fasta> <tmpdir>
#include <cstdio>
#include <filesystem>
#include <iostream>
#include <thread>
#include <vector>

#include "../../include/readbouncer_amd.hpp"
#include "../../readbouncer_amd/host/seqio.hpp"

static int failures = 0, checks = 0;
#define EXPECT_EQ(a, b)                                                                                      \
    do {                                                                                                     \
        ++checks;                                                                                            \
        auto va__ = (a);                                                                                     \
        auto vb__ = (b);                                                                                     \
        if (!(va__ == vb__)) {                                                                               \
            ++failures;                                                                                      \
            std::cerr << __FILE__ << ":" << __LINE__ << " EXPECT_EQ(" #a ", " #b ") got " << va__ << " vs " << vb__ << "\n"; \
        }                                                                                                    \
    } while (0)
#define EXPECT_TRUE(a) EXPECT_EQ((bool)(a), true)
#define EXPECT_THROW(stmt, Exc)                                                              \
    do {                                                                                     \
        ++checks;                                                                            \
        bool ok__ = false;                                                                   \
        try { stmt; } catch (const Exc&) { ok__ = true; } catch (...) {}                     \
        if (!ok__) { ++failures; std::cerr << __FILE__ << ":" << __LINE__ << " expected " #Exc "\n"; } \
    } while (0)

using namespace interleave;

static std::vector<RefSeq> read_fasta(const std::string& path)
{
    seqio::Reader in(path);
    std::vector<RefSeq> out;
    std::string id, seq;
    while (in.read_record(id, seq)) out.push_back({id.substr(0, id.find(' ')), seq});
    return out;
}

int main(int argc, char** argv)
{
    if (argc < 3) { std::cerr << "usage: test_mirror <fixture dir> <tmp dir> [revcomp-of-n: 3 | 4]\n"; return 2; }
    const std::string fix = argv[1], tmp = argv[2];
    if (argc > 3) set_revcomp_of_n((uint32_t)std::stoul(argv[3]));  // the other candidate of the N rule, for the whole process
    std::filesystem::create_directories(tmp);

    // ---- IBFTest.CreateFilterTest (createfilter.hpp:43-203): build test.ibf from test.fasta
    IBFConfig config{};
    config.reference_files.emplace_back(fix + "/libIBFTests_test.fasta");
    config.output_filter_file = tmp + "/test.ibf";
    config.kmer_size = 13;
    config.threads_build = 1;
    config.fragment_length = 100000;
    IBF ibf{};
    FilterStats stats = ibf.create_filter(config, read_fasta(config.reference_files[0]));
    EXPECT_EQ(stats.invalidSeqs, 0u);
    EXPECT_EQ(stats.sumSeqLen, 72u);               // one pass over the 72-base cutOutNNNs string (the test doubles it by parsing twice)
    EXPECT_EQ(stats.totalBinsBinId, 1u);
    EXPECT_EQ(config.hash_functions, 3);           // createfilter.hpp:159
    EXPECT_EQ(config.filter_size_bits, 1236269ull * 64ull);  // createfilter.hpp:148 (79121216 for <= 63 bins)
    EXPECT_TRUE(std::filesystem::exists(config.output_filter_file));

    // ---- IBFTest.FilterStatsTest (createfilter.hpp:205-): test1.ibf from test1.fasta
    IBFConfig config1{};
    config1.reference_files.emplace_back(fix + "/libIBFTests_test1.fasta");
    config1.output_filter_file = tmp + "/test1.ibf";
    config1.kmer_size = 13;
    config1.fragment_length = 100000;
    IBF ibf1{};
    FilterStats stats1 = ibf1.create_filter(config1, read_fasta(config1.reference_files[0]));
    EXPECT_EQ(stats1.totalSeqsFile, 2u);
    EXPECT_EQ(stats1.totalBinsBinId, 2u);

    // ---- ReadTest.ClassificationTest (read.hpp:93-255): load both filters back from disk
    std::vector<IBFMeta> filters{};
    std::vector<TIbf> IBFs;
    for (std::string file : {tmp + "/test.ibf", tmp + "/test1.ibf"}) {
        IBFMeta filter{};
        filter.name = std::filesystem::path(file).stem().string();
        IBF f{};
        IBFConfig FilterIBFconfig{};
        FilterIBFconfig.input_filter_file = file;
        FilterStats st = f.load_filter(FilterIBFconfig);
        EXPECT_EQ(FilterIBFconfig.kmer_size, 13);  // load_filter sets config.kmer_size (IBFBuild.cpp:381)
        EXPECT_TRUE(st.totalBinsFile >= 1);
        filter.filter = f.getFilter();
        filters.emplace_back(filter);
        IBFs.emplace_back(f.getFilter());
    }
    EXPECT_EQ(IBFs.size(), 2u);

    ClassifyConfig cconf{};
    cconf.error_rate = 0.1;
    cconf.significance = 0.95;
    cconf.strata_filter = (uint16_t)-1;

    TInterval ci = calculateCI(cconf.error_rate, (uint8_t)IBFs[0].kmerSize, 35, cconf.significance);
    EXPECT_EQ(ci.first, 5);    // read.hpp:156
    EXPECT_EQ(ci.second, 30);  // read.hpp:157
    int16_t threshold = (int16_t)(35 - IBFs[0].kmerSize + 1 - ci.second);
    EXPECT_EQ(threshold, -7);  // read.hpp:164

    Read read;
    for (int i = 0; i < 6; ++i) read.sequence += "AAAAAAAACCCCCCCCCGAGAGAGGAGAGAGGAGAGAGAGAGCCCCAAAAGAGAGGAGA";  // read.hpp:22
    read.id = "kat";

    std::vector<TIbf> emptyVector;
    EXPECT_THROW(read.classify(emptyVector, cconf), NullFilterException);  // read.hpp:188
    EXPECT_EQ(IBFs[0].kmerSize, 13u);                                      // read.hpp:199
    EXPECT_EQ(read.getReadLength(), 354u);                                 // read.hpp:200
    EXPECT_EQ(read.classify(IBFs, cconf), true);                           // read.hpp:202
    std::vector<IBFMeta> emptyVectorMeta;
    EXPECT_THROW(read.classify(emptyVectorMeta, cconf), NullFilterException);  // read.hpp:208

    // count_matches through the pair overload, one filter per side: 282 and 182 (read.hpp:221-229)
    std::vector<IBFMeta> v1{filters[0]}, v2{filters[1]};
    std::pair<int, int> p = read.classify(v1, v2, cconf);
    EXPECT_EQ(p.first, 282);   // read.hpp:250-251
    EXPECT_EQ(p.second, 182);
    EXPECT_EQ(read.classify(filters, cconf), 0);  // read.hpp:231: best matching filter is test.ibf
    std::vector<IBFMeta> emptyM1, emptyM2;
    EXPECT_THROW(read.classify(emptyM1, emptyM2, cconf), NullFilterException);  // read.hpp:241
    EXPECT_THROW(read.classify(v1, emptyM2, cconf), NullFilterException);

    // short reads: list overloads throw ShortReadException, the pair overload skips the filters (IBFClassify.cpp:289-295, 318)
    Read shorty("short", "ACGTACGTACGT");
    EXPECT_THROW(shorty.classify(filters, cconf), ShortReadException);
    EXPECT_THROW(shorty.classify(IBFs, cconf), ShortReadException);
    std::pair<int, int> ps = shorty.classify(v1, v2, cconf);
    EXPECT_EQ(ps.first, 0);
    EXPECT_EQ(ps.second, 0);

    // N on the reverse strand (TSeqRevComp = ModReverse<ModComplementDna<Dna5String>>, IBF.hpp:96-97: the four-letter functor
    // sees N as A, so the reverse strand holds T there): the reverse complement of test.fasta[30:72] with the A that mirrors a T
    // replaced by N shares all 30 13-mers of that window on its reverse strand -- 18 (below the threshold of 25 at r = 0.001,
    // i.e. a count of 0) if the reverse strand saw N.  tests/test_oracle_kat.py holds the same read.
    Read nrev("n_reverse", "ATAATATATAANATCTCCTCTCTTTTGGGGCTCTCTCTCTCC");
    ClassifyConfig tight = cconf;
    tight.error_rate = 0.001;
    std::pair<int, int> pn = nrev.classify(v1, v2, tight);
    EXPECT_EQ(pn.first, 30);
    EXPECT_EQ(pn.second, 0);

    // the 35-mer of read.hpp:113: threshold -7 wraps to 65529 in production code -> no match
    Read mer("35mer", "AAAAAAACCCCCCCCCGAGAGAGGAGAGAGGAGAG");
    EXPECT_EQ(mer.classify(filters, cconf), -1);

    // select_matches with a threshold of 0 (IBFClassify.cpp:16-38): at k = 13, r = 0.1 the CI upper bound equals the k-mer
    // count for 123..130 bp, the uint16_t threshold is 0 and `count >= 0` holds for every bin -- the bool overload says
    // true for a read without a single match, the argmax overload (max_matches -> 0) says -1
    {
        std::string nohit;  // fixed pseudo-random 126 bp sharing no 13-mer with the fixtures
        for (uint32_t i = 0, x = 777u; i < 126; ++i) { x = x * 1664525u + 1013904223u; nohit += "ACGT"[(x >> 24) & 3]; }
        EXPECT_EQ((int)rb_threshold(126, 13, 0.1, 0.95), 0);
        Read quirk("quirk", nohit);
        EXPECT_EQ(quirk.classify(filters, cconf), -1);
        EXPECT_EQ(quirk.classify(IBFs, cconf), true);
        Read longer("longer", nohit + nohit);  // 252 bp: threshold 18, no match -> false
        EXPECT_EQ(longer.classify(IBFs, cconf), false);
    }

    // ---- check_unblock (adaptive_sampling.hpp:35-113) on the same objects
    EXPECT_EQ((int)check_unblock(read, cconf, v1, v2), 0);            // hits both, also at r-0.02 -> keep sequencing
    EXPECT_EQ((int)check_unblock(read, cconf, v1, emptyVectorMeta), 1);   // deplete only, match -> unblock
    EXPECT_EQ((int)check_unblock(read, cconf, emptyVectorMeta, v2), 2);   // target only, match -> stop_receiving
    std::string unrelated;  // fixed pseudo-random sequence sharing no 13-mer with the fixtures
    for (uint32_t i = 0, x = 12345u; i < 354; ++i) { x = x * 1664525u + 1013904223u; unrelated += "ACGT"[(x >> 24) & 3]; }
    Read other("other", unrelated);
    EXPECT_EQ((int)check_unblock(other, cconf, emptyVectorMeta, v2), 1);  // target only, no match -> unblock
    EXPECT_EQ((int)check_unblock(other, cconf, v1, emptyVectorMeta), 0);
    EXPECT_THROW(check_unblock(shorty, cconf, v1, emptyVectorMeta), ShortReadException);
    EXPECT_THROW(check_unblock(read, cconf, emptyVectorMeta, emptyVectorMeta), NullFilterException);

    // ---- load_filter error conventions (IBFBuild.cpp:329-376)
    {
        IBF f{};
        IBFConfig c{};
        EXPECT_THROW(f.load_filter(c), MissingIBFFileException);
        c.input_filter_file = fix + "/libIBFTests_test.fasta";
        EXPECT_THROW(f.load_filter(c), ParseIBFFileException);
        c.input_filter_file = tmp + "/does_not_exist.ibf";
        EXPECT_THROW(f.load_filter(c), ParseIBFFileException);
    }

    // ---- update_filter (IBFBuild.cpp:223-321): add test1.fasta's sequences to test.ibf as new bins
    {
        std::filesystem::copy_file(tmp + "/test.ibf", tmp + "/upd.ibf", std::filesystem::copy_options::overwrite_existing);
        IBF f{};
        IBFConfig c{};
        c.update_filter_file = tmp + "/upd.ibf";
        c.fragment_length = 100000;
        FilterStats us = f.update_filter(c, read_fasta(fix + "/libIBFTests_test1.fasta"));
        EXPECT_EQ(us.totalBinsFile, 1u);
        EXPECT_EQ(us.newBins, 2u);
        EXPECT_EQ(us.totalBinsBinId, 3u);
        EXPECT_EQ(getNumberOfBins(f.getFilter()), 3u);
        std::vector<IBFMeta> upd{IBFMeta{f.getFilter(), "upd", 0}};
        std::pair<int, int> pu = read.classify(upd, v2, cconf);
        EXPECT_EQ(pu.first, 282);  // max over the old bin (282) and the two new bins (182 between them)
        EXPECT_EQ(pu.second, 182);
        IBF g{};
        IBFConfig cg{};
        cg.input_filter_file = tmp + "/upd.ibf";
        EXPECT_EQ(g.load_filter(cg).totalBinsFile, 3u);
    }

    // ---- the reference's threading model (adaptive_sampling.hpp:745-751): N classification threads share the filters
    // read-only and call check_unblock one read at a time; every thread gets its own engine (streams, workspaces)
    {
        const int n_threads = 6, per_thread = 300;
        std::vector<int> bad(n_threads, 0);
        std::vector<std::thread> pool;
        for (int t = 0; t < n_threads; ++t) {
            pool.emplace_back([&, t] {
                ClassifyConfig c = cconf;  // the reference shares one mutable config between its threads: a data race there
                for (int i = 0; i < per_thread; ++i) {
                    try {
                        if ((int)check_unblock(read, c, v1, v2) != 0) ++bad[t];
                        if ((int)check_unblock(read, c, v1, emptyVectorMeta) != 1) ++bad[t];
                        if ((int)check_unblock(other, c, emptyVectorMeta, v2) != 1) ++bad[t];
                        std::pair<int, int> q = read.classify(v1, v2, c);
                        if (q.first != 282 || q.second != 182) ++bad[t];
                        if (read.classify(filters, c) != 0) ++bad[t];
                    } catch (...) {
                        ++bad[t];
                    }
                }
            });
        }
        for (std::thread& th : pool) th.join();
        int total_bad = 0;
        for (int b : bad) total_bad += b;
        EXPECT_EQ(total_bad, 0);
    }

    std::cout << "mirror checks: " << checks << ", failures: " << failures << std::endl;
    return failures ? 1 : 0;
}

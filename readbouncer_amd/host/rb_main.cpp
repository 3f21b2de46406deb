// rb_main.cpp -- `readbouncer_amd --config file.toml`: the build and classify usages of ReadBouncer's CLI
// (src/main/main.cpp:274-406, src/main/parser.hpp:13-37, src/main/ibfbuild.hpp:21-180,
// src/main/classify.hpp:142-380) on top of the C++ mirror.  The chunk loop of classify_reads is run
// batch-wise: all reads of a batch are classified on chunk i in one GPU launch, reads that are still
// unclassified go on to chunk i+1 -- per read this is the reference's loop (classify.hpp:262-299).
// usage = "target" runs as an offline replay of the live classification step (pre-basecalled chunks from read_files
// through rb_live_*); the MinKNOW client, the basecallers and usage "test" (connection test) are out of scope.
#include <sys/resource.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <exception>
#include <memory>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

#include "../../include/readbouncer_amd.hpp"
#include "../../include/readbouncer_amd_tuning.h"  // --calibrate only
#include "config_reader.hpp"
#include "seqio.hpp"

bool ConfigReader::filterException(std::filesystem::path& file) { return rb_is_ibf_file(file.string().c_str()) == 1; }

// plain-text stand-in for the reference's spdlog "ReadBouncerLog" (src/main/main.cpp:97): <log_directory>/ReadBouncerLog.txt
static std::ofstream g_log;
static void log_line(const std::string& level, const std::string& msg)
{
    if (g_log.is_open()) g_log << "[" << level << "] " << msg << std::endl;
}

// results struct of the reference's tests (classify.hpp:127-134)
struct ClassificationResults
{
    uint64_t found = 0;
    uint16_t failed = 0;
    uint64_t too_short = 0;
    uint64_t readCounter = 0;
} ClassificationResults_;

// buildIBF, src/main/ibfbuild.hpp:21-59
static interleave::TIbf buildIBF(ConfigReader config_reader, const std::string reference_file,
                                 const std::string bloom_filter_output_path)
{
    interleave::IBFConfig config{};
    config.reference_files.emplace_back(reference_file);
    config.output_filter_file = bloom_filter_output_path;
    config.kmer_size = (uint16_t)config_reader.IBF_Parsed.size_k;
    config.threads_build = (uint16_t)config_reader.IBF_Parsed.threads;
    config.fragment_length = (uint64_t)config_reader.IBF_Parsed.fragment_size;
    const auto t_parse = std::chrono::steady_clock::now();
    seqio::Reader in(reference_file);
    if (!in.is_open()) throw interleave::FileParserException("Unable to open the file: " + reference_file);
    std::vector<interleave::RefSeq> records;
    std::string id, seq;
    try {
        while (in.read_record(id, seq)) records.push_back({id.substr(0, id.find(' ')), seq});
    } catch (const std::exception& e) {
        throw interleave::FileParserException("ERROR: Problems parsing the file: " + reference_file + "[" + e.what() + "]");
    }
    const double parse_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_parse).count();
    interleave::IBF filter{};
    const auto t0 = std::chrono::steady_clock::now();
    interleave::FilterStats stats = filter.create_filter(config, records);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // where a build's time goes (profiles/cli_build.py reads this line): parsing the FASTA, cutOutNNNs + sizing, the filter's allocation
    // in HBM, the concatenation, rb_dibf_insert (H2D + the insert kernel), download + file
    std::cout << "BUILD_PHASES file=" << reference_file << " parse_s=" << parse_s << " load_seq_s=" << stats.timeLoadSeq
              << " alloc_filter_s=" << stats.timeAllocFilter << " concat_s=" << stats.timeConcat << " insert_s=" << stats.timeInsert
              << " save_s=" << stats.timeSaveFilter << " create_filter_s=" << stats.timeIBF << " bins=" << stats.totalBinsFile
              << " bases=" << stats.sumSeqLen << std::endl;
    const uint64_t validSeqs = stats.totalSeqsFile - stats.invalidSeqs;
    std::cerr << "IBF-build processed " << validSeqs << " sequences (" << stats.sumSeqLen / 1000000.0 << " Mbp) in " << secs
              << " seconds" << std::endl;
    if (stats.invalidSeqs > 0) std::cerr << " - " << stats.invalidSeqs << " invalid sequences were skipped" << std::endl;
    std::cerr << " - " << validSeqs << " sequences in " << stats.totalBinsFile + stats.newBins
              << " bins were written to the IBF" << std::endl;
    return filter.getFilter();
}

// getIBF, src/main/ibfbuild.hpp:69-180: load the file if it is an IBF, else build one from the FASTA.  Filters that only have
// to be LOADED are read and sent to the GPU side by side (a thread each: four 10-20 MB filters cost the time of one, part of every
// run's fixed cost); a list with a FASTA in it is worked through in order like the reference's.  The messages keep the list's order.
static std::vector<interleave::IBFMeta> getIBF(ConfigReader config, bool depleteFilter, bool targetFilter)
{
    const std::vector<std::filesystem::path>& files =
        depleteFilter ? config.IBF_Parsed.deplete_files : (targetFilter ? config.IBF_Parsed.target_files : std::vector<std::filesystem::path>{});
    struct Loaded { interleave::IBFMeta meta; std::string message; std::exception_ptr error; };
    std::vector<Loaded> loaded(files.size());
    bool all_ibf = files.size() > 1;
    for (std::filesystem::path file : files) all_ibf = all_ibf && config.filterException(file);
    auto load_one = [&](size_t i) {
        try {
            std::filesystem::path file = files[i];
            interleave::IBFMeta& filter = loaded[i].meta;
            filter.name = file.stem().string();
            if (config.filterException(file)) {
                interleave::IBF tf{};
                interleave::IBFConfig cfg{};
                cfg.input_filter_file = file.string();
                const auto t0 = std::chrono::steady_clock::now();
                interleave::FilterStats stats = tf.load_filter(cfg);
                filter.filter = tf.getFilter();
                loaded[i].message = std::to_string(stats.totalBinsFile) + " bins were loaded in " +
                                    std::to_string(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()) +
                                    " seconds from the IBF";
            } else {
                std::filesystem::path out_path = std::filesystem::path(config.output_dir);
                out_path /= file.filename();
                out_path.replace_extension("ibf");
                filter.filter = buildIBF(config, file.string(), out_path.string());
            }
        } catch (...) {
            loaded[i].error = std::current_exception();
        }
    };
    if (all_ibf) {
        std::vector<std::thread> th;
        for (size_t i = 0; i < files.size(); ++i) th.emplace_back(load_one, i);
        for (std::thread& t : th) t.join();
    } else {
        for (size_t i = 0; i < files.size(); ++i) load_one(i);
    }
    std::vector<interleave::IBFMeta> out;
    for (Loaded& l : loaded) {
        if (l.error) std::rethrow_exception(l.error);
        if (!l.message.empty()) std::cerr << l.message << std::endl;
        out.emplace_back(std::move(l.meta));
    }
    return out;
}

// --verify-ibf: does this .ibf hold exactly what the reference fragmenter + insertKmer put there for this FASTA?  The
// first thing to run on a filter written by the reference itself: the hash and layout constants of the IBF are restated
// from SeqAn (readbouncer_amd/csrc/ibf_spec.h), and a file is where a wrong one would first show.
static int verify_ibf(const std::string& ibf_path, const std::string& fasta_path, uint64_t fragment_size)
{
    seqio::Reader in(fasta_path);
    if (!in.is_open()) { std::cerr << "ERROR: Unable to open the file: " << fasta_path << std::endl; return 1; }
    std::vector<interleave::RefSeq> records;
    std::string id, seq;
    while (in.read_record(id, seq)) records.push_back({id.substr(0, id.find(' ')), seq});
    interleave::IBFConfig cfg{};
    cfg.input_filter_file = ibf_path;
    cfg.fragment_length = fragment_size;
    interleave::IBF f{};
    interleave::IBF::VerifyReport rep = f.verify_filter(cfg, records);
    if (!rep.warning.empty()) std::cout << "WARNING: " << rep.warning << std::endl;
    const double load = rep.bits.payload_bits ? (double)rep.bits.file_bits / (double)rep.bits.payload_bits : 0.0;
    std::cout << "VERIFY file=" << ibf_path << " kmer_size=" << cfg.kmer_size << " bins_file=" << rep.bins_file
              << " bins_expected=" << rep.bins_expected << " file_bits=" << rep.bits.file_bits
              << " rebuilt_bits=" << rep.bits.rebuilt_bits << " new_bits=" << rep.bits.new_bits
              << " explained=" << (rep.bits.file_bits ? (double)(rep.bits.rebuilt_bits - rep.bits.new_bits) / (double)rep.bits.file_bits : 0.0)
              << " load=" << load << std::endl;
    if (rep.ok()) {
        std::cout << "VERIFY OK: re-inserting " << records.size() << " sequences sets no new bit" << std::endl;
        return 0;
    }
    if (rep.bins_file != rep.bins_expected)
        std::cout << "VERIFY FAILED: the file has " << rep.bins_file << " bins, these sequences at fragment_size " << fragment_size
                  << " make " << rep.bins_expected << " (wrong FASTA or fragment size?)" << std::endl;
    else
        std::cout << "VERIFY FAILED: " << rep.bits.new_bits << " of " << rep.bits.rebuilt_bits
                  << " re-inserted bits are clear in the file -- hash/layout constants (ibf_spec.h) or k-mer encoding differ" << std::endl;
    return 3;
}

// (seqan::Dna5String) of a character, printed back: ACGT in either case -> upper case, U/u -> T, everything else -> N
static const struct Dna5CharTable {
    char t[256];
    Dna5CharTable()
    {
        for (int c = 0; c < 256; ++c) t[c] = 'N';
        t['A'] = t['a'] = 'A'; t['C'] = t['c'] = 'C'; t['G'] = t['g'] = 'G'; t['T'] = t['t'] = 'T'; t['U'] = t['u'] = 'T';
    }
    char operator[](unsigned char c) const { return t[c]; }
} kDna5Char;

// n bytes of a read through that table.  unclassified.fasta is most of what a depletion run writes (every read that is
// not a target), so the table walk is given a 32-bytes-at-a-time form where the CPU has AVX2: fold case (c & 0xDF),
// compare with A C G T U, blend.
#if defined(__x86_64__) && defined(__GNUC__)
#include <immintrin.h>
__attribute__((target("avx2"))) static void dna5_map_avx2(char* dst, const char* src, size_t n)
{
    const __m256i fold = _mm256_set1_epi8((char)0xDF), cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'),
                  cG = _mm256_set1_epi8('G'), cT = _mm256_set1_epi8('T'), cU = _mm256_set1_epi8('U'), cN = _mm256_set1_epi8('N');
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        const __m256i v = _mm256_and_si256(_mm256_loadu_si256((const __m256i*)(src + i)), fold);
        const __m256i isU = _mm256_cmpeq_epi8(v, cU);
        const __m256i ok = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(v, cA), _mm256_cmpeq_epi8(v, cC)),
                                           _mm256_or_si256(_mm256_cmpeq_epi8(v, cG), _mm256_cmpeq_epi8(v, cT)));
        __m256i r = _mm256_blendv_epi8(cN, v, ok);
        r = _mm256_blendv_epi8(r, cT, isU);
        _mm256_storeu_si256((__m256i*)(dst + i), r);
    }
    for (; i < n; ++i) dst[i] = kDna5Char[(unsigned char)src[i]];
}
static const bool kHaveAvx2 = __builtin_cpu_supports("avx2");
#else
static const bool kHaveAvx2 = false;
static void dna5_map_avx2(char*, const char*, size_t) {}
#endif
static inline void dna5_map(char* dst, const char* src, size_t n)
{
    if (kHaveAvx2) { dna5_map_avx2(dst, src, n); return; }
    for (size_t i = 0; i < n; ++i) dst[i] = kDna5Char[(unsigned char)src[i]];
}

struct ReadState
{
    bool classified = false, failed = false;
    int best = -1;
};

// host pipeline knobs (command line)
struct IngestOptions
{
    size_t batch_reads = 65536;  // most reads per GPU call
    unsigned threads = 6;        // parser threads (one does ~3 GB/s of 250 bp FASTQ records)
    unsigned classify_threads = 0;  // threads running the chunk loop on the GPU and formatting their segment's output (one engine
                                    // each); 0 = [IBF] threads when the TOML sets it above 1 (the reference's meaning of that key:
                                    // classification threads, adaptive_sampling.hpp:745), else 4: since the engines of a device share
                                    // the merged copy of their filters the GPU side of a segment is short, and what a fifth and sixth
                                    // thread add is contention among the formatters (profiles/r04/cli_classifier_threads_*.txt, README
                                    // shape, M reads/s in-process with 4 / 5 / 6 threads: 23.8 / 21.9 / 21.2 on 16 M reads, 29.6 / 31.6 /
                                    // 28.0 on 64 M)
    bool calibrate = false;      // --calibrate: every classifier thread's engine fits the windows of its clock-phased gathers to this device first
                                 // (rb_engine_calibrate, one engine after the other: ~30 ms each; pays on long runs over narrow filters)
    bool mmap_output = false;    // --mmap-output: the classifier threads write the outputs through shared mappings of the files instead of
                                 // positional writes of a small buffer each (measured slower on tmpfs: page faults on fresh page-cache pages)
    size_t segment_mb = 32;      // file bytes per parsed segment (one GPU call of ~65 k reads of 250 bp; the page-locked staging blocks scale with it)
    size_t segment_bytes = 0;    // tests: segments far smaller than a megabyte (0 = segment_mb)
    size_t live_batch = 64;      // usage "target" replay: chunks per micro-batch
    size_t bytes() const { return segment_bytes ? segment_bytes : (segment_mb << 20); }
};

// what a classified segment adds to the outputs: exact byte counts per output file (target files first, unclassified.fasta last)
// -- known before a byte is formatted, so that the files can be laid out in read order while the formatting itself runs in
// parallel -- and the tallies of the segment
struct SegmentOutput
{
    std::vector<uint64_t> bytes;         // one per target file, unclassified.fasta last
    std::vector<uint64_t> per_target;    // IBFMeta.classified increments (classify.hpp:80,288)
    std::vector<std::string> error_lines;
    uint64_t found = 0, failed = 0;
};

// classify_reads, src/main/classify.hpp:142-380, as a pipeline: parser threads cut the memory-mapped read file into
// segments and stage the first chunk of every read (seqio::ParallelReader); classifier threads take the segments in file
// order, run the chunk loop on the GPU (one engine each: while one waits for the GPU the others prepare and format), size
// their segment's share of every output file, reserve it in file order (seqio::OrderedOutput) and then format the FASTA
// text straight into the files' page cache, all of them at once -- outputs byte-identical to a serial run.
static void classify_reads(ConfigReader& config, std::vector<interleave::IBFMeta> DepletionFilters,
                           std::vector<interleave::IBFMeta> TargetFilters, const IngestOptions& opt,
                           const std::vector<int>& devices)
{
    interleave::ClassifyConfig Conf{};
    std::unique_ptr<interleave::MultiDeviceClassifier> multi;
    if (devices.size() > 1 && (!DepletionFilters.empty() || !TargetFilters.empty()))
        multi.reset(new interleave::MultiDeviceClassifier(devices, DepletionFilters, TargetFilters));
    const bool deplete = DepletionFilters.size() >= 1, target = TargetFilters.size() >= 1;
    if (!deplete && !target) {
        std::cerr << "[Error] No depletion or target filters have been provided for classification! " << '\n';
        exit(1);
    }
    for (std::filesystem::path read_file : config.IBF_Parsed.read_files) {
        Conf.strata_filter = (uint16_t)-1;
        Conf.significance = 0.95;
        Conf.error_rate = config.IBF_Parsed.error_rate;
        uint64_t found = 0, too_short = 0, readCounter = 0;
        uint16_t failed = 0;
        double classify_seconds = 0.0;
        uint64_t classify_reads_n = 0;
        const auto wall0 = std::chrono::steady_clock::now();

        const size_t n_out = TargetFilters.size() + 1;  // <target>.fasta ..., unclassified.fasta last
        std::vector<std::unique_ptr<seqio::OrderedOutput>> outputs;
        for (size_t i = 0; i < n_out; ++i) {
            std::filesystem::path outfile(config.output_dir);
            outfile /= i < TargetFilters.size() ? TargetFilters[i].name + ".fasta" : std::string("unclassified.fasta");
            outputs.emplace_back(new seqio::OrderedOutput());
            const bool opened = outputs.back()->open(outfile.string(), opt.mmap_output);
            if (!opened && i + 1 == n_out) {  // (the reference checks unclassified.fasta only, classify.hpp:205-211)
                std::cerr << "ERROR: Unable to open the file: " << outfile.string() << std::endl;
                return;
            }
        }
        seqio::MappedFile mapped(read_file.string());
        if (!mapped.is_open()) {
            std::cerr << "ERROR: Unable to open the file: " << read_file.string() << std::endl;
            return;
        }
        std::cout << '\n' << "Classification results of: " << read_file.string() << '\n' << '\n';

        const uint32_t chunk_length = (uint32_t)config.IBF_Parsed.chunk_length;
        const uint32_t max_chunks = (uint8_t)config.IBF_Parsed.max_chunks;  // "uint8_t i" in the reference

        // pass 1 over a classified segment: which file every read goes to and how many bytes it takes there (classify.hpp:289-316)
        auto fasta_bytes = [&](const seqio::Record& r, bool one_line) -> uint64_t {
            // target FASTAs: `targetFastas[i] << ">" << id << endl << seq << endl` -- the raw read on one line;
            // unclassified.fasta: seqan::writeRecord(..., (seqan::Dna5String)seq) -- SeqAn's default 70-column lines
            return 1 + (uint64_t)r.id_len + 1 + r.seq_len + (one_line ? 1 : (r.seq_len + 69) / 70);
        };
        auto size_segment = [&](const seqio::Segment& seg, const std::vector<ReadState>& state) {
            SegmentOutput so;
            so.bytes.assign(n_out, 0);
            so.per_target.assign(TargetFilters.size(), 0);
            const std::vector<seqio::Record>& recs = seg.batch.records;
            for (size_t i = 0; i < recs.size(); ++i) {
                const seqio::Record& r = recs[i];
                if (r.seq_len < chunk_length) continue;
                if (state[i].failed) {  // classify.hpp:306-316
                    so.failed++;
                    so.error_lines.push_back("Error classifying Read : " + std::string(r.id, r.id_len) + "(Len=" + std::to_string(r.seq_len) + ")");
                    continue;
                }
                if (state[i].classified) {
                    so.found++;
                    if (!(target && state[i].best >= 0)) continue;
                    so.per_target[state[i].best] += 1;
                    so.bytes[state[i].best] += fasta_bytes(r, true);
                } else {
                    so.bytes[n_out - 1] += fasta_bytes(r, false);
                }
            }
            return so;
        };
        // pass 2: the FASTA text of the segment, written where pass 1 and the reservation said (out[f] = cursor over this segment's share of file f)
        auto format_segment = [&](const seqio::Segment& seg, const std::vector<ReadState>& state, std::vector<seqio::OrderedOutput::Writer>& out) {
            const std::vector<seqio::Record>& recs = seg.batch.records;
            for (size_t i = 0; i < recs.size(); ++i) {
                const seqio::Record& r = recs[i];
                if (r.seq_len < chunk_length || state[i].failed) continue;
                size_t f = n_out - 1;  // unclassified.fasta
                if (state[i].classified) {
                    if (!(target && state[i].best >= 0)) continue;
                    f = (size_t)state[i].best;
                }
                const bool one_line = f != n_out - 1;
                char* dst = out[f].take((size_t)fasta_bytes(r, one_line));
                *dst++ = '>';
                std::memcpy(dst, r.id, r.id_len);
                dst += r.id_len;
                *dst++ = '\n';
                if (one_line) {
                    std::memcpy(dst, r.seq, r.seq_len);
                    dst[r.seq_len] = '\n';
                } else {
                    // the Dna5 alphabet (upper case, everything but ACGT[U] becomes N) in 70-column lines
                    for (size_t p = 0; p < r.seq_len; p += 70) {
                        const size_t n = std::min<size_t>(70, r.seq_len - p);
                        dna5_map(dst, r.seq + p, n);
                        dst[n] = '\n';
                        dst += n + 1;
                    }
                }
            }
        };

        // first chunks are staged in page-locked memory: the copy to the GPU is then a plain DMA (a pageable source is
        // pinned on the fly by the runtime, under the same mm lock the parser threads' page faults need)
        seqio::BlockAllocator pinned;
        pinned.alloc = [](size_t bytes) -> void* { void* p = nullptr; return rb_host_alloc(bytes, &p) == RB_OK ? p : nullptr; };
        pinned.release = [](void* p) { rb_host_free(p); };
        seqio::ParallelReader reader(mapped.data(), mapped.size(), opt.threads, opt.bytes(), chunk_length, pinned);
        double wait_reader_s = 0.0, wait_turn_s = 0.0, format_s = 0.0;
        auto seconds_since = [](std::chrono::steady_clock::time_point t) {
            return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count();
        };
        // the chunk loop of one segment (classify.hpp:262-299, batch-wise): the reads go to the GPU in calls of at most
        // batch_reads; chunk 0 comes ready-made from the parser threads, later chunks are gathered here for the reads that
        // are still unclassified
        auto classify_segment = [&](const seqio::Segment& seg, std::vector<ReadState>& state) {
            const std::vector<seqio::Record>& recs = seg.batch.records;
            std::vector<char> flat;
            std::vector<uint64_t> offs;
            std::vector<uint32_t> lens;
            std::vector<size_t> idx;
            for (size_t b0 = 0; b0 < seg.prefix_idx.size(); b0 += opt.batch_reads) {
                const size_t b1 = std::min(seg.prefix_idx.size(), b0 + opt.batch_reads);
                std::vector<size_t> active(b1 - b0);
                for (size_t j = b0; j < b1; ++j) active[j - b0] = seg.prefix_idx[j];
                for (uint32_t c = 0; c < max_chunks && !active.empty(); ++c) {
                    const char* base = nullptr;
                    offs.clear();
                    lens.clear();
                    idx.clear();
                    if (c == 0) {
                        base = seg.prefix + b0 * (size_t)chunk_length;
                        for (size_t j = 0; j < active.size(); ++j) {
                            offs.push_back(j * (uint64_t)chunk_length);
                            lens.push_back(chunk_length);
                        }
                        idx = active;
                    } else {
                        flat.clear();
                        for (size_t i : active) {
                            const seqio::Record& r = recs[i];
                            uint64_t fragend = (uint64_t)(c + 1) * chunk_length, fragstart = (uint64_t)c * chunk_length;
                            if (fragend > r.seq_len) fragend = r.seq_len;
                            if (fragstart > fragend) { state[i].failed = true; continue; }  // undefined infix in the reference
                            offs.push_back(flat.size());
                            lens.push_back((uint32_t)(fragend - fragstart));
                            flat.insert(flat.end(), r.seq + fragstart, r.seq + fragend);
                            idx.push_back(i);
                        }
                        if (flat.empty()) flat.push_back('N');
                        base = flat.data();
                    }
                    std::vector<size_t> next;
                    if (!idx.empty()) {
                        interleave::BatchResult res =
                            multi ? multi->classify_flat(Conf, base, offs.data(), lens.data(), idx.size(), RB_MODE_CLASSIFY_CHUNK)
                                  : interleave::classify_batch_flat(DepletionFilters, TargetFilters, Conf, base, offs.data(),
                                                                    lens.data(), idx.size(), RB_MODE_CLASSIFY_CHUNK);
                        for (size_t j = 0; j < idx.size(); ++j) {
                            ReadState& st = state[idx[j]];
                            if (res.status[j] != RB_OK) { st.failed = true; continue; }  // exception -> failed++ (:306-316)
                            if (res.decision[j]) {
                                st.classified = true;
                                st.best = target ? res.best_target[j] : -1;
                            } else {
                                next.push_back(idx[j]);
                            }
                        }
                    }
                    active.swap(next);
                }
            }
        };
        // classifier threads: each takes the next segment in file order, runs its chunk loop on an engine of its own (the
        // mirror keeps one engine per calling thread) and hands the result to the writer when its turn has come -- while
        // one thread waits for the GPU, the other prepares and post-processes its segment
        std::mutex rmu, omu;
        std::condition_variable ocv;
        uint64_t next_seq = 0, write_seq = 0;
        std::string worker_error;
        std::mutex cal_mu;
        std::condition_variable cal_cv;
        unsigned cal_parties = 0, cal_arrived = 0;  // (cal_parties is set before the first classifier starts)
        auto classifier = [&] {
            try {
                if (opt.calibrate && !multi) {
                    // one engine at a time, and nobody classifies before the last engine is calibrated: an engine timed beside
                    // another thread's classification picks its windows from noise
                    std::unique_lock<std::mutex> lock(cal_mu);
                    uint32_t n_tables = 0, n_changed = 0;
                    int rc = RB_ERR_INVALID_ARG;
                    try {
                        rc = rb_engine_calibrate(interleave::detail::engine_for(DepletionFilters, TargetFilters), opt.batch_reads, chunk_length, 40.0,
                                                 &n_tables, &n_changed);
                    } catch (...) {
                        ++cal_arrived;  // (the others must not wait for a thread that is on its way out)
                        cal_cv.notify_all();
                        throw;
                    }
                    if (rc != RB_OK) log_line("warn", std::string("calibration skipped: ") + rb_last_error());
                    else log_line("info", "calibrated " + std::to_string(n_tables) + " phased table(s), " + std::to_string(n_changed) + " window(s) changed");
                    ++cal_arrived;
                    cal_cv.notify_all();
                    cal_cv.wait(lock, [&] { return cal_arrived >= cal_parties; });
                }
                for (;;) {
                    std::unique_ptr<seqio::Segment> seg;
                    uint64_t seq = 0;
                    {
                        std::lock_guard<std::mutex> lock(rmu);
                        const auto tr = std::chrono::steady_clock::now();
                        seg = reader.next();
                        wait_reader_s += seconds_since(tr);
                        if (!seg) return;
                        seq = next_seq++;
                        readCounter += seg->batch.records.size();
                        too_short += seg->batch.records.size() - seg->prefix_idx.size();  // classify.hpp:247-250
                        classify_reads_n += seg->prefix_idx.size();
                        if (!seg->batch.error.empty()) std::cerr << "ERROR: " << seg->batch.error << std::endl;
                    }
                    std::vector<ReadState> state(seg->batch.records.size());
                    const auto t0 = std::chrono::steady_clock::now();
                    classify_segment(*seg, state);
                    const double secs = seconds_since(t0);
                    SegmentOutput so = size_segment(*seg, state);
                    std::vector<uint64_t> at(n_out, 0);
                    {
                        // this segment's turn: its share of every output file starts where the segment before it ends
                        const auto tw = std::chrono::steady_clock::now();
                        std::unique_lock<std::mutex> lock(omu);
                        ocv.wait(lock, [&] { return write_seq == seq || write_seq == ~0ULL; });
                        if (write_seq == ~0ULL) return;  // another worker failed
                        wait_turn_s += seconds_since(tw);
                        classify_seconds += secs;
                        for (size_t f = 0; f < n_out; ++f)
                            if (outputs[f]->is_open()) at[f] = outputs[f]->reserve(so.bytes[f]);
                        found += so.found;
                        failed += (uint16_t)so.failed;
                        for (const std::string& l : so.error_lines) log_line("error", l);
                        for (size_t f = 0; f < TargetFilters.size(); ++f) TargetFilters[f].classified += so.per_target[f];
                        ++write_seq;
                    }
                    ocv.notify_all();
                    // ... and the text itself, by every classifier thread at once
                    const auto tf = std::chrono::steady_clock::now();
                    std::vector<seqio::OrderedOutput::Writer> out(n_out);
                    // (a file that could not be opened -- target FASTAs are not checked by the reference -- is formatted into the void)
                    for (size_t f = 0; f < n_out; ++f) out[f] = outputs[f]->writer(at[f], so.bytes[f]);
                    format_segment(*seg, state, out);
                    for (size_t f = 0; f < n_out; ++f) {
                        const bool complete = outputs[f]->is_open() ? out[f].finish() : true;
                        if (!complete) throw std::runtime_error("output layout mismatch");
                    }
                    out.clear();            // (the cursors first: their buffers are on their way to the files' writer threads)
                    const size_t seg_index = seg->index;
                    seg.reset();
                    reader.release(seg_index);  // the records of this segment are not looked at again
                    {
                        std::lock_guard<std::mutex> lock(omu);
                        format_s += seconds_since(tf);
                    }
                }
            } catch (const std::exception& ex) {
                std::lock_guard<std::mutex> lock(omu);
                if (worker_error.empty()) worker_error = ex.what();
                write_seq = ~0ULL;  // nobody's turn any more: the other workers stop at their hand-over
                ocv.notify_all();
            }
        };
        unsigned n_classifiers = opt.classify_threads ? opt.classify_threads
                                                      : (config.IBF_Parsed.threads > 1 ? (unsigned)config.IBF_Parsed.threads : 4u);
        {
            // (a multi-device pool spreads every call over its devices itself; two callers keep it fed while one formats)
            if (multi) n_classifiers = std::min(n_classifiers, 2u);
            cal_parties = n_classifiers;
            std::vector<std::thread> workers;
            for (unsigned i = 1; i < n_classifiers; ++i) workers.emplace_back(classifier);
            classifier();
            for (std::thread& t : workers) t.join();
        }
        if (!worker_error.empty()) throw std::runtime_error(worker_error);
        for (auto& o : outputs) {
            o->close();
            if (!o->ok()) std::cerr << "ERROR: writing an output file failed: " << o->error() << std::endl;
        }
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
        const double avg = classify_reads_n ? classify_seconds / (double)classify_reads_n : 0.0;
        std::cout << "------------------------------- Final Results -------------------------------" << std::endl;
        std::cout << "Number of classified reads                         :   " << found << std::endl;
        std::cout << "Number of of too short reads (len < " << config.IBF_Parsed.chunk_length << ")           :   " << too_short << std::endl;
        std::cout << "Number of all reads                                :   " << readCounter << std::endl;
        for (interleave::IBFMeta& f : TargetFilters)
            std::cout << f.name << "\t : " << f.classified << "\t\t" << ((float)f.classified) / ((float)readCounter) << std::endl;
        std::cout << "Average Processing Time Read Classification        :   " << avg << std::endl;
        std::cout << "-----------------------------------------------------------------------------------" << std::endl;
        log_line("info", "classified " + std::to_string(found) + " too_short " + std::to_string(too_short) + " failed " +
                             std::to_string(failed) + " all " + std::to_string(readCounter) + " reads of " + read_file.string());
        std::cout << "RESULT found=" << found << " failed=" << failed << " too_short=" << too_short
                  << " readCounter=" << readCounter << std::endl;
        std::cout << "THROUGHPUT reads_per_s=" << (wall > 0 ? (double)readCounter / wall : 0.0) << " wall_s=" << wall
                  << " classify_s=" << classify_seconds << " wait_reader_s=" << wait_reader_s
                  << " wait_turn_s=" << wait_turn_s << " format_s=" << format_s << " classifiers=" << n_classifiers
                  << " parsers=" << opt.threads << std::endl;
        ClassificationResults_.found = found;
        ClassificationResults_.failed = failed;
        ClassificationResults_.too_short = too_short;
        ClassificationResults_.readCounter = readCounter;
        for (interleave::IBFMeta& f : TargetFilters) f.classified = 0;
    }
}

// usage = "target" as an OFFLINE REPLAY of the live classification step (src/main/adaptive_sampling.hpp:214-356: what
// happens to a basecalled chunk between classification_queue and action_queue).  The MinKNOW gRPC client and the
// basecallers are out of scope; what they would deliver -- basecalled chunks, in arrival order -- comes from
// [IBF] read_files instead: one record per chunk, the first word of the record id is the read id (chunks of one read
// carry the same id and appear in the order they were sequenced; anything after the first blank, e.g. "ch=12 chunk=3",
// is carried along as a comment).  The chunks go through rb_live_process in micro-batches of `live_batch` records (what
// the basecaller threads would have queued while the GPU was busy), with the reference's bookkeeping: once_seen,
// concatenation of undecided chunks, the 1500 bp cut-off, decision -> action (Data::sendActions, Data.cpp:169-187).
// Output: <output_directory>/live_actions.tsv, one line per chunk in arrival order:
//   record  read_id  action(none|unblock_read|stop_receiving_data)  status  classified_length
// and the tallies the reference logs for a run.
static int replay_target(ConfigReader& config, std::vector<interleave::IBFMeta>& DepletionFilters,
                         std::vector<interleave::IBFMeta>& TargetFilters, size_t live_batch)
{
    rb_engine* engine = interleave::detail::engine_for(DepletionFilters, TargetFilters);
    rb_live* live = nullptr;
    // Conf.significance = 0.95, Conf.error_rate = exp_seq_error_rate (adaptive_sampling.hpp:563-566); 1500 bp cut-off (:315)
    interleave::throw_status(rb_live_create(engine, config.IBF_Parsed.error_rate, 0.95, 1500, &live), "rb_live_create");
    std::ofstream out(std::filesystem::path(config.output_dir) / "live_actions.tsv");
    out << "record\tread_id\taction\tstatus\tclassified_length\n";
    static const char* kAction[3] = {"none", "unblock_read", "stop_receiving_data"};
    uint64_t n_records = 0, n_unblock = 0, n_stop = 0, n_failed = 0, n_calls = 0;
    double classify_s = 0.0;
    const auto t_begin = std::chrono::steady_clock::now();
    for (const std::filesystem::path& read_file : config.IBF_Parsed.read_files) {
        seqio::MappedFile mapped(read_file.string());
        if (!mapped.is_open()) { rb_live_destroy(live); throw interleave::FileParserException("ERROR: Unable to open the file: " + read_file.string()); }
        seqio::Parser parser(mapped.data(), mapped.size());
        seqio::Batch batch;
        for (;;) {
            parser.next_batch(batch, live_batch);
            const size_t n = batch.records.size();
            if (n) {
                std::string ids, seqs;
                std::vector<uint64_t> id_off(n), off(n);
                std::vector<uint32_t> id_len(n), len(n);
                for (size_t i = 0; i < n; ++i) {
                    const seqio::Record& r = batch.records[i];
                    size_t w = 0;
                    while (w < r.id_len && r.id[w] != ' ' && r.id[w] != '\t') ++w;
                    id_off[i] = ids.size(); id_len[i] = (uint32_t)w; ids.append(r.id, w);
                    off[i] = seqs.size(); len[i] = (uint32_t)r.seq_len; seqs.append(r.seq, r.seq_len);
                }
                if (seqs.empty()) seqs.push_back('N');
                std::vector<uint8_t> action(n), status(n);
                std::vector<uint32_t> clen(n);
                const auto t0 = std::chrono::steady_clock::now();
                const int rc = rb_live_process(live, ids.data(), id_off.data(), id_len.data(), seqs.data(), off.data(), len.data(), n,
                                               action.data(), status.data(), clen.data());
                classify_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (rc != RB_OK) { rb_live_destroy(live); interleave::throw_status(rc, "rb_live_process"); }
                ++n_calls;
                for (size_t i = 0; i < n; ++i) {
                    out << n_records + i << '\t';
                    out.write(ids.data() + id_off[i], id_len[i]);
                    out << '\t' << kAction[action[i] < 3 ? action[i] : 0] << '\t' << (int)status[i] << '\t' << clen[i] << '\n';
                    n_unblock += action[i] == 1;
                    n_stop += action[i] == 2;
                    if (status[i] != RB_OK) {
                        ++n_failed;  // adaptive_sampling.hpp:340-349: logged, no action, state untouched
                        log_line("error", "Error classifying Read : " + std::string(ids.data() + id_off[i], id_len[i]) + "(Len=" + std::to_string(len[i]) + ")");
                    }
                }
                n_records += n;
            }
            if (!batch.error.empty()) { rb_live_destroy(live); throw interleave::FileParserException("ERROR: " + batch.error); }
            if (batch.eof) break;
        }
    }
    const size_t pending = rb_live_pending(live);
    rb_live_destroy(live);
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    std::cout << "------------------------------- Live Replay Results -------------------------------" << std::endl;
    std::cout << "Number of chunks replayed                          :   " << n_records << std::endl;
    std::cout << "Number of unblock_read actions                     :   " << n_unblock << std::endl;
    std::cout << "Number of stop_receiving_data actions              :   " << n_stop << std::endl;
    std::cout << "Number of chunks that failed to classify           :   " << n_failed << std::endl;
    std::cout << "Reads still waiting for more data (once_seen)      :   " << pending << std::endl;
    std::cout << "LIVE chunks=" << n_records << " unblock=" << n_unblock << " stop=" << n_stop << " failed=" << n_failed
              << " pending=" << pending << " calls=" << n_calls << " classify_s=" << classify_s << " wall_s=" << wall << std::endl;
    log_line("info", "live replay: " + std::to_string(n_records) + " chunks, " + std::to_string(n_unblock) + " unblock, " +
                         std::to_string(n_stop) + " stop_receiving");
    return 0;
}

static int run_program(ConfigReader& config, const IngestOptions& opt, const std::vector<int>& devices)
{
    config.parse();
    config.createLog(config.usage);  // main.cpp:283
    g_log.open(config.log_dir / "ReadBouncerLog.txt", std::ios::app);
    log_line("info", "usage " + config.usage);
    if (config.usage == "build") {  // main.cpp:286-344
        for (const auto& files : {config.IBF_Parsed.target_files, config.IBF_Parsed.deplete_files}) {
            for (std::filesystem::path file : files) {
                std::filesystem::path out = config.output_dir;
                out /= file.filename();
                out.replace_extension("ibf");
                buildIBF(config, file.string(), out.string());
            }
        }
        return 0;
    }
    if (config.usage == "classify") {  // main.cpp:346-376
        const auto t_load = std::chrono::steady_clock::now();
        std::vector<interleave::IBFMeta> DepletionFilters = getIBF(config, true, false);
        std::vector<interleave::IBFMeta> TargetFilters = getIBF(config, false, true);
        const double load_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_load).count();
        const auto t_cls = std::chrono::steady_clock::now();
        classify_reads(config, DepletionFilters, TargetFilters, opt, devices);
        std::cout << "PHASES load_filters_s=" << load_s << " classify_reads_s="
                  << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_cls).count() << std::endl;
        return 0;
    }
    if (config.usage == "target") {  // main.cpp:365-378, with the chunk source replaced (see replay_target)
        if (config.IBF_Parsed.read_files.empty()) {  // before any filter is loaded or built
            std::cerr << "usage \"target\": the live MinKNOW/basecaller connection is outside this engine's scope; list pre-basecalled "
                         "chunk files under [IBF] read_files to replay them through the live classification step" << std::endl;
            return 2;
        }
        std::vector<interleave::IBFMeta> DepletionFilters = getIBF(config, true, false);
        std::vector<interleave::IBFMeta> TargetFilters = getIBF(config, false, true);
        return replay_target(config, DepletionFilters, TargetFilters, opt.live_batch);
    }
    std::cerr << "usage \"" << config.usage << "\" is outside this engine's scope (supported: build, classify, target as an offline replay)" << std::endl;
    return 2;
}

int main(int argc, char const* argv[])
{
    std::string config_path;
    bool dump_only = false;
    IngestOptions opt;
    bool no_digest = false;
    std::vector<int> devices{0};
    std::string verify_path, verify_ref;
    uint64_t verify_fragment = 100000;  // [IBF] fragment_size default (configReader.cpp)
    for (int i = 1; i < argc; ++i) {
        if (!std::strcmp(argv[i], "--devices") && i + 1 < argc) {  // e.g. --devices 0,1,2,3,4,5,6,7
            devices.clear();
            std::string list = argv[++i];
            size_t pos = 0;
            while (pos <= list.size()) {
                const size_t comma = list.find(',', pos);
                const std::string tok = list.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
                if (!tok.empty()) devices.push_back(std::stoi(tok));
                if (comma == std::string::npos) break;
                pos = comma + 1;
            }
            if (devices.empty()) devices.push_back(0);
            continue;
        }
        if ((!std::strcmp(argv[i], "--config") || !std::strcmp(argv[i], "-c")) && i + 1 < argc) config_path = argv[++i];
        else if (!std::strcmp(argv[i], "--dump-config")) dump_only = true;
        else if (!std::strcmp(argv[i], "--verify-ibf") && i + 1 < argc) verify_path = argv[++i];
        else if (!std::strcmp(argv[i], "--reference") && i + 1 < argc) verify_ref = argv[++i];
        else if (!std::strcmp(argv[i], "--fragment-size") && i + 1 < argc) verify_fragment = (uint64_t)std::stoull(argv[++i]);
        else if (!std::strcmp(argv[i], "--no-digest")) no_digest = true;
        else if (!std::strcmp(argv[i], "--batch-reads") && i + 1 < argc) opt.batch_reads = std::max<size_t>(1, (size_t)std::stoull(argv[++i]));
        else if (!std::strcmp(argv[i], "--ingest-threads") && i + 1 < argc) opt.threads = (unsigned)std::max(1, std::stoi(argv[++i]));
        else if (!std::strcmp(argv[i], "--classify-threads") && i + 1 < argc) opt.classify_threads = (unsigned)std::max(1, std::stoi(argv[++i]));
        else if (!std::strcmp(argv[i], "--calibrate")) opt.calibrate = true;
        // tables of 1 GiB and more are placed by trial (up to five allocations probed, 1-2 s at load time): 1 switches it off
        else if (!std::strcmp(argv[i], "--placement-tries") && i + 1 < argc) {
            if (rb_set_placement_tries(std::stoi(argv[++i])) != RB_OK) { std::cerr << "ERROR: --placement-tries 0 .. 8" << std::endl; return 1; }
        }
        else if (!std::strcmp(argv[i], "--revcomp-of-n") && i + 1 < argc) interleave::set_revcomp_of_n((uint32_t)std::stoul(argv[++i]));
        else if (!std::strcmp(argv[i], "--mmap-output")) opt.mmap_output = true;
        else if (!std::strcmp(argv[i], "--no-mmap-output")) opt.mmap_output = false;
        else if (!std::strcmp(argv[i], "--segment-mb") && i + 1 < argc) opt.segment_mb = std::max<size_t>(1, (size_t)std::stoull(argv[++i]));
        else if (!std::strcmp(argv[i], "--segment-bytes") && i + 1 < argc) opt.segment_bytes = (size_t)std::stoull(argv[++i]);
        else if (!std::strcmp(argv[i], "--live-batch") && i + 1 < argc) opt.live_batch = std::max<size_t>(1, (size_t)std::stoull(argv[++i]));
        else if (!std::strcmp(argv[i], "--parse-stats") && i + 1 < argc) {
            // ingest self-check (no GPU): records, bases and an FNV-1a digest over "id\tseq\n" of every record
            seqio::MappedFile mf(argv[++i]);
            if (!mf.is_open()) { std::cerr << "ERROR: Unable to open the file: " << argv[i] << std::endl; return 1; }
            // through the parallel reader (--ingest-threads / --segment-bytes given BEFORE --parse-stats apply): the digest is
            // over the records in file order, so it is the same for every thread count and segment size
            uint64_t n = 0, bases = 0, h = 1469598103934665603ull;
            auto mix = [&](const char* d, size_t len) { for (size_t k = 0; k < len; ++k) { h ^= (unsigned char)d[k]; h *= 1099511628211ull; } };
            const auto t0 = std::chrono::steady_clock::now();
            seqio::ParallelReader reader(mf.data(), mf.size(), opt.threads, opt.bytes(), 0);
            const size_t n_segments = reader.segments();
            while (std::unique_ptr<seqio::Segment> seg = reader.next()) {
                for (const seqio::Record& r : seg->batch.records) {
                    ++n; bases += r.seq_len;
                    if (!no_digest) { mix(r.id, r.id_len); mix("\t", 1); mix(r.seq, r.seq_len); mix("\n", 1); }
                }
                if (!seg->batch.error.empty()) { std::cerr << "ERROR: " << seg->batch.error << std::endl; return 1; }
            }
            const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::cout << "records=" << n << " bases=" << bases << " fnv=" << h << " seconds=" << secs
                      << " MB_per_s=" << (secs > 0 ? mf.size() / 1e6 / secs : 0.0) << " segments=" << n_segments << std::endl;
            return 0;
        }
        else if (!std::strcmp(argv[i], "--help") || !std::strcmp(argv[i], "-h")) {
            std::cout << "readbouncer_amd --config <file.toml> [--dump-config] [--batch-reads N] [--ingest-threads N] [--classify-threads N] [--segment-mb N] [--mmap-output] [--calibrate] [--placement-tries N] [--revcomp-of-n 3|4] "
                         "[--devices 0,1,...] [--parse-stats file]\n"
                         "readbouncer_amd --verify-ibf <file.ibf> --reference <file.fasta> [--fragment-size N]" << std::endl;
            return 0;
        }
    }
    if (!verify_path.empty()) {
        if (verify_ref.empty()) { std::cerr << "ERROR: --verify-ibf <file.ibf> needs --reference <file.fasta> [--fragment-size N]" << std::endl; return 1; }
        try {
            return verify_ibf(verify_path, verify_ref, verify_fragment);
        } catch (const std::exception& e) {
            std::cerr << "ERROR: " << e.what() << std::endl;
            return 1;
        }
    }
    if (config_path.empty()) {
        std::cerr << "ERROR: --config <file.toml> is required" << std::endl;
        return 1;
    }
    try {
        ConfigReader config(config_path);
        config.parse_general();
        if (dump_only) {
            config.parse();
            std::cout << config.dump();
            return 0;
        }
        // end-of-run report of main.cpp:438-444 (getrusage on Linux, main.cpp:140-152)
        const auto t_begin = std::chrono::steady_clock::now();
        const int rc = run_program(config, opt, devices);
        struct rusage ru;
        getrusage(RUSAGE_SELF, &ru);
        const double cpu = ru.ru_utime.tv_sec + ru.ru_stime.tv_sec + 1e-6 * (ru.ru_utime.tv_usec + ru.ru_stime.tv_usec);
        std::cout << "Real time : " << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() << " sec" << std::endl;
        std::cout << "CPU time  : " << cpu << " sec" << std::endl;
        std::cout << "Peak RSS  : " << (int)((ru.ru_maxrss * 1024L) / (1024 * 1024)) << " MByte" << std::endl;
        return rc;
    } catch (const ConfigReaderException& e) {
        std::cerr << "Error in reading TOML configuration file!" << std::endl << e.what() << std::endl;
        return 1;
    } catch (const std::exception& e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
        return 1;
    }
}

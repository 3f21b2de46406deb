// rb_main.cpp -- `readbouncer_amd --config file.toml`: the build and classify usages of ReadBouncer's CLI
// (src/main/main.cpp:274-406, src/main/parser.hpp:13-37, src/main/ibfbuild.hpp:21-180,
// src/main/classify.hpp:142-380) on top of the C++ mirror.  The chunk loop of classify_reads is run
// batch-wise: all reads of a batch are classified on chunk i in one GPU launch, reads that are still
// unclassified go on to chunk i+1 -- per read this is the reference's loop (classify.hpp:262-299).
// usage = "target" (live MinKNOW sampling) and "test" (connection test) are out of scope.
#include <sys/resource.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

#include "../../include/readbouncer_amd.hpp"
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
    seqio::Reader in(reference_file);
    if (!in.is_open()) throw interleave::FileParserException("Unable to open the file: " + reference_file);
    std::vector<interleave::RefSeq> records;
    std::string id, seq;
    try {
        while (in.read_record(id, seq)) records.push_back({id.substr(0, id.find(' ')), seq});
    } catch (const std::exception& e) {
        throw interleave::FileParserException("ERROR: Problems parsing the file: " + reference_file + "[" + e.what() + "]");
    }
    interleave::IBF filter{};
    const auto t0 = std::chrono::steady_clock::now();
    interleave::FilterStats stats = filter.create_filter(config, records);
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const uint64_t validSeqs = stats.totalSeqsFile - stats.invalidSeqs;
    std::cerr << "IBF-build processed " << validSeqs << " sequences (" << stats.sumSeqLen / 1000000.0 << " Mbp) in " << secs
              << " seconds" << std::endl;
    if (stats.invalidSeqs > 0) std::cerr << " - " << stats.invalidSeqs << " invalid sequences were skipped" << std::endl;
    std::cerr << " - " << validSeqs << " sequences in " << stats.totalBinsFile + stats.newBins
              << " bins were written to the IBF" << std::endl;
    return filter.getFilter();
}

// getIBF, src/main/ibfbuild.hpp:69-180: load the file if it is an IBF, else build one from the FASTA
static std::vector<interleave::IBFMeta> getIBF(ConfigReader config, bool depleteFilter, bool targetFilter)
{
    std::vector<interleave::IBFMeta> out;
    const std::vector<std::filesystem::path>& files =
        depleteFilter ? config.IBF_Parsed.deplete_files : (targetFilter ? config.IBF_Parsed.target_files : std::vector<std::filesystem::path>{});
    for (std::filesystem::path file : files) {
        interleave::IBFMeta filter{};
        filter.name = file.stem().string();
        if (config.filterException(file)) {
            interleave::IBF tf{};
            interleave::IBFConfig cfg{};
            cfg.input_filter_file = file.string();
            const auto t0 = std::chrono::steady_clock::now();
            interleave::FilterStats stats = tf.load_filter(cfg);
            filter.filter = tf.getFilter();
            std::cerr << stats.totalBinsFile << " bins were loaded in "
                      << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()
                      << " seconds from the IBF" << std::endl;
        } else {
            std::filesystem::path out_path = std::filesystem::path(config.output_dir);
            out_path /= file.filename();
            out_path.replace_extension("ibf");
            filter.filter = buildIBF(config, file.string(), out_path.string());
        }
        out.emplace_back(std::move(filter));
    }
    return out;
}

struct ReadState
{
    bool classified = false, failed = false;
    int best = -1;
};

// one parsed batch travelling from the reader thread to the classifying thread
struct ParsedBatch
{
    seqio::Batch batch;
    uint64_t n_records = 0;
};

// classify_reads, src/main/classify.hpp:142-380.  A reader thread parses batch i+1 from the memory-mapped read
// file while the GPU works on batch i.
static void classify_reads(ConfigReader& config, std::vector<interleave::IBFMeta> DepletionFilters,
                           std::vector<interleave::IBFMeta> TargetFilters, size_t batch_reads,
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

        std::vector<std::ofstream> targetFastas{};
        std::vector<std::vector<char>> outbufs(TargetFilters.size() + 1, std::vector<char>(1 << 20));
        for (size_t i = 0; i < TargetFilters.size(); ++i) {
            std::filesystem::path outfile(config.output_dir);
            outfile /= TargetFilters[i].name + ".fasta";
            targetFastas.emplace_back();
            targetFastas.back().rdbuf()->pubsetbuf(outbufs[i].data(), (std::streamsize)outbufs[i].size());
            targetFastas.back().open(outfile, std::ios::out);
        }
        std::filesystem::path outfile(config.output_dir);
        outfile /= "unclassified.fasta";
        std::ofstream UnclassifiedOut;
        UnclassifiedOut.rdbuf()->pubsetbuf(outbufs.back().data(), (std::streamsize)outbufs.back().size());
        UnclassifiedOut.open(outfile, std::ios::out);
        if (!UnclassifiedOut.is_open()) {
            std::cerr << "ERROR: Unable to open the file: " << outfile.string() << std::endl;
            return;
        }
        seqio::MappedFile mapped(read_file.string());
        if (!mapped.is_open()) {
            std::cerr << "ERROR: Unable to open the file: " << read_file.string() << std::endl;
            return;
        }
        std::cout << '\n' << "Classification results of: " << read_file.string() << '\n' << '\n';

        const uint32_t chunk_length = (uint32_t)config.IBF_Parsed.chunk_length;
        const uint32_t max_chunks = (uint8_t)config.IBF_Parsed.max_chunks;  // "uint8_t i" in the reference

        // ---- reader thread: two batches in flight
        std::mutex mu;
        std::condition_variable cv_full, cv_free;
        std::deque<std::unique_ptr<ParsedBatch>> ready;
        bool reader_done = false;
        std::thread reader([&] {
            seqio::Parser parser(mapped.data(), mapped.size());
            for (;;) {
                std::unique_ptr<ParsedBatch> pb(new ParsedBatch());
                parser.next_batch(pb->batch, batch_reads);
                const bool last = pb->batch.eof;
                {
                    std::unique_lock<std::mutex> lock(mu);
                    cv_free.wait(lock, [&] { return ready.size() < 2; });
                    ready.push_back(std::move(pb));
                }
                cv_full.notify_one();
                if (last) break;
            }
            {
                std::lock_guard<std::mutex> lock(mu);
                reader_done = true;
            }
            cv_full.notify_one();
        });

        std::vector<char> flat;
        std::vector<uint64_t> offs;
        std::vector<uint32_t> lens;
        for (;;) {
            std::unique_ptr<ParsedBatch> pb;
            {
                std::unique_lock<std::mutex> lock(mu);
                cv_full.wait(lock, [&] { return !ready.empty() || reader_done; });
                if (ready.empty()) break;
                pb = std::move(ready.front());
                ready.pop_front();
            }
            cv_free.notify_one();
            const std::vector<seqio::Record>& recs = pb->batch.records;
            readCounter += recs.size();
            if (!pb->batch.error.empty()) std::cerr << "ERROR: " << pb->batch.error << std::endl;
            std::vector<ReadState> state(recs.size());
            std::vector<size_t> active;
            for (size_t i = 0; i < recs.size(); ++i) {
                if (recs[i].seq_len < chunk_length) too_short++;  // classify.hpp:247-250
                else active.push_back(i);
            }
            const size_t n_candidates = active.size();
            const auto t0 = std::chrono::steady_clock::now();
            for (uint32_t c = 0; c < max_chunks && !active.empty(); ++c) {
                flat.clear();
                offs.clear();
                lens.clear();
                std::vector<size_t> idx;
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
                std::vector<size_t> next;
                if (!idx.empty()) {
                    if (flat.empty()) flat.push_back('N');
                    interleave::BatchResult res =
                        multi ? multi->classify_flat(Conf, flat.data(), offs.data(), lens.data(), idx.size(), RB_MODE_CLASSIFY_CHUNK)
                              : interleave::classify_batch_flat(DepletionFilters, TargetFilters, Conf, flat.data(), offs.data(),
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
            classify_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            classify_reads_n += n_candidates;
            for (size_t i = 0; i < recs.size(); ++i) {  // outputs in read order
                const seqio::Record& r = recs[i];
                if (r.seq_len < chunk_length) continue;
                if (state[i].failed) {  // classify.hpp:306-316
                    failed++;
                    log_line("error", "Error classifying Read : " + std::string(r.id, r.id_len) + "(Len=" + std::to_string(r.seq_len) + ")");
                    continue;
                }
                if (state[i].classified) {
                    found++;
                    if (target && state[i].best >= 0) {
                        TargetFilters[state[i].best].classified += 1;
                        seqio::write_fasta(targetFastas[state[i].best], r.id, r.id_len, r.seq, r.seq_len);
                    }
                } else {
                    seqio::write_fasta(UnclassifiedOut, r.id, r.id_len, r.seq, r.seq_len);
                }
            }
        }
        reader.join();
        for (auto& f : targetFastas) f.close();
        UnclassifiedOut.close();
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
                  << " classify_s=" << classify_seconds << std::endl;
        ClassificationResults_.found = found;
        ClassificationResults_.failed = failed;
        ClassificationResults_.too_short = too_short;
        ClassificationResults_.readCounter = readCounter;
        for (interleave::IBFMeta& f : TargetFilters) f.classified = 0;
    }
}

static int run_program(ConfigReader& config, size_t batch_reads, const std::vector<int>& devices)
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
        std::vector<interleave::IBFMeta> DepletionFilters = getIBF(config, true, false);
        std::vector<interleave::IBFMeta> TargetFilters = getIBF(config, false, true);
        classify_reads(config, DepletionFilters, TargetFilters, batch_reads, devices);
        return 0;
    }
    std::cerr << "usage \"" << config.usage << "\" is outside this engine's scope (supported: build, classify)" << std::endl;
    return 2;
}

int main(int argc, char const* argv[])
{
    std::string config_path;
    bool dump_only = false;
    size_t batch_reads = 65536;
    std::vector<int> devices{0};
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
        else if (!std::strcmp(argv[i], "--batch-reads") && i + 1 < argc) batch_reads = (size_t)std::stoull(argv[++i]);
        else if (!std::strcmp(argv[i], "--parse-stats") && i + 1 < argc) {
            // ingest self-check (no GPU): records, bases and an FNV-1a digest over "id\tseq\n" of every record
            seqio::MappedFile mf(argv[++i]);
            if (!mf.is_open()) { std::cerr << "ERROR: Unable to open the file: " << argv[i] << std::endl; return 1; }
            seqio::Parser parser(mf.data(), mf.size());
            seqio::Batch b;
            uint64_t n = 0, bases = 0, h = 1469598103934665603ull;
            auto mix = [&](const char* d, size_t len) { for (size_t k = 0; k < len; ++k) { h ^= (unsigned char)d[k]; h *= 1099511628211ull; } };
            const auto t0 = std::chrono::steady_clock::now();
            do {
                parser.next_batch(b, batch_reads);
                for (const seqio::Record& r : b.records) {
                    ++n; bases += r.seq_len;
                    mix(r.id, r.id_len); mix("\t", 1); mix(r.seq, r.seq_len); mix("\n", 1);
                }
                if (!b.error.empty()) { std::cerr << "ERROR: " << b.error << std::endl; return 1; }
            } while (!b.eof);
            const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::cout << "records=" << n << " bases=" << bases << " fnv=" << h << " seconds=" << secs
                      << " MB_per_s=" << (secs > 0 ? mf.size() / 1e6 / secs : 0.0) << std::endl;
            return 0;
        }
        else if (!std::strcmp(argv[i], "--help") || !std::strcmp(argv[i], "-h")) {
            std::cout << "readbouncer_amd --config <file.toml> [--dump-config] [--batch-reads N] [--devices 0,1,...] [--parse-stats file]" << std::endl;
            return 0;
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
        const int rc = run_program(config, batch_reads, devices);
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

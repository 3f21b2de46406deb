// rb_main.cpp -- `readbouncer_amd --config file.toml`: the build and classify usages of ReadBouncer's CLI
// (src/main/main.cpp:274-406, src/main/parser.hpp:13-37, src/main/ibfbuild.hpp:21-180,
// src/main/classify.hpp:142-380) on top of the C++ mirror.  The chunk loop of classify_reads is run
// batch-wise: all reads of a batch are classified on chunk i in one GPU launch, reads that are still
// unclassified go on to chunk i+1 -- per read this is the reference's loop (classify.hpp:262-299).
// usage = "target" (live MinKNOW sampling) and "test" (connection test) are out of scope.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

#include "../../include/readbouncer_amd.hpp"
#include "config_reader.hpp"
#include "seqio.hpp"

bool ConfigReader::filterException(std::filesystem::path& file) { return rb_is_ibf_file(file.string().c_str()) == 1; }

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

struct PendingRead
{
    std::string id, seq;
    bool classified = false, failed = false;
    int best = -1;
};

// classify_reads, src/main/classify.hpp:142-380
static void classify_reads(ConfigReader& config, std::vector<interleave::IBFMeta> DepletionFilters,
                           std::vector<interleave::IBFMeta> TargetFilters, size_t batch_reads)
{
    interleave::ClassifyConfig Conf{};
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

        std::vector<std::ofstream> targetFastas{};
        for (interleave::IBFMeta& f : TargetFilters) {
            std::filesystem::path outfile(config.output_dir);
            outfile /= f.name + ".fasta";
            targetFastas.emplace_back(outfile, std::ios::out);
        }
        std::filesystem::path outfile(config.output_dir);
        outfile /= "unclassified.fasta";
        std::ofstream UnclassifiedOut(outfile, std::ios::out);
        if (!UnclassifiedOut.is_open()) {
            std::cerr << "ERROR: Unable to open the file: " << outfile.string() << std::endl;
            return;
        }
        seqio::Reader seqFileIn(read_file.string());
        if (!seqFileIn.is_open()) {
            std::cerr << "ERROR: Unable to open the file: " << read_file.string() << std::endl;
            return;
        }
        std::cout << '\n' << "Classification results of: " << read_file.string() << '\n' << '\n';

        const uint32_t chunk_length = (uint32_t)config.IBF_Parsed.chunk_length;
        const uint32_t max_chunks = (uint8_t)config.IBF_Parsed.max_chunks;  // "uint8_t i" in the reference
        bool eof = false;
        while (!eof) {
            std::vector<PendingRead> batch;
            while (batch.size() < batch_reads) {
                PendingRead r;
                try {
                    if (!seqFileIn.read_record(r.id, r.seq)) { eof = true; break; }
                    readCounter++;
                } catch (const std::exception& e) {
                    std::cerr << "ERROR: " << e.what() << " [@" << r.id << "]" << std::endl;
                    eof = true;
                    break;
                }
                if (r.seq.size() < chunk_length) { too_short++; continue; }  // classify.hpp:247-250
                batch.push_back(std::move(r));
            }
            if (batch.empty()) continue;
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<size_t> active(batch.size());
            for (size_t i = 0; i < batch.size(); ++i) active[i] = i;
            for (uint32_t c = 0; c < max_chunks && !active.empty(); ++c) {
                std::vector<std::string> frags;
                std::vector<size_t> idx;
                for (size_t i : active) {
                    PendingRead& r = batch[i];
                    uint64_t fragend = (uint64_t)(c + 1) * chunk_length, fragstart = (uint64_t)c * chunk_length;
                    if (fragend > r.seq.size()) fragend = r.seq.size();
                    if (fragstart > fragend) { r.failed = true; continue; }  // undefined infix in the reference
                    frags.push_back(r.seq.substr(fragstart, fragend - fragstart));
                    idx.push_back(i);
                }
                std::vector<size_t> next;
                if (!frags.empty()) {
                    interleave::BatchResult res =
                        interleave::classify_batch(DepletionFilters, TargetFilters, Conf, frags, RB_MODE_CLASSIFY_CHUNK);
                    for (size_t j = 0; j < idx.size(); ++j) {
                        PendingRead& r = batch[idx[j]];
                        if (res.status[j] != RB_OK) { r.failed = true; continue; }  // exception -> failed++ (:306-316)
                        if (res.decision[j]) {
                            r.classified = true;
                            r.best = target ? res.best_target[j] : -1;
                        } else {
                            next.push_back(idx[j]);
                        }
                    }
                }
                active.swap(next);
            }
            classify_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            classify_reads_n += batch.size();
            for (PendingRead& r : batch) {  // outputs in read order
                if (r.failed) { failed++; continue; }
                if (r.classified) {
                    found++;
                    if (target && r.best >= 0) {
                        TargetFilters[r.best].classified += 1;
                        seqio::write_fasta(targetFastas[r.best], r.id, r.seq);
                    }
                } else {
                    seqio::write_fasta(UnclassifiedOut, r.id, r.seq);
                }
            }
        }
        for (auto& f : targetFastas) f.close();
        UnclassifiedOut.close();
        const double avg = classify_reads_n ? classify_seconds / (double)classify_reads_n : 0.0;
        std::cout << "------------------------------- Final Results -------------------------------" << std::endl;
        std::cout << "Number of classified reads                         :   " << found << std::endl;
        std::cout << "Number of of too short reads (len < " << config.IBF_Parsed.chunk_length << ")           :   " << too_short << std::endl;
        std::cout << "Number of all reads                                :   " << readCounter << std::endl;
        for (interleave::IBFMeta& f : TargetFilters)
            std::cout << f.name << "\t : " << f.classified << "\t\t" << ((float)f.classified) / ((float)readCounter) << std::endl;
        std::cout << "Average Processing Time Read Classification        :   " << avg << std::endl;
        std::cout << "-----------------------------------------------------------------------------------" << std::endl;
        std::cout << "RESULT found=" << found << " failed=" << failed << " too_short=" << too_short
                  << " readCounter=" << readCounter << std::endl;
        ClassificationResults_.found = found;
        ClassificationResults_.failed = failed;
        ClassificationResults_.too_short = too_short;
        ClassificationResults_.readCounter = readCounter;
        for (interleave::IBFMeta& f : TargetFilters) f.classified = 0;
    }
}

static int run_program(ConfigReader& config, size_t batch_reads)
{
    config.parse();
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
        classify_reads(config, DepletionFilters, TargetFilters, batch_reads);
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
    for (int i = 1; i < argc; ++i) {
        if ((!std::strcmp(argv[i], "--config") || !std::strcmp(argv[i], "-c")) && i + 1 < argc) config_path = argv[++i];
        else if (!std::strcmp(argv[i], "--dump-config")) dump_only = true;
        else if (!std::strcmp(argv[i], "--batch-reads") && i + 1 < argc) batch_reads = (size_t)std::stoull(argv[++i]);
        else if (!std::strcmp(argv[i], "--help") || !std::strcmp(argv[i], "-h")) {
            std::cout << "readbouncer_amd --config <file.toml> [--dump-config] [--batch-reads N]" << std::endl;
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
        return run_program(config, batch_reads);
    } catch (const ConfigReaderException& e) {
        std::cerr << "Error in reading TOML configuration file!" << std::endl << e.what() << std::endl;
        return 1;
    } catch (const std::exception& e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
        return 1;
    }
}

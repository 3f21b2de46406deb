// readbouncer_amd.hpp -- C++ mirror of ReadBouncer's classify interface over the C ABI
// (include/readbouncer_amd.h).  Same names, argument meaning and error behaviour as the reference:
//
//   interleave::IBF::load_filter / create_filter   src/IBF/IBF.hpp:137-138, src/IBF/IBFBuild.cpp:329,421
//   interleave::IBFMeta                            src/IBF/IBF.hpp:161-167
//   interleave::ClassifyConfig / IBFConfig         src/IBF/IBFConfig.hpp:23-46, 48-145
//   interleave::Read::classify x3                  src/IBF/IBF.hpp:211-213, src/IBF/IBFClassify.cpp:181,239,299
//   check_unblock                                  src/main/adaptive_sampling.hpp:35-113
//   exception types                                src/IBF/IBFExceptions.hpp
//
// Differences that are deliberate: TIbf is a handle to a filter resident in GPU HBM (no SeqAn
// type); sequences are std::string (the Dna5 conversion happens on the device); there is a batch
// classifier next to the per-read calls, because one launch per read wastes the GPU.
// Header-only; link with -lreadbouncer_amd.
#pragma once
#include <cstdint>
#include <exception>
#include <stdexcept>
#include <map>
#include <chrono>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "readbouncer_amd.h"

namespace interleave
{

// ---- exceptions (src/IBF/IBFExceptions.hpp) -------------------------------------------------
class IBFBuildException : public std::exception
{
    std::string msg_;
public:
    explicit IBFBuildException(const std::string& m = "") : msg_(m) {}
    const char* what() const noexcept override { return msg_.c_str(); }
};
#define RB_DEFINE_EXC(Name, Base)                                         \
    class Name : public Base                                              \
    {                                                                     \
    public:                                                               \
        explicit Name(const std::string& m = "") : Base(m) {}             \
    };
RB_DEFINE_EXC(NullFilterException, IBFBuildException)
RB_DEFINE_EXC(ShortReadException, IBFBuildException)
RB_DEFINE_EXC(CountKmerException, IBFBuildException)
RB_DEFINE_EXC(ParseIBFFileException, IBFBuildException)
RB_DEFINE_EXC(MissingIBFFileException, IBFBuildException)
RB_DEFINE_EXC(StoreFilterException, IBFBuildException)
RB_DEFINE_EXC(InvalidConfigException, IBFBuildException)
RB_DEFINE_EXC(MissingReferenceFilesException, IBFBuildException)
RB_DEFINE_EXC(FileParserException, IBFBuildException)
RB_DEFINE_EXC(InsertSequenceException, IBFBuildException)
RB_DEFINE_EXC(DeviceException, IBFBuildException)  // no GPU / HIP failure: no CPU fallback exists
#undef RB_DEFINE_EXC

inline void throw_status(int st, const std::string& ctx = "")
{
    if (st == RB_OK) return;
    const std::string m = ctx + (ctx.empty() ? "" : ": ") + rb_status_string(st) + " [" + rb_last_error() + "]";
    switch (st) {
    case RB_ERR_NULL_FILTER: throw NullFilterException("No IBF provided to classify the read!");
    case RB_ERR_SHORT_READ: throw ShortReadException(m);
    case RB_ERR_COUNT_KMER: throw CountKmerException(m);
    case RB_ERR_PARSE_IBF: throw ParseIBFFileException(m);
    case RB_ERR_MISSING_FILE: throw MissingIBFFileException(m);
    case RB_ERR_STORE: throw StoreFilterException(m);
    case RB_ERR_INVALID_ARG: throw InvalidConfigException(m);
    default: throw DeviceException(m);
    }
}

// ---- configs (src/IBF/IBFConfig.hpp) ----------------------------------------------------------
class ClassifyConfig
{
public:
    double significance = 0.95;
    double error_rate = 0.1;
    uint16_t max_error = 0;
    uint16_t strata_filter = 0;
};

class IBFConfig
{
public:
    static constexpr uint32_t MBinBits = 8388608;
    std::vector<std::string> reference_files;
    std::string output_filter_file = "";
    std::string input_filter_file = "";
    std::string update_filter_file = "";
    uint64_t filter_size = 0;
    uint64_t filter_size_bits = 0;
    uint64_t fragment_length = 0;
    uint16_t overlap_length = 1500;
    uint16_t kmer_size = 13;
    uint16_t hash_functions = 3;
    uint16_t threads = 2;
    uint32_t n_refs = 400;
    uint32_t n_batches = 500000;
    double max_fp = 0.01;
    bool verbose = false;
    bool quiet = false;
    uint16_t threads_build = 1;
    int device = 0;  // GPU that will hold the filter (not in the reference)

    bool validate()  // src/IBF/IBFConfig.hpp:96-144
    {
        threads_build = threads <= 2 ? 1 : threads - 1;
        if (n_batches < 1) n_batches = 1;
        if (n_refs < 1) n_refs = 1;
        if (!update_filter_file.empty()) {
            kmer_size = 0; hash_functions = 0; filter_size = 0; filter_size_bits = 0;
        } else if (filter_size_bits != 0) {
            filter_size = filter_size_bits / MBinBits;
        } else if (filter_size != 0) {
            filter_size_bits = filter_size * MBinBits;
        }
        return true;
    }
};

struct FilterStats  // src/IBF/IBF.hpp:51-79; the reference's StopClock members as seconds
{
    uint64_t sumSeqLen = 0;
    uint64_t totalSeqsBinId = 0;
    uint32_t totalBinsBinId = 0;
    uint64_t totalSeqsFile = 0;
    uint32_t totalBinsFile = 0;
    uint64_t invalidSeqs = 0;
    uint32_t newBins = 0;
    // timeLoadSeq (cutOutNNNs + sizing, IBFBuild.cpp:441-446), timeBuild (:462-497) and inside it the filter's allocation in HBM
    // (with the placement trial of tables >= 1 GiB), the concatenation of the cleaned records and rb_dibf_insert (H2D + the insert
    // kernel), timeSaveFilter (:503-515: download + file), timeIBF (:433-517)
    double timeLoadSeq = 0.0, timeBuild = 0.0, timeSaveFilter = 0.0, timeIBF = 0.0;
    double timeAllocFilter = 0.0, timeConcat = 0.0, timeInsert = 0.0;
};

// ---- TIbf: the filter, resident in HBM ---------------------------------------------------------
class TIbf
{
    std::shared_ptr<rb_dibf> h_;
public:
    uint64_t noOfBins = 0;      // read at IBFClassify.cpp:27,58
    uint64_t kmerSize = 0;      // read at IBFClassify.cpp:102,154,192,248
    uint64_t noOfHashFunc = 0;
    uint64_t noOfBits = 0;
    TIbf() = default;
    explicit TIbf(rb_dibf* raw) : h_(raw, [](rb_dibf* p) { rb_dibf_free(p); })
    {
        rb_ibf_info i;
        throw_status(rb_dibf_get_info(raw, &i), "TIbf");
        noOfBins = i.n_bins; kmerSize = i.kmer_size; noOfHashFunc = i.n_hash; noOfBits = i.n_bits;
    }
    // TIbf(bins, hash_functions, kmer_size, filter_size_bits)  src/IBF/IBFBuild.cpp:465
    TIbf(uint64_t bins, uint64_t hash_functions, uint64_t kmer_size, uint64_t bits, int device = 0)
    {
        rb_dibf* raw = nullptr;
        throw_status(rb_dibf_create(device, bins, hash_functions, kmer_size, bits, &raw), "TIbf");
        *this = TIbf(raw);
    }
    rb_dibf* handle() const { return h_.get(); }
    bool empty() const { return !h_; }
    void store(const std::string& path) const  // seqan::store
    {
        rb_ibf* host = nullptr;
        throw_status(rb_dibf_download(h_.get(), &host), "store");
        const int st = rb_ibf_store(host, path.c_str());
        rb_ibf_close(host);
        throw_status(st, "store");
    }
};
inline uint64_t getNumberOfBins(const TIbf& f) { return f.noOfBins; }
inline uint64_t getKmerSize(const TIbf& f) { return f.kmerSize; }

struct IBFMeta  // src/IBF/IBF.hpp:161-167
{
    TIbf filter;
    std::string name;
    uint64_t classified = 0;
};

// ---- IBF: load / create (src/IBF/IBFBuild.cpp) ---------------------------------------------------
struct RefSeq
{
    std::string seqid;
    std::string seq;
};

class IBF
{
    TIbf filter{};

    // add_sequences_to_filter (IBFBuild.cpp:143-215) for every queued sequence in ONE device call: the fragments
    // of all sequences (reference fragmenter, bins numbered consecutively across sequences) go into one table, the
    // sequences into one buffer -- a reference of 100 000 contigs costs one launch, not 100 000
    static double seconds_since(std::chrono::steady_clock::time_point t0)
    {
        return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    void insert_all(const std::vector<std::string>& cleaned, const IBFConfig& config, uint64_t first_bin, FilterStats* stats = nullptr)
    {
        const auto t_concat = std::chrono::steady_clock::now();
        std::string all;
        size_t total = 0;
        for (const std::string& c : cleaned) total += c.size();
        all.reserve(total);
        std::vector<uint64_t> starts, ends, bins, s, e;
        uint64_t binid = first_bin;
        for (const std::string& c : cleaned) {
            const size_t n = rb_fragment_bounds(c.size(), config.fragment_length, config.kmer_size, config.overlap_length,
                                                nullptr, nullptr, 0);
            s.resize(n);
            e.resize(n);
            rb_fragment_bounds(c.size(), config.fragment_length, config.kmer_size, config.overlap_length, s.data(), e.data(), n);
            for (size_t i = 0; i < n; ++i) {
                starts.push_back(all.size() + s[i]);
                ends.push_back(all.size() + e[i]);
                bins.push_back(binid++);
            }
            all += c;
        }
        if (stats) stats->timeConcat = seconds_since(t_concat);
        const auto t_insert = std::chrono::steady_clock::now();
        const int st = rb_dibf_insert(filter.handle(), all.data(), all.size(), starts.data(), ends.data(), bins.data(),
                                      starts.size());
        if (stats) stats->timeInsert = seconds_since(t_insert);
        if (st != RB_OK) throw InsertSequenceException(std::string("Error inserting the sequences to the IBF: ") + rb_last_error());
    }

public:
    // IBF::load_filter, IBFBuild.cpp:329-396: input_filter_file or update_filter_file must be set
    FilterStats load_filter(IBFConfig& config)
    {
        FilterStats stats;
        const std::string& path = !config.update_filter_file.empty() ? config.update_filter_file : config.input_filter_file;
        if (path.empty())
            throw MissingIBFFileException("Error: Either update_filter_file or input_filter_file have to be specified.");
        rb_dibf* raw = nullptr;
        const int st = rb_dibf_open(config.device, path.c_str(), &raw);
        if (st == RB_ERR_PARSE_IBF || st == RB_ERR_MISSING_FILE)
            throw ParseIBFFileException("Error parsing IBF input file " + path + ": " + rb_last_error());
        throw_status(st, "load_filter");
        filter = TIbf(raw);
        stats.totalBinsFile = (uint32_t)getNumberOfBins(filter);
        config.kmer_size = (uint16_t)getKmerSize(filter);
        return stats;
    }

    // IBF::create_filter, IBFBuild.cpp:421-521, from already parsed reference records
    // (parse_ref_seqs' per-record logic, :66-92, is applied here; file parsing is the caller's)
    FilterStats create_filter(IBFConfig& config, const std::vector<RefSeq>& records)
    {
        if (!config.validate()) throw InvalidConfigException("Config not valid!");
        if (records.empty() && config.reference_files.empty())
            throw MissingReferenceFilesException("There were no reference files specified!");
        FilterStats stats;
        const auto t_all = std::chrono::steady_clock::now();
        std::vector<std::string> cleaned;
        for (const RefSeq& r : records) {
            stats.totalSeqsFile += 1;
            if (r.seq.size() < config.kmer_size) { stats.invalidSeqs += 1; continue; }
            std::string c(r.seq.size(), '\0');
            c.resize(rb_cut_out_nnns(r.seq.data(), r.seq.size(), &c[0]));
            stats.totalBinsBinId += (uint32_t)(c.size() / config.fragment_length) + 1;
            stats.sumSeqLen += c.size();
            cleaned.push_back(std::move(c));
        }
        config.filter_size_bits = rb_calculate_filter_size_bits(config.fragment_length, config.kmer_size,
                                                                config.hash_functions, config.max_fp, stats.totalBinsBinId);
        stats.timeLoadSeq = seconds_since(t_all);
        const auto t_build = std::chrono::steady_clock::now();
        try {
            filter = TIbf(stats.totalBinsBinId, config.hash_functions, config.kmer_size, config.filter_size_bits, config.device);
        } catch (const InvalidConfigException&) {
            throw NullFilterException("Could not instantiate IBF Filter");
        }
        stats.timeAllocFilter = seconds_since(t_build);
        stats.totalBinsFile = (uint32_t)getNumberOfBins(filter);
        insert_all(cleaned, config, 0, &stats);
        stats.timeBuild = seconds_since(t_build);
        const auto t_save = std::chrono::steady_clock::now();
        if (!config.output_filter_file.empty()) filter.store(config.output_filter_file);
        stats.timeSaveFilter = seconds_since(t_save);
        stats.timeIBF = seconds_since(t_all);
        return stats;
    }

    // Not in the reference: first-contact check of a filter file against the reference sequences it was built from.
    // Loads input_filter_file, re-runs parse_ref_seqs' per-record logic + the fragmenter + insertKmer (IBFBuild.cpp:66-92,
    // 165-204, 190) over `records` into an EMPTY filter of the file's geometry and compares (rb_dibf_compare).  With the
    // right layout/hash constants report.new_bits == 0.  bins_expected = the bin count create_filter would have chosen.
    struct VerifyReport
    {
        rb_ibf_compare bits{};
        uint64_t bins_file = 0, bins_expected = 0;
        std::string warning;  // rb_last_warning() of the load
        bool ok() const { return bits.new_bits == 0 && bins_file == bins_expected && bits.rebuilt_bits > 0; }
    };
    VerifyReport verify_filter(IBFConfig& config, const std::vector<RefSeq>& records)
    {
        VerifyReport rep;
        load_filter(config);  // sets config.kmer_size
        rep.warning = rb_last_warning();
        TIbf file = filter;
        std::vector<std::string> cleaned;
        for (const RefSeq& r : records) {
            if (r.seq.size() < config.kmer_size) continue;
            std::string c(r.seq.size(), '\0');
            c.resize(rb_cut_out_nnns(r.seq.data(), r.seq.size(), &c[0]));
            rep.bins_expected += (uint32_t)(c.size() / config.fragment_length) + 1;
            cleaned.push_back(std::move(c));
        }
        rep.bins_file = file.noOfBins;
        // bins beyond the file's count cannot be inserted; such fragments are the mismatch bins_expected already reports
        filter = TIbf(file.noOfBins, file.noOfHashFunc, file.kmerSize, file.noOfBits, config.device);
        if (rep.bins_expected <= rep.bins_file) insert_all(cleaned, config, 0);
        throw_status(rb_dibf_compare(file.handle(), filter.handle(), &rep.bits), "rb_dibf_compare");
        filter = file;
        return rep;
    }

    // IBF::update_filter, IBFBuild.cpp:223-321: load update_filter_file, resizeBins(old + new), add the new
    // sequences starting at bin id totalBinsFile, store back to update_filter_file
    FilterStats update_filter(IBFConfig& config, const std::vector<RefSeq>& records)
    {
        if (!config.validate()) throw InvalidConfigException("Config not valid!");
        const std::string path = config.update_filter_file;
        FilterStats stats = load_filter(config);  // sets config.kmer_size from the file
        std::vector<std::string> cleaned;
        for (const RefSeq& r : records) {
            stats.totalSeqsFile += 1;
            if (r.seq.size() < config.kmer_size) { stats.invalidSeqs += 1; continue; }
            std::string c(r.seq.size(), '\0');
            c.resize(rb_cut_out_nnns(r.seq.data(), r.seq.size(), &c[0]));
            stats.totalBinsBinId += (uint32_t)(c.size() / config.fragment_length) + 1;
            stats.sumSeqLen += c.size();
            cleaned.push_back(std::move(c));
        }
        const uint32_t number_new_bins = stats.totalBinsBinId + stats.totalBinsFile;
        if (number_new_bins > stats.totalBinsFile) {
            rb_dibf* wider = nullptr;
            throw_status(rb_dibf_resize_bins(filter.handle(), number_new_bins, &wider), "resizeBins");
            filter = TIbf(wider);
            stats.newBins = stats.totalBinsBinId;
            stats.totalBinsBinId = number_new_bins;
        }
        insert_all(cleaned, config, stats.totalBinsFile);
        try {
            filter.store(path);
        } catch (const IBFBuildException& e) {
            throw StoreFilterException("Could not store IBF to " + path + ":" + e.what());
        }
        return stats;
    }

    inline TIbf getFilter() { return filter; }
};

// ---- engines are cached per (deplete set, target set) -----------------------------------------
// What the reverse strand holds for an N of the read, for every engine this process creates from now on (a recalled SeqAn fact,
// readbouncer_amd.h: rb_engine_set_revcomp_of_n): 3 = T, the default; 4 = N.  Call it before the first classify.
inline void set_revcomp_of_n(uint32_t ordinal) { throw_status(rb_set_default_revcomp_of_n(ordinal), "set_revcomp_of_n"); }

namespace detail
{
// One engine per (calling thread, filter set).  Engines only borrow the filters (no copy in HBM) and own their streams
// and workspaces, so the N classification threads of the reference (src/main/adaptive_sampling.hpp:745-751) overlap on
// the GPU instead of queueing behind one engine's staging buffers.  The key carries the geometry next to the handles: a
// filter freed and another one allocated at the same address must not inherit threshold tables made for another k.
struct EngineKey
{
    std::vector<rb_dibf*> d, t;
    std::vector<uint64_t> geo;
    bool operator<(const EngineKey& o) const { return d != o.d ? d < o.d : (t != o.t ? t < o.t : geo < o.geo); }
};
inline rb_engine* engine_for(const std::vector<IBFMeta>& dep, const std::vector<IBFMeta>& tgt)
{
    thread_local std::map<EngineKey, std::shared_ptr<rb_engine>> cache;
    EngineKey k;
    for (const IBFMeta& m : dep) k.d.push_back(m.filter.handle());
    for (const IBFMeta& m : tgt) k.t.push_back(m.filter.handle());
    for (const std::vector<rb_dibf*>* set : {&k.d, &k.t})
        for (rb_dibf* f : *set) {
            rb_ibf_info g{};
            if (f) rb_dibf_get_info(f, &g);
            k.geo.insert(k.geo.end(), {g.n_bins, g.n_hash, g.kmer_size, g.n_bits});
        }
    auto it = cache.find(k);
    if (it != cache.end()) return it->second.get();
    if (cache.size() >= 16) cache.clear();  // filter sets come and go (tests, rebuilt filters): do not hoard engines
    rb_engine* e = nullptr;
    const int dev = !k.d.empty() ? rb_dibf_device(k.d[0]) : (!k.t.empty() ? rb_dibf_device(k.t[0]) : 0);
    throw_status(rb_engine_create(dev, k.d.data(), k.d.size(), k.t.data(), k.t.size(), &e), "engine");
    cache[k] = std::shared_ptr<rb_engine>(e, [](rb_engine* p) { rb_engine_destroy(p); });
    return e;
}
}  // namespace detail

// ---- batch form (one launch for many reads) ----------------------------------------------------
struct BatchResult
{
    std::vector<uint16_t> maxcount;  // [n_reads x (n_deplete + n_target)], deplete first
    std::vector<int32_t> best_target;
    std::vector<uint8_t> decision;
    std::vector<uint8_t> status;
};

// flat form: read i = seqs[offsets[i] .. offsets[i]+lens[i])
inline BatchResult classify_batch_flat(const std::vector<IBFMeta>& DepletionFilters, const std::vector<IBFMeta>& TargetFilters,
                                       const ClassifyConfig& conf, const char* seqs, const uint64_t* offsets,
                                       const uint32_t* lens, size_t n, int mode)
{
    if (DepletionFilters.empty() && TargetFilters.empty()) throw NullFilterException("No IBF provided to classify the read!");
    rb_engine* e = detail::engine_for(DepletionFilters, TargetFilters);
    const size_t nf = DepletionFilters.size() + TargetFilters.size();
    BatchResult r;
    r.maxcount.resize(n * nf);
    r.best_target.resize(n);
    r.decision.resize(n);
    r.status.resize(n);
    throw_status(rb_classify_batch(e, seqs, offsets, lens, n, conf.error_rate, conf.significance, mode, r.maxcount.data(),
                                   r.best_target.data(), r.decision.data(), r.status.data()),
                 "classify_batch");
    return r;
}

inline BatchResult classify_batch(const std::vector<IBFMeta>& DepletionFilters, const std::vector<IBFMeta>& TargetFilters,
                                  const ClassifyConfig& conf, const std::vector<std::string>& seqs, int mode)
{
    const size_t n = seqs.size();
    std::string flat;
    std::vector<uint64_t> offs(n);
    std::vector<uint32_t> lens(n);
    size_t total = 0;
    for (const std::string& s : seqs) total += s.size();
    flat.reserve(total + 1);
    for (size_t i = 0; i < n; ++i) {
        offs[i] = flat.size();
        lens[i] = (uint32_t)seqs[i].size();
        flat += seqs[i];
    }
    if (flat.empty()) flat.push_back('N');
    return classify_batch_flat(DepletionFilters, TargetFilters, conf, flat.data(), offs.data(), lens.data(), n, mode);
}

// ---- several GPUs in one process: filters replicated, batches read-sharded (rb_pool) --------------
class MultiDeviceClassifier
{
    rb_pool* pool_ = nullptr;
    size_t nf_ = 0;
public:
    MultiDeviceClassifier(const std::vector<int>& devices, const std::vector<IBFMeta>& DepletionFilters,
                          const std::vector<IBFMeta>& TargetFilters)
    {
        if (DepletionFilters.empty() && TargetFilters.empty()) throw NullFilterException("No IBF provided to classify the read!");
        std::vector<rb_ibf*> images;
        auto cleanup = [&] { for (rb_ibf* i : images) rb_ibf_close(i); };
        for (const std::vector<IBFMeta>* set : {&DepletionFilters, &TargetFilters})
            for (const IBFMeta& m : *set) {
                rb_ibf* img = nullptr;
                const int st = rb_dibf_download(m.filter.handle(), &img);
                if (st != RB_OK) { cleanup(); throw_status(st, "download"); }
                images.push_back(img);
            }
        nf_ = images.size();
        const int st = rb_pool_create(devices.data(), devices.size(), images.data(), DepletionFilters.size(),
                                      images.data() + DepletionFilters.size(), TargetFilters.size(), &pool_);
        cleanup();  // the pool holds its own replicas in HBM
        throw_status(st, "rb_pool_create");
    }
    ~MultiDeviceClassifier() { rb_pool_destroy(pool_); }
    MultiDeviceClassifier(const MultiDeviceClassifier&) = delete;
    MultiDeviceClassifier& operator=(const MultiDeviceClassifier&) = delete;

    BatchResult classify_flat(const ClassifyConfig& conf, const char* seqs, const uint64_t* offsets, const uint32_t* lens,
                              size_t n, int mode)
    {
        BatchResult r;
        r.maxcount.resize(n * nf_);
        r.best_target.resize(n);
        r.decision.resize(n);
        r.status.resize(n);
        throw_status(rb_pool_classify_batch(pool_, seqs, offsets, lens, n, conf.error_rate, conf.significance, mode,
                                            r.maxcount.data(), r.best_target.data(), r.decision.data(), r.status.data()),
                     "rb_pool_classify_batch");
        return r;
    }
};

// ---- Read (src/IBF/IBF.hpp:169-226) -------------------------------------------------------------
class Read
{
public:
    std::string sequence{};
    std::string id{};

    Read() {}
    Read(const std::string& id_, const std::string& seq) : sequence(seq), id(id_) {}
    ~Read() {}

    inline uint32_t getReadLength() { return (uint32_t)sequence.size(); }

    // IBFClassify.cpp:181-226: true if any filter holds a bin at or above the threshold
    bool classify(std::vector<TIbf>& filters, ClassifyConfig& config)
    {
        if (filters.empty()) throw NullFilterException("No IBF provided to classify the read!");
        std::vector<IBFMeta> metas;
        for (TIbf& f : filters) metas.push_back(IBFMeta{f, "", 0});
        // select_matches semantics (any bin >= threshold, so a threshold of 0 is a hit even without a match) live in the
        // decision kernel as their own mode; this is NOT `classify(metas) > -1`
        static const std::vector<IBFMeta> none;
        BatchResult r = classify_batch(none, metas, config, {sequence}, RB_MODE_CLASSIFY_ANY);
        if (r.status[0] == RB_ERR_SHORT_READ) throw ShortReadException("Read " + id + " shorter than kmer size");
        throw_status(r.status[0], "classify");
        return r.decision[0] != 0;
    }

    // IBFClassify.cpp:239-297: index of the best matching filter or -1
    int classify(std::vector<IBFMeta>& filters, ClassifyConfig& config)
    {
        if (filters.empty()) throw NullFilterException("No IBF provided to classify the read!");
        static const std::vector<IBFMeta> none;
        BatchResult r = classify_batch(none, filters, config, {sequence}, RB_MODE_CLASSIFY_CHUNK);
        if (r.status[0] == RB_ERR_SHORT_READ) throw ShortReadException("Read " + id + " shorter than kmer size");
        throw_status(r.status[0], "classify");
        return r.best_target[0];
    }

    // IBFClassify.cpp:299-365: (max count over filt1, max count over filt2), thresholded
    std::pair<int, int> classify(std::vector<IBFMeta>& filt1, std::vector<IBFMeta>& filt2, ClassifyConfig& config)
    {
        if (filt1.empty() || filt2.empty()) throw NullFilterException("No IBF provided to classify the read!");
        BatchResult r = classify_batch(filt1, filt2, config, {sequence}, RB_MODE_CHECK_UNBLOCK);
        auto group_max = [&](const std::vector<IBFMeta>& fl, size_t base) {
            uint64_t best = 0;
            for (size_t i = 0; i < fl.size(); ++i) {
                if (sequence.size() < fl[i].filter.kmerSize) continue;  // :318,:340
                const uint16_t m = r.maxcount[base + i];
                const uint16_t t = rb_threshold(sequence.size(), fl[i].filter.kmerSize, config.error_rate, config.significance);
                const uint64_t c = m >= t ? m : 0;
                if (c > best) best = c;
            }
            return best;
        };
        return std::make_pair((int)group_max(filt1, 0), (int)group_max(filt2, filt1.size()));
    }
};

typedef std::vector<Read> TReads;

// calculateCI (src/IBF/IBF.hpp:320-338)
typedef std::pair<uint16_t, uint16_t> TInterval;
inline TInterval calculateCI(const double r, const uint8_t kmer_size, const uint32_t readlen, const double confidence)
{
    uint16_t lo = 0, hi = 0;
    if (rb_calculate_ci(r, kmer_size, readlen, confidence, &lo, &hi) != RB_OK)
        throw std::invalid_argument("Invalid input argument; must be larger than 0 but less than 1.");
    return TInterval{lo, hi};
}

}  // namespace interleave

// check_unblock (src/main/adaptive_sampling.hpp:35-113): 0 => do nothing; 1 => unblock; 2 => stop_further
inline uint8_t check_unblock(interleave::Read& read, interleave::ClassifyConfig& conf,
                             std::vector<interleave::IBFMeta>& DepletionFilters,
                             std::vector<interleave::IBFMeta>& TargetFilters)
{
    interleave::BatchResult r =
        interleave::classify_batch(DepletionFilters, TargetFilters, conf, {read.sequence}, RB_MODE_CHECK_UNBLOCK);
    if (r.status[0] == RB_ERR_SHORT_READ) throw interleave::ShortReadException("Read " + read.id + " shorter than kmer size");
    interleave::throw_status(r.status[0], "check_unblock");
    return r.decision[0];
}

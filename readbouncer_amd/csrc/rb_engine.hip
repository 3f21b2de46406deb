// rb_engine.hip -- device-resident filters (rb_dibf) and the per-GPU classification engine.
// Replaces, behind the C ABI: IBF::load_filter (src/IBF/IBFBuild.cpp:329-396), the three
// Read::classify overloads (src/IBF/IBF.hpp:211-213), check_unblock
// (src/main/adaptive_sampling.hpp:35-113) and one chunk of classify_reads
// (src/main/classify.hpp:262-299), all in batch form.  No CPU fallback: every entry point that
// computes fails with RB_ERR_NO_DEVICE / RB_ERR_HIP when there is no GPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <shared_mutex>

#include "rb_device.h"
#include "rb_io.h"
#include "rb_phase_plan.h"

using namespace rb;
using namespace rbplan;

#define RB_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            return rb::fail(e__ == hipErrorNoDevice ? RB_ERR_NO_DEVICE : RB_ERR_HIP,              \
                            std::string(#call) + ": " + hipGetErrorString(e__));                  \
        }                                                                                         \
    } while (0)

struct rb_dibf {
    int device = 0;
    rb_ibf_info geo{};
    uint64_t *d_words = nullptr;
    uint64_t stride = 0;  // words between consecutive blocks in HBM (>= geo.bin_width)
    IbfDev dev{};
    std::atomic<uint64_t> version{0};  // bumped by everything that changes the bits (insert, synthetic fill): engines that keep a
                                       // merged copy of several filters (MergedGroup) rebuild it when a member has moved on
    // placement by trial (dibf_alloc): allocations that were probed for this table, what the kept one and the worst one delivered
    uint32_t placement_tries = 0;
    double placement_gbps = 0.0, placement_worst_gbps = 0.0;
    // ... and what the trial cost (rb_dibf_placement_cost): seconds spent allocating and probing candidates, seconds waited afterwards
    // for the device to calm down, the most HBM the candidates held at once, and why a table of that size was NOT placed by trial
    double placement_trial_s = 0.0, placement_settle_s = 0.0;
    uint64_t placement_peak_bytes = 0;
    uint32_t placement_skipped = 0;  // 0: placed by trial (or too small / switched off); 1: an engine was live on the device; 2: not enough free HBM
    // ... and the wait that follows a trial (the driver clears the freed candidates in the background), when the caller of dibf_alloc has
    // work of its own to do first (streaming the file in): see dibf_settle
    bool settle_pending = false;
    uint32_t settle_row = 0;
    std::chrono::steady_clock::time_point settle_t0{};
};

// Tables of 1 GiB and more are PLACED BY TRIAL: the same table allocated at another moment of one process gathers 1.7-2.9 % slower or
// faster (config 3 at the reference's sizing, 4.7 GB: 628.5 against 618.0 ms per 2 M reads; GRCh38 at the default fragment size, 4.8 GB:
// 2 460 against 2 389 ms; the no-compute probe shows the same two levels, 6 750-6 835 and 6 900-6 936 GB/s, and K1 follows it --
// profiles/r05/placement_*.txt), whichever allocation comes first or last; a power-of-two table (8 GiB) always gets the fast kind.  What
// differs is where the driver finds the pages (contiguity, i.e. translation reach), which the library cannot ask for but can measure: up
// to `tries` allocations (default 5) are probed (random whole-block gathers, ~0.1 s each) while the earlier ones are still
// held; the trial ends early only when one is 3 % faster than the slowest seen; the best is kept, the others are freed.  Costs at most (tries - 1) x the table of HBM
// for a fraction of a second at load time; results never depend on it.  rb_set_placement_tries(1) switches it off.
static std::atomic<int> g_placement_tries{5};
static constexpr uint64_t kPlacementMinBytes = 1ull << 30;
// engines alive per device: a process that is already classifying on a device is not stalled for seconds (probe launches of 32 k waves,
// up to four more copies of the table held at once) because a second filter is loaded or cloned there -- such a table takes the first
// allocation.  The trial is for the start-up of a process: filters first, engines afterwards.
static constexpr int kPlacementDevices = 64;
static std::atomic<int> g_engines_on_device[kPlacementDevices];

// the wait after a trial: until the kept table probes like it did in the trial, at most 3 s after the candidates were freed
static double settle_table(void *table, uint64_t bytes, uint32_t row, double trial_gbps, std::chrono::steady_clock::time_point t0)
{
    const auto begin = std::chrono::steady_clock::now();
    while (std::chrono::steady_clock::now() - t0 < std::chrono::milliseconds(3000)) {
        double g = 0.0;
        if (rb::probe_read_peak_raw(table, bytes - 64, row, bytes > (512ull << 20), 24, 30.0, &g, nullptr) != RB_OK) break;
        if (g >= 0.993 * trial_gbps) break;
    }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - begin).count();
}

static hipError_t alloc_table_by_trial(uint64_t bytes, uint32_t block_bytes, uint64_t **out, uint32_t *tries_out, double *gbps_out, double *worst_out,
                                       rb_dibf *account, bool defer_settle)
{
    *tries_out = 0;
    *gbps_out = *worst_out = 0.0;
    int tries = g_placement_tries.load();
    size_t free_b = 0, total_b = 0;
    int dev = 0;
    if (bytes >= kPlacementMinBytes && tries > 1 && hipGetDevice(&dev) == hipSuccess && g_engines_on_device[(unsigned)dev % kPlacementDevices].load() > 0) {
        tries = 1;
        if (account) account->placement_skipped = 1;
    }
    if (bytes >= kPlacementMinBytes && tries > 1 && hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const int room = (int)std::min<uint64_t>((uint64_t)tries, (uint64_t)free_b / 2 / bytes);  // never more than half of what is free
        if (room <= 1 && account) account->placement_skipped = 2;
        tries = room;
    }
    if (bytes < kPlacementMinBytes || tries <= 1) {
        (void)hipGetLastError();
        return hipMalloc((void **)out, bytes);
    }
    const auto trial_t0 = std::chrono::steady_clock::now();
    const uint32_t row = block_bytes >= 3072 ? 4096u : block_bytes >= 1024 ? 1024u : 128u;
    std::vector<std::pair<double, void *>> cand;
    double worst = 0.0, best_g = 0.0;
    for (int t = 0; t < tries; ++t) {
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        double g = 0.0;
        // (the first probe of a trial may find the device in an idle clock state -- a filter is often created after seconds of host work --
        // and read 3-6 % low, which would make the SECOND candidate look like the fast kind: one discarded run of ~60 ms first)
        if (t == 0) (void)rb::probe_read_peak_raw(p, bytes - 64, row, bytes > (512ull << 20), 24, 60.0, &g, nullptr);
        if (rb::probe_read_peak_raw(p, bytes - 64, row, bytes > (512ull << 20), 24, 30.0, &g, nullptr) != RB_OK) g = 0.0;  // (a probe that fails only ends the trial)
        cand.emplace_back(g, p);
        worst = cand.size() == 1 ? g : std::min(worst, g);
        best_g = std::max(best_g, g);
        if (g <= 0.0) break;
        // the trial ends early only on clear evidence: a candidate 3 % above the slowest seen is of the fast kind (slow 6 680-6 750 GB/s,
        // fast 6 920-6 960; there is a middle kind at 6 820-6 900 that is NOT good enough to stop at, and candidates that are all alike
        // may all be slow -- the first versions stopped there and kept a slow table in one start of six)
        if (cand.size() >= 2 && best_g >= 1.03 * worst) break;
    }
    if (cand.empty()) return hipErrorOutOfMemory;
    size_t best = 0;
    for (size_t i = 1; i < cand.size(); ++i)
        if (cand[i].first > cand[best].first) best = i;
    for (size_t i = 0; i < cand.size(); ++i)
        if (i != best) (void)hipFree(cand[i].second);
    // Freeing the other candidates is not free: for about a second afterwards (34 GB of 8 GiB candidates) everything on the device gathers
    // ~2.4 % slower -- the driver clears released VRAM in the background (profiles/r05/placement/after_free_check.txt: 6 700 GB/s until 2 s
    // after the call, 6 870 from then on; no such stretch when nothing was freed).  A filter lives for hours, but a caller that measures or
    // serves at once should find a quiet device: wait, bounded, until the kept table probes like it did in the trial.
    // A caller that fills the table from a file first (rb_dibf_open: 0.3-0.8 s for 8 GiB) does its work in that second and waits for the rest
    // afterwards (dibf_settle).
    if (account) {
        account->placement_trial_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - trial_t0).count();
        account->placement_peak_bytes = (uint64_t)cand.size() * bytes;
    }
    if (cand.size() > 1) {
        const auto t0 = std::chrono::steady_clock::now();
        if (defer_settle && account) {
            account->settle_pending = true;
            account->settle_row = row;
            account->settle_t0 = t0;
        } else {
            const double waited = settle_table(cand[best].second, bytes, row, cand[best].first, t0);
            if (account) account->placement_settle_s = waited;
        }
    }
    *out = (uint64_t *)cand[best].second;
    *tries_out = (uint32_t)cand.size();
    *gbps_out = cand[best].first;
    *worst_out = worst;
    return hipSuccess;
}

// growable device buffer
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return RB_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = std::max(bytes, (size_t)256);
        want = want + want / 4;  // slack so that slowly growing batches do not reallocate every call
        RB_HIP(hipMalloc(&p, want));
        cap = want;
        return RB_OK;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// growable pinned host buffer (staging of micro-batches)
struct PinnedBuf {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes)
    {
        if (bytes <= cap) return RB_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        size_t want = std::max(bytes, (size_t)65536);
        want = want + want / 2;
        // (coherent, whatever HIP_HOST_COHERENT says: kernels read micro-batches from here and write results here)
        RB_HIP(hipHostMalloc(&p, want, hipHostMallocCoherent));
        cap = want;
        return RB_OK;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
    }
};

// Several narrow filters of ONE hash geometry merged into one table.  Every filter the reference builds with one
// fragment_size has noOfBits = BinSizeBits x 64 x binWidth (src/IBF/IBFBuild.cpp:404-413), i.e. noOfBlocks = BinSizeBits
// whatever its bin count: with equal k and h a k-mer hashes to the SAME block number in all of them.  The engine then keeps a
// copy in which block b holds the blocks b of all members side by side (deplete 2 words + three targets of 1 word = 5 words,
// padded to 8: one 64-byte gather), and ONE lookup per (k-mer, hash function) serves every member -- the path is bound by
// requests, so a read costs 1 428 requests instead of 5 712 on the reference's README shape.  The copy belongs to the engine
// (the filters are borrowed); it is rebuilt when a member's bits have changed since it was made.
//
// The copy itself (MergedTable) is shared by the engines of a process: engines that were given the same filters in the same order
// on one device -- the reference's N classification threads, one engine each (adaptive_sampling.hpp:745-751) -- gather from ONE
// merged table.  With a copy per engine, K engines on a GPU put K x 40 MB tables through the 4 MiB L2s at once and the
// clock-phased slices of one evicted those of the others (README shape, four engines: 0.78 x the rate of one).  A registry of
// weak references hands out the table; a per-device reader/writer lock keeps a call's "is the copy fresh? -> enqueue the kernel"
// (shared) apart from "make the copy again" (exclusive, after hipDeviceSynchronize: kernels of other engines may be reading it).
struct MergedTable {
    int device = 0;
    std::vector<const rb_dibf *> key_filters;  // the members, in the order their bins sit in a merged block
    std::vector<uint32_t> key_bit_begin;
    uint64_t *d_words = nullptr;
    uint64_t *d_inv = nullptr;  // two-word copies of fewer than 2^21 - 1 blocks: the COMPLEMENT of the copy, behind it in the same allocation (the
                                // multi-read build of the phased kernel ORs complemented words instead of masking and ANDing: rb_kernels.hip)
    uint64_t stride = 0, width = 0, n_blocks = 0;
    IbfDev dev{};
    std::vector<uint64_t> versions;  // rb_dibf::version of each member when the copy was made
    ~MergedTable()
    {
        if (!d_words) return;
        int cur = 0;
        const bool have = hipGetDevice(&cur) == hipSuccess;
        (void)hipSetDevice(device);
        (void)hipFree(d_words);
        if (have) (void)hipSetDevice(cur);
    }
};
constexpr int kMergedLockDevices = 64;
static std::shared_mutex &merged_rw(int device)
{
    static std::shared_mutex locks[kMergedLockDevices];
    return locks[(unsigned)device % kMergedLockDevices];
}
static std::shared_ptr<MergedTable> merged_table_for(int device, const std::vector<const rb_dibf *> &filters,
                                                     const std::vector<uint32_t> &bit_begin, uint64_t width)
{
    static std::mutex mu;
    static std::vector<std::weak_ptr<MergedTable>> known;
    std::lock_guard<std::mutex> lock(mu);
    std::shared_ptr<MergedTable> found;
    size_t keep = 0;
    for (size_t i = 0; i < known.size(); ++i) {
        std::shared_ptr<MergedTable> t = known[i].lock();
        if (!t) continue;  // its last engine is gone
        known[keep++] = known[i];
        if (!found && t->device == device && t->width == width && t->key_filters == filters && t->key_bit_begin == bit_begin) found = t;
    }
    known.resize(keep);
    if (found) return found;
    std::shared_ptr<MergedTable> t(new (std::nothrow) MergedTable());
    if (!t) return t;
    t->device = device;
    t->key_filters = filters;
    t->key_bit_begin = bit_begin;
    t->width = width;
    known.push_back(t);
    return t;
}

struct MergedGroup {
    std::vector<uint32_t> members;  // filter indices, engine order
    std::vector<uint32_t> bit_begin;  // per member: its first bin in a merged block (multiples of 64 unless `packed`)
    bool packed = false;            // members sit bit to bit: fewer word columns per block (README shape: 243 bins in four words, not five)
    std::shared_ptr<MergedTable> tab;  // the copy (shared with every engine that merges the same filters the same way)
    uint64_t stride = 0, width = 0, n_blocks = 0;  // width: word columns of a merged block as it is laid out (packed or not)
    IbfDev dev{};                   // = tab->dev once the copy exists
    MergeMap map{};                 // this engine's view: where each member's maximum goes
};

// What engines created from now on take for the reverse strand's image of N (rb_set_default_revcomp_of_n; ibf_spec.h for the rule)
static std::atomic<uint32_t> g_default_revcomp_of_n{rbspec::kRevCompOfN};

struct rb_engine {
    int device = 0;
    bool counted_on_device = false;  // g_engines_on_device (placement by trial is skipped on a device that is already classifying)
    std::vector<rb_dibf *> filters;  // deplete first, then target (borrowed)
    uint32_t nd = 0, nt = 0;
    hipStream_t stream = nullptr;
    // hipEvent pairs around the count kernels of each call, on the launch stream; resolved lazily
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_ring;
    size_t ev_used = 0;
    bool timing = false;
    std::mutex host_mu;
    int shard_rank = 0, shard_world = 1;
    // Filters of one hash geometry merged into one table (see MergedGroup below): 0 = never, 1 = when it pays (default), 2 = always
    int merge_mode = 1;
    std::vector<struct MergedGroup *> merged;
    std::vector<int> merged_of;  // per filter: index into `merged`, or -1
    bool merged_planned = false;
    uint64_t merge_max_bytes = 16ull << 30;  // a merged copy costs HBM beside its members: larger groups stay apart (RB_MERGE_MAX_BYTES)
    uint32_t revcomp_of_n = rbspec::kRevCompOfN;  // see ibf_spec.h: what the reverse strand holds for an N of the read
    uint64_t nt_threshold_bytes = 512ull << 20;  // 2x the 256 MiB Infinity Cache: beyond it caching cannot help
    uint64_t serial_table_bytes = 128ull << 20;  // filters up to this size never run beside another filter (L2 share)
    // clock-phased gathers (rb_kernels.hip): tables between these sizes, batches of at least phase_min_reads reads
    uint64_t phase_min_bytes = 5ull << 18, phase_max_bytes = 128ull << 20;
    // window length in 10 ns ticks.  phase_explicit: base + per MiB of table, as given to rb_engine_set_phased; otherwise the
    // built-in rule of phase_window_ticks() below (measured per kernel shape, profiles/r03/window_sweep.txt).
    uint32_t phase_base_ticks = 450, phase_ticks_per_mib = 0;
    bool phase_explicit = false;
    uint32_t wall_clock_khz = 100000;  // rate of the device's wall clock (s_memrealtime): windows are given in 10 ns ticks
    // rb_engine_set_phase_slices (RB_PHASE_MAX_SLICES, RB_PHASE_SLICE_LOG2 for whole processes): tests and experiments
    uint32_t phase_max_slices = 32;   // slices a table is cut into (<= 32: a wave keeps a bit per slice)
    uint32_t phase_slice_log2 = 0;    // slices of 2^n bytes instead of the rule of phase_slice_log2(); 1-5: as small as phase_max_slices allows
    uint32_t phase_n_slices = 0;      // RB_PHASE_N_SLICES: that many equal-length slices for the four-word one-lane builds (0: phase_equal_slices())
    uint32_t phase_xcd_skew = 2;      // rb_engine_set_phase_xcd_skew (RB_PHASE_XCD_SKEW): bit 0: slice = (window + XCD number) mod n_slices; bit 1: the
                                      // XCDs' windows start an eighth of a window apart (they refill their L2s one after the other).  Bit 1 is the
                                      // default since round 6: 1.2-4.4 % on every phased shape, never a loss (profiles/r06/multi/s13_skew_all_shapes.txt)
    // rb_engine_set_reads_per_wave: two-word tables of up to 2^21 - 1 blocks, reads of up to 256 k-mers, phased: the build that carries
    // that many reads per wave through a pass of the windows, offsets in LDS (rb_kernels.hip, ibf_count_max_phased_multi_kernel); 0: the
    // one-read build
    uint32_t multi_reads = 1;   // (default since round 6: 250 bp -10 ... -11 %, 360 bp -16 ... -19 % on two-word tables, profiles/r06/multi/)
    uint32_t phase_tskew_div = 8;  // RB_PHASE_TSKEW_DIV: the XCDs' windows start 1 / this of a window apart (time skew)
    // rb_engine_set_early_decision (opt-in, off by default): RB_MODE_CHECK_UNBLOCK calls of the throughput form that do not ask for the raw maxima
    // let a wave of the plain count kernel stop once a bin has reached the larger of the read's two thresholds (rb_kernels.hip, EarlyCfg)
    bool early_decision = false;
    bool multi_one_word = true;  // ... for one-word blocks of up to 2^22 - 2 of them (32 MiB) (RB_MULTI_ONE_WORD=0: the register builds)
    bool multi_wide = true;      // ... and for blocks of three and four words (RB_MULTI_WIDE=0: those keep the register builds)
    bool multi_wide_six = true;  // ... six tiles in one round for their reads of 257-384 k-mers (RB_MULTI_WIDE_SIX=0: rounds of three tiles)
    bool multi_no_inv = false;  // (bit 4 of rb_engine_set_reads_per_wave's argument: the AND form on merged copies too; measurements)
    // rb_engine_calibrate: window lengths measured on this device that replace the planner's for a (table, kernel shape, slice size)
    struct PhaseOverride {
        uint64_t table_bytes;
        uint32_t stride, slice_log2, ticks;
        int shape, lg;
    };
    std::vector<PhaseOverride> phase_overrides;
    uint32_t phase_min_reads = 2049;  // everything above the latency kernel's batches: README shape at 2 049 reads per call 8.7 -> 9.2 M reads/s, 4 096: 11.6 -> 16.5 M, 65 536: 16.0 -> 28.3 M (profiles/r03/phased_batch_size.txt)
    bool short_read_kernel = true;
    int six_tile_kernel = 1;  // reads of 257-384 k-mers (360 bp): one round of six tiles per strand (RB_SIX_TILES=0: two rounds of four)
    uint32_t split_threshold = 2048;  // batches up to this many (read, slice) items use the latency kernel
    uint32_t split_max_parts = 8, split_max_sub = 4;  // latency kernel on wide filters: workgroups per read, shares per tile
    DevBuf d_split_ws, d_split_tickets;
    bool tickets_dirty = false;  // a call that launched the multi-workgroup latency kernel did not come back clean: the arrival
                                 // counters may be non-zero (the kernel zeroes them itself only when it runs to its end)
    // one-filter engines, micro-batches up to fold_max_reads in the latency form: the count kernel makes the decisions too (FoldJob,
    // rb_device.h) -- one dependent launch less per call (deplete only: 1 read 41.5 -> 40.3 us, 64: 64.1 -> 62.8, 256: 119.2 -> 117.4;
    // nothing from 1 024 reads on: profiles/r05/fold_decide_ab.txt)
    bool fold_decide = true;
    uint32_t fold_max_reads = 512;
    // Opt-in (rb_engine_set_completion_word): host micro-batches of up to completion_max_reads reads learn of their results from a word of
    // page-locked memory that the call's last kernel stores, and the calling thread spins on that word instead of waiting for the stream --
    // hipStreamSynchronize returns about 4 us after the word is there (one read 40.6 -> 36.1 us, 64 reads 64.5 -> 60.0, config 5's service time
    // 37.9 -> 33.0 us: profiles/r05/completion_word_ab.txt).  Not the default because of the tail: the runtime retires its commands in
    // hipStreamSynchronize, a stream that is not waited for retires them in bulk -- bursts of 5-8 calls of + 13 us every ~330 calls -- and
    // p99 of the replay rises by 0-20 us from box to box (completion_word_tail.txt).  A word that does not arrive within completion_spin_us
    // falls back to the stream (which also reports a failed kernel); every completion_sync_every-th call waits for the stream as well.
    bool completion_word = false;
    uint32_t completion_max_reads = 2048;
    uint32_t completion_spin_us = 20000;
    PinnedBuf h_done;         // the word
    uint32_t done_seq = 0;    // last sequence number handed out (never 0)
    DevBuf d_done_count;      // arrival counter of decision kernels with more than one workgroup; zero between calls
    bool done_dirty = true;   // ... unless a call did not come back clean (or the counter is new)
    uint32_t completion_sync_every = 256;
    uint32_t word_calls = 0;  // calls since the stream was last waited for
    // threshold tables u16[len][filter][{r, r-0.02}], one per (error rate, significance) pair; the two most recently used
    // pairs stay resident so that a caller alternating two error rates never rebuilds (or waits for) a table.  A table
    // that is replaced or outgrown may still be read by queued kernels: its device block is parked in thr_retired.
    struct ThrTable {
        void *d = nullptr;       // device copy
        PinnedBuf host;          // page-locked host copy (source of the asynchronous upload; rows are appended on growth)
        uint32_t len = 0;        // rows
        size_t bytes = 0;        // size of each copy
        double r = -1.0, conf = -1.0;
        uint64_t last_use = 0;
        hipStream_t up_stream = nullptr;  // stream the last upload was queued on
        hipEvent_t ready = nullptr;       // recorded behind it: a call on ANOTHER stream waits for this before it reads the table
    };
    ThrTable thr[2];
    uint64_t thr_clock = 0;
    // replaced copies, device and host: a queued kernel may still read the device block, a queued upload the host block
    std::vector<void *> thr_retired_dev;
    std::vector<PinnedBuf> thr_retired_host;
    size_t thr_retired_bytes = 0;
    // workspaces
    DevBuf d_maxcount;
    std::vector<DevBuf> d_parts;  // per filter: partial maxima of the column slices
    // filters run concurrently: filter 0 on the caller's stream, the others on auxiliary streams (fork/join by events)
    std::vector<hipStream_t> aux;
    hipEvent_t fork_ev = nullptr;
    std::vector<hipEvent_t> join_ev;
    bool overlap = true;
    // staging for the host-pointer API
    DevBuf d_seqs, d_offsets, d_lens, d_best, d_decision, d_status;
    DevBuf d_efflens, d_prestatus;  // on-GPU chunking: effective lengths and the bad-chunk status per item
    PinnedBuf h_in, h_out;
    // large host batches: slice i+1 is copied on this stream while slice i is counted on `stream`
    hipStream_t copy_stream = nullptr;
    std::vector<hipEvent_t> copy_ev;
    uint64_t host_slice_bytes = (uint64_t)32 << 20;
    uint64_t micro_copy_kernel_bytes = 1 << 20;  // micro-batches up to this size enter HBM by launch_copy_from_host (0: always the runtime's copy)
    std::mutex mu;
};

static int check_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return rb::fail(RB_ERR_NO_DEVICE, std::string("no HIP device: ") + (e == hipSuccess ? "count is 0" : hipGetErrorString(e)));
    if (device < 0 || device >= n) return rb::fail(RB_ERR_INVALID_ARG, "device index out of range");
    RB_HIP(hipSetDevice(device));
    return RB_OK;
}

// ---- HBM layout -------------------------------------------------------------------------------------------
// The .ibf file stores the blocks back to back (W = bin_width words each).  In HBM a block starts every `stride`
// words, stride >= W chosen so that a block never straddles more 128-byte lines than it has to:
//   W a multiple of 16 (128 B): stride = W (file layout, verbatim);  W < 16: next power of two;  else: next multiple of 16.
// Measured on a 600-bin filter (W = 10, 80-byte blocks): the same gathers from 128-byte aligned blocks run 1.37x faster
// although 1.6x more bytes are fetched; gfx950 also needs 16-byte alignment for the dwordx4 lanes of odd widths only
// to be fast, not to be correct.  Upload widens, download narrows; the file format is untouched.
static uint64_t hbm_stride(uint64_t W)
{
    if (W % 16 == 0) return W;
    if (W < 16) {
        uint64_t s = 1;
        while (s < W) s <<= 1;
        return s;
    }
    return (W + 15) / 16 * 16;
}

static int make_dev_desc(const rb_ibf_info &g, const uint64_t *d_words, uint64_t stride, IbfDev *d)
{
    if (g.n_hash > rbspec::kMaxHash) return rb::fail(RB_ERR_UNSUPPORTED, "more than 8 hash functions");
    if (g.kmer_size > rbspec::kMaxKmer) return rb::fail(RB_ERR_UNSUPPORTED, "k-mer size above 32");
    if (g.n_blocks == 0) return rb::fail(RB_ERR_INVALID_ARG, "filter has no blocks (n_bits smaller than one block)");
    if (g.n_blocks >= (1ULL << 32)) return rb::fail(RB_ERR_UNSUPPORTED, "more than 2^32-1 blocks");
    if (g.n_bins >= (1ULL << 31)) return rb::fail(RB_ERR_UNSUPPORTED, "too many bins");
    d->words = d_words;
    d->n_blocks = (uint32_t)g.n_blocks;
    const bool pow2 = (g.n_blocks & (g.n_blocks - 1)) == 0;
    d->pow2_mask = pow2 ? (uint32_t)(g.n_blocks - 1) : 0xFFFFFFFFu;
    d->magic = g.n_blocks >= 2 ? rbspec::fastmod_magic(g.n_blocks) : 0;
    for (unsigned i = 0; i < rbspec::kMaxHash; ++i) d->precalc[i] = rbspec::precalc(g.kmer_size, i);
    d->n_bins = (uint32_t)g.n_bins;
    d->bin_width = (uint32_t)g.bin_width;
    d->stride = (uint32_t)stride;
    d->k = (uint32_t)g.kmer_size;
    d->n_hash = (uint32_t)g.n_hash;
    d->comp_n = rbspec::kRevCompOfN;
    return RB_OK;
}

// device words of a filter: n_blocks * stride, plus a small zero tail (a 16-byte lane may read one word past a block)
static uint64_t dibf_device_words(const rb_dibf *f) { return f->geo.n_blocks * f->stride + 8; }

static int dibf_alloc(int device, const rb_ibf_info &g, bool zero, rb_dibf **out, bool defer_settle = false)
{
    int st = check_device(device);
    if (st != RB_OK) return st;
    rb_dibf *f = new (std::nothrow) rb_dibf();
    if (!f) return rb::fail(RB_ERR_NOMEM, "alloc");
    f->device = device;
    f->geo = g;
    f->stride = hbm_stride(g.bin_width);
    hipError_t e = alloc_table_by_trial(dibf_device_words(f) * 8, (uint32_t)(f->stride * 8), &f->d_words, &f->placement_tries, &f->placement_gbps,
                                        &f->placement_worst_gbps, f, defer_settle);
    if (e != hipSuccess) {
        delete f;
        return rb::fail(RB_ERR_HIP, std::string("hipMalloc of the IBF failed: ") + hipGetErrorString(e));
    }
    if (zero) {
        e = hipMemset(f->d_words, 0, dibf_device_words(f) * 8);
        if (e != hipSuccess) { rb_dibf_free(f); return rb::fail(RB_ERR_HIP, hipGetErrorString(e)); }
    } else {
        e = hipMemset(f->d_words + f->geo.n_blocks * f->stride, 0, 8 * 8);
        if (e != hipSuccess) { rb_dibf_free(f); return rb::fail(RB_ERR_HIP, hipGetErrorString(e)); }
    }
    st = make_dev_desc(g, f->d_words, f->stride, &f->dev);
    if (st != RB_OK) { rb_dibf_free(f); return st; }
    *out = f;
    return RB_OK;
}

// the second half of a placement trial whose wait was deferred (dibf_alloc(..., defer_settle = true)): call it when the table is filled
static void dibf_settle(rb_dibf *f)
{
    if (!f || !f->settle_pending) return;
    f->settle_pending = false;
    f->placement_settle_s = settle_table(f->d_words, dibf_device_words(f) * 8, f->settle_row, f->placement_gbps, f->settle_t0);
}

// `words` 64-bit words into d_dst through two page-locked 64 MiB staging buffers: `fill(dst, first_word, n_words)` produces chunk i + 1
// (several threads: rb_io.h) while chunk i crosses PCIe.  Returns RB_OK, or the failure of fill (its own code) / of the runtime.
template <typename Fill>
static int stream_words_to_device(uint64_t *d_dst, uint64_t words, Fill fill)
{
    const size_t chunk_words = (size_t)8 << 20;  // 64 MiB
    uint64_t *stage[2] = {nullptr, nullptr};
    hipStream_t s = nullptr;
    hipError_t e = hipStreamCreate(&s);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipHostMalloc((void **)&stage[i], std::min<uint64_t>(chunk_words, std::max<uint64_t>(words, 1)) * 8, hipHostMallocDefault);
    hipEvent_t done[2] = {nullptr, nullptr};
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
    int rc = RB_OK;
    if (e != hipSuccess) rc = rb::fail(RB_ERR_HIP, std::string("staging setup: ") + hipGetErrorString(e));
    uint64_t pos = 0;
    int slot = 0;
    bool used_slot[2] = {false, false};
    while (rc == RB_OK && pos < words) {
        const size_t nw = (size_t)std::min<uint64_t>(chunk_words, words - pos);
        if (used_slot[slot]) (void)hipEventSynchronize(done[slot]);
        if ((rc = fill(stage[slot], pos, nw)) != RB_OK) break;
        e = hipMemcpyAsync(d_dst + pos, stage[slot], nw * 8, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipEventRecord(done[slot], s);
        if (e != hipSuccess) { rc = rb::fail(RB_ERR_HIP, hipGetErrorString(e)); break; }
        used_slot[slot] = true;
        pos += nw;
        slot ^= 1;
    }
    if (s) (void)hipStreamSynchronize(s);
    for (int i = 0; i < 2; ++i) {
        if (done[i]) (void)hipEventDestroy(done[i]);
        if (stage[i]) (void)hipHostFree(stage[i]);
    }
    if (s) (void)hipStreamDestroy(s);
    return rc;
}

// file-layout words already on the device (d_compact: n_blocks * W words) -> this filter's padded layout
static int dibf_from_compact(rb_dibf *f, const uint64_t *d_compact)
{
    hipError_t e = launch_restride_blocks(d_compact, (uint32_t)f->geo.bin_width, f->d_words, (uint32_t)f->stride,
                                          (uint32_t)f->geo.bin_width, f->geo.n_blocks, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return rb::fail(RB_ERR_HIP, std::string("layout conversion: ") + hipGetErrorString(e));
    return RB_OK;
}

extern "C" {

int rb_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return -(int)RB_ERR_NO_DEVICE;
    return n;
}

int rb_dibf_create(int device, uint64_t n_bins, uint64_t n_hash, uint64_t kmer_size, uint64_t n_bits, rb_dibf **out)
{
    if (!out) return rb::fail(RB_ERR_INVALID_ARG, "null out");
    rb_ibf_info g;
    if (!geometry_from(n_bins, n_hash, kmer_size, n_bits, &g)) return rb::fail(RB_ERR_INVALID_ARG, "bad IBF geometry");
    return dibf_alloc(device, g, true, out);
}

int rb_dibf_upload(int device, const rb_ibf *host, rb_dibf **out)
{
    if (!host || !out) return rb::fail(RB_ERR_INVALID_ARG, "null argument");
    rb_dibf *f = nullptr;
    int st = dibf_alloc(device, host->geo, false, &f, true);
    if (st != RB_OK) return st;
    const uint64_t used = host->geo.n_blocks * host->geo.bin_width;  // block payload; tail bits and metadata stay on the host
    // the image is pageable memory: copied by the runtime it crosses PCIe at 3.9 GB/s (8 GiB: 2.2 s); staged through page-locked buffers by
    // a few threads it does not wait for the host (profiles/r05/load_throughput.txt)
    const bool padded = f->stride != host->geo.bin_width;
    uint64_t *d_dst = f->d_words;
    if (padded && hipMalloc((void **)&d_dst, std::max<uint64_t>(used, 1) * 8) != hipSuccess) {
        (void)hipGetLastError();
        rb_dibf_free(f);
        return rb::fail(RB_ERR_HIP, "hipMalloc of the file-layout buffer failed");
    }
    const uint64_t *src = host->words;
    rb::IoGang gang(rb::io_threads((size_t)std::min<uint64_t>(used, (uint64_t)8 << 20) * 8));  // (the threads of this load: one chunk is 64 MiB)
    st = stream_words_to_device(d_dst, used, [src, &gang](uint64_t *dst, uint64_t first, size_t n) {
        gang.memcpy(dst, src + first, n * 8);
        return (int)RB_OK;
    });
    if (st == RB_OK && padded) st = dibf_from_compact(f, d_dst);
    if (padded && d_dst) (void)hipFree(d_dst);
    if (st != RB_OK) { rb_dibf_free(f); return st; }
    dibf_settle(f);
    *out = f;
    return RB_OK;
}

int rb_dibf_open(int device, const char *path, rb_dibf **out)
{
    if (!out) return rb::fail(RB_ERR_INVALID_ARG, "null out");
    FILE *fp = nullptr;
    rb_ibf_info g;
    int st = open_ibf_stream(path, &fp, &g);
    if (st != RB_OK) return st;
    rb_dibf *f = nullptr;
    st = dibf_alloc(device, g, false, &f, true);
    if (st != RB_OK) { std::fclose(fp); return st; }
    // stream the block payload through two pinned staging buffers: the file read of chunk i+1 (several threads, pread) overlaps the H2D
    // copy of chunk i; the 8 GB GRCh38 filter never needs a host-side image.  A padded layout lands in a temporary file-layout buffer
    // on the device first and is widened there.
    const uint64_t used = g.n_blocks * g.bin_width;
    const bool padded = f->stride != g.bin_width;
    uint64_t *d_dst = f->d_words;
    int rc = RB_OK;
    if (padded && hipMalloc((void **)&d_dst, std::max<uint64_t>(used, 1) * 8) != hipSuccess) {
        (void)hipGetLastError();
        d_dst = nullptr;
        rc = rb::fail(RB_ERR_HIP, "hipMalloc of the file-layout buffer failed");
    }
    const int fd = fileno(fp);
    const std::string name = path;
    rb::IoGang gang(rb::io_threads((size_t)std::min<uint64_t>(used, (uint64_t)8 << 20) * 8));  // (the threads of this load: one chunk is 64 MiB)
    if (rc == RB_OK)
        rc = stream_words_to_device(d_dst, used, [fd, &name, &gang](uint64_t *dst, uint64_t first, size_t n) {
            if (!gang.pread(fd, (off_t)(8 + first * 8), dst, n * 8)) return rb::fail(RB_ERR_PARSE_IBF, name + ": short read");
            return (int)RB_OK;
        });
    std::fclose(fp);
    if (rc == RB_OK && padded) rc = dibf_from_compact(f, d_dst);
    if (padded && d_dst) (void)hipFree(d_dst);
    if (rc != RB_OK) { rb_dibf_free(f); return rc; }
    dibf_settle(f);
    *out = f;
    return RB_OK;
}

// Replica of a device-resident filter on another GPU (or a second copy on the same one), device to device: over xGMI
// when the two devices can reach each other (peer access is switched on for the pair), else staged by the runtime.  The
// padded HBM image travels as it lies -- no host image, no layout conversion.  `stream_out`: NULL = wait for the copy;
// otherwise the copy is left running on a new stream returned there (the caller synchronises and destroys it), so that
// one source can feed several destinations at once, one xGMI link each.
int rb_dibf_clone_to_impl(const rb_dibf *src, int device, rb_dibf **out, hipStream_t *stream_out, int *used_peer)
{
    if (!src || !out) return rb::fail(RB_ERR_INVALID_ARG, "null argument");
    rb_dibf *f = nullptr;
    int st = dibf_alloc(device, src->geo, false, &f);
    if (st != RB_OK) return st;
    const size_t bytes = dibf_device_words(src) * 8;
    hipError_t e = hipSuccess;
    int peer = 0;
    if (device != src->device) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, device, src->device) == hipSuccess && can) {
            e = hipDeviceEnablePeerAccess(src->device, 0);  // current device (= destination, set by dibf_alloc) maps the source
            if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
            peer = e == hipSuccess;
            e = hipSuccess;  // without the mapping hipMemcpyPeer stages through the host: slower, still correct
        }
    }
    hipStream_t s = nullptr;
    e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e == hipSuccess) {
        e = device == src->device ? hipMemcpyAsync(f->d_words, src->d_words, bytes, hipMemcpyDeviceToDevice, s)
                                  : hipMemcpyPeerAsync(f->d_words, device, src->d_words, src->device, bytes, s);
    }
    if (e == hipSuccess && !stream_out) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        if (s) (void)hipStreamDestroy(s);
        rb_dibf_free(f);
        return rb::fail(RB_ERR_HIP, std::string("device-to-device replication: ") + hipGetErrorString(e));
    }
    if (stream_out) *stream_out = s;
    else (void)hipStreamDestroy(s);
    if (used_peer) *used_peer = peer;
    *out = f;
    return RB_OK;
}

int rb_dibf_clone_to(const rb_dibf *src, int device, rb_dibf **out)
{
    return rb_dibf_clone_to_impl(src, device, out, nullptr, nullptr);
}

int rb_dibf_clone_to_ex(const rb_dibf *src, int device, rb_dibf **out, int *used_peer, double *seconds)
{
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = rb_dibf_clone_to_impl(src, device, out, nullptr, used_peer);
    if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

// wait for a clone started with a stream_out (rb_pool.cpp is a host translation unit: it does not see HIP types)
int rb_dibf_clone_finish(void *stream)
{
    if (!stream) return RB_OK;
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) return rb::fail(RB_ERR_HIP, std::string("device-to-device replication: ") + hipGetErrorString(e));
    return RB_OK;
}

int rb_dibf_clone_start(const rb_dibf *src, int device, rb_dibf **out, void **stream, int *used_peer)
{
    hipStream_t s = nullptr;
    const int rc = rb_dibf_clone_to_impl(src, device, out, &s, used_peer);
    if (stream) *stream = (void *)s;
    return rc;
}

int rb_dibf_download(const rb_dibf *f, rb_ibf **out)
{
    if (!f || !out) return rb::fail(RB_ERR_INVALID_ARG, "null argument");
    rb_ibf *h = nullptr;
    int st = rb_ibf_create(f->geo.n_bins, f->geo.n_hash, f->geo.kmer_size, f->geo.n_bits, &h);  // zeroed: tail + metadata
    if (st != RB_OK) return st;
    st = check_device(f->device);
    if (st != RB_OK) { rb_ibf_close(h); return st; }
    const uint64_t used = f->geo.n_blocks * f->geo.bin_width;
    hipError_t e = hipSuccess;
    if (f->stride == f->geo.bin_width) {
        e = hipMemcpy(h->words, f->d_words, used * 8, hipMemcpyDeviceToHost);
    } else {
        uint64_t *tmp = nullptr;
        e = hipMalloc((void **)&tmp, std::max<uint64_t>(used, 1) * 8);
        if (e == hipSuccess)
            e = launch_restride_blocks(f->d_words, (uint32_t)f->stride, tmp, (uint32_t)f->geo.bin_width,
                                       (uint32_t)f->geo.bin_width, f->geo.n_blocks, nullptr);
        if (e == hipSuccess) e = hipMemcpy(h->words, tmp, used * 8, hipMemcpyDeviceToHost);
        if (tmp) (void)hipFree(tmp);
    }
    if (e != hipSuccess) { rb_ibf_close(h); return rb::fail(RB_ERR_HIP, hipGetErrorString(e)); }
    *out = h;
    return RB_OK;
}

int rb_dibf_get_info(const rb_dibf *f, rb_ibf_info *info)
{
    if (!f || !info) return rb::fail(RB_ERR_INVALID_ARG, "null argument");
    *info = f->geo;
    return RB_OK;
}

void *rb_dibf_device_words(rb_dibf *f) { return f ? f->d_words : nullptr; }
int rb_dibf_touch(rb_dibf *f)
{
    if (!f) return rb::fail(RB_ERR_INVALID_ARG, "null filter");
    f->version.fetch_add(1);
    return RB_OK;
}
uint64_t rb_dibf_device_stride(const rb_dibf *f) { return f ? f->stride : 0; }
int rb_dibf_device(const rb_dibf *f) { return f ? f->device : -1; }

void rb_dibf_free(rb_dibf *f)
{
    if (!f) return;
    if (f->d_words) {
        (void)hipSetDevice(f->device);
        (void)hipFree(f->d_words);
    }
    delete f;
}

int rb_dibf_resize_bins(const rb_dibf *f, uint64_t new_bins, rb_dibf **out)
{
    if (!f || !out) return rb::fail(RB_ERR_INVALID_ARG, "null argument");
    if (new_bins < f->geo.n_bins) return rb::fail(RB_ERR_INVALID_ARG, "resizeBins cannot shrink a filter");
    const uint64_t new_width = (new_bins + rbspec::kIntSize - 1) / rbspec::kIntSize;
    rb_ibf_info g;
    if (!geometry_from(new_bins, f->geo.n_hash, f->geo.kmer_size, f->geo.n_blocks * new_width * 64, &g))
        return rb::fail(RB_ERR_INVALID_ARG, "bad IBF geometry");
    rb_dibf *n = nullptr;
    int st = dibf_alloc(f->device, g, true, &n);
    if (st != RB_OK) return st;
    hipError_t e = launch_restride_blocks(f->d_words, (uint32_t)f->stride, n->d_words, (uint32_t)n->stride,
                                          (uint32_t)f->geo.bin_width, f->geo.n_blocks, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { rb_dibf_free(n); return rb::fail(RB_ERR_HIP, std::string("resize: ") + hipGetErrorString(e)); }
    *out = n;
    return RB_OK;
}

int rb_dibf_fill_synth(rb_dibf *f, uint64_t seed)
{
    if (!f) return rb::fail(RB_ERR_INVALID_ARG, "null filter");
    int st = check_device(f->device);
    if (st != RB_OK) return st;
    const uint64_t used = f->geo.n_blocks * f->geo.bin_width;
    const uint64_t rem = f->geo.n_bins & 63;
    const uint64_t last_mask = rem ? ((1ULL << rem) - 1) : ~0ULL;
    // the version moves before the first write AND after the last one has landed: an engine that copies the table into a merged
    // group in between records the intermediate number and finds its copy stale at its next call
    f->version.fetch_add(1);
    RB_HIP(hipMemset(f->d_words, 0, dibf_device_words(f) * 8));
    RB_HIP(launch_fill_synth(f->d_words, used, (uint32_t)f->geo.bin_width, (uint32_t)f->stride, last_mask, seed, nullptr));
    RB_HIP(hipDeviceSynchronize());
    f->version.fetch_add(1);
    return RB_OK;
}

int rb_dibf_compare(const rb_dibf *file_filter, const rb_dibf *rebuilt, rb_ibf_compare *out)
{
    if (!file_filter || !rebuilt || !out) return rb::fail(RB_ERR_INVALID_ARG, "null argument");
    const rb_ibf_info &a = file_filter->geo, &b = rebuilt->geo;
    if (a.n_bins != b.n_bins || a.n_hash != b.n_hash || a.kmer_size != b.kmer_size || a.n_bits != b.n_bits ||
        file_filter->device != rebuilt->device)
        return rb::fail(RB_ERR_INVALID_ARG, "filters of different geometry (or on different devices) cannot be compared");
    int st = check_device(file_filter->device);
    if (st != RB_OK) return st;
    uint64_t *d_out = nullptr;
    RB_HIP(hipMalloc((void **)&d_out, 24));
    hipError_t e = hipMemset(d_out, 0, 24);
    // padding words and padding bits are zero in both images, so the padded HBM form can be compared as it lies
    if (e == hipSuccess) e = launch_compare_bits(file_filter->d_words, rebuilt->d_words, a.n_blocks * file_filter->stride, d_out, nullptr);
    uint64_t h[3] = {0, 0, 0};
    if (e == hipSuccess) e = hipMemcpy(h, d_out, 24, hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    if (e != hipSuccess) return rb::fail(RB_ERR_HIP, std::string("compare: ") + hipGetErrorString(e));
    out->file_bits = h[0];
    out->rebuilt_bits = h[1];
    out->new_bits = h[2];
    out->payload_bits = a.n_blocks * a.n_bins;
    return RB_OK;
}

int rb_dibf_insert(rb_dibf *f, const char *seq, size_t len, const uint64_t *starts, const uint64_t *ends,
                   const uint64_t *bins, size_t n_fragments)
{
    if (!f || (!seq && len) || (n_fragments && (!starts || !ends || !bins))) return rb::fail(RB_ERR_INVALID_ARG, "null argument");
    if (n_fragments == 0) return RB_OK;
    if (n_fragments >= (1ULL << 31)) return rb::fail(RB_ERR_INVALID_ARG, "too many fragments in one call");
    int st = check_device(f->device);
    if (st != RB_OK) return st;
    const uint64_t k = f->geo.kmer_size;
    std::vector<uint64_t> prefix(n_fragments + 1, 0);
    for (size_t i = 0; i < n_fragments; ++i) {
        if (ends[i] > len || starts[i] > ends[i]) return rb::fail(RB_ERR_INVALID_ARG, "fragment outside the sequence");
        const uint64_t flen = ends[i] - starts[i];
        const uint64_t nk = flen >= k ? flen - k + 1 : 0;
        // a tail fragment shorter than k carries no k-mer; the reference hands it a bin id past the
        // predicted bin count (IBFBuild.cpp:171-190 vs :90) and insertKmer then touches nothing
        if (nk && bins[i] >= f->geo.n_bins) return rb::fail(RB_ERR_INVALID_ARG, "bin id outside the filter");
        prefix[i + 1] = prefix[i] + nk;
    }
    const uint64_t total = prefix[n_fragments];
    if (total == 0) return RB_OK;
    if ((total + 255) / 256 >= (1ULL << 31)) return rb::fail(RB_ERR_INVALID_ARG, "too many k-mers in one call");
    f->version.fetch_add(1);
    DevBuf d_seq, d_tab;
    st = d_seq.ensure(len ? len : 1);
    if (st == RB_OK) st = d_tab.ensure((4 * n_fragments + 1) * 8);
    if (st != RB_OK) { d_seq.release(); d_tab.release(); return st; }
    uint64_t *t = (uint64_t *)d_tab.p;
    hipError_t e = hipMemcpy(d_seq.p, seq, len, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t, starts, n_fragments * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t + n_fragments, ends, n_fragments * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t + 2 * n_fragments, bins, n_fragments * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(t + 3 * n_fragments, prefix.data(), (n_fragments + 1) * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = launch_insert(f->dev, f->d_words, (const uint8_t *)d_seq.p, t, t + n_fragments, t + 2 * n_fragments,
                          t + 3 * n_fragments, (uint32_t)n_fragments, total, nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    d_seq.release();
    d_tab.release();
    f->version.fetch_add(1);  // (before and after the writes, see rb_dibf_fill_synth)
    if (e != hipSuccess) return rb::fail(RB_ERR_HIP, std::string("insert: ") + hipGetErrorString(e));
    return RB_OK;
}

int rb_dibf_add_sequence(rb_dibf *f, const char *seq, size_t len, uint64_t fragment_length, uint64_t overlap_length,
                         uint64_t first_bin, uint64_t *next_bin)
{
    if (!f) return rb::fail(RB_ERR_INVALID_ARG, "null filter");
    const size_t n = rb_fragment_bounds(len, fragment_length, f->geo.kmer_size, overlap_length, nullptr, nullptr, 0);
    std::vector<uint64_t> starts(n), ends(n), bins(n);
    rb_fragment_bounds(len, fragment_length, f->geo.kmer_size, overlap_length, starts.data(), ends.data(), n);
    for (size_t i = 0; i < n; ++i) bins[i] = first_bin + i;
    if (next_bin) *next_bin = first_bin + n;
    return rb_dibf_insert(f, seq, len, starts.data(), ends.data(), bins.data(), n);
}

// ------------------------------------------------------------------------------------------------
int rb_engine_create(int device, rb_dibf *const *deplete, size_t n_deplete, rb_dibf *const *target, size_t n_target,
                     rb_engine **out)
{
    if (!out) return rb::fail(RB_ERR_INVALID_ARG, "null out");
    if (n_deplete + n_target == 0) return rb::fail(RB_ERR_NULL_FILTER, "No IBF provided to classify the read!");
    if (n_deplete + n_target > kMaxFilters) return rb::fail(RB_ERR_UNSUPPORTED, "more than 16 filters");
    int st = check_device(device);
    if (st != RB_OK) return st;
    rb_engine *e = new (std::nothrow) rb_engine();
    if (!e) return rb::fail(RB_ERR_NOMEM, "alloc");
    e->device = device;
    for (size_t i = 0; i < n_deplete + n_target; ++i) {
        rb_dibf *f = i < n_deplete ? deplete[i] : target[i - n_deplete];
        if (!f || f->device != device) { delete e; return rb::fail(RB_ERR_INVALID_ARG, "filter is null or lives on another device"); }
        e->filters.push_back(f);
    }
    e->nd = (uint32_t)n_deplete;
    e->nt = (uint32_t)n_target;
    {
        int khz = 0;
        if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) e->wall_clock_khz = (uint32_t)khz;
        else (void)hipGetLastError();
    }
    e->revcomp_of_n = g_default_revcomp_of_n.load();  // rb_set_default_revcomp_of_n: the one switch that changes results, never an environment variable
    // A/B switches for measurements, read from the environment ONLY when RB_TUNING_ENV=1 says the process is a measurement (profiles/,
    // bench.py sweeps): they pick kernel forms and slice cuts and never change a result; each has a setter in
    // include/readbouncer_amd_tuning.h.  Every override that was ACCEPTED is named in rb_last_warning() of the creating thread, so
    // that a stray variable cannot change behaviour without a trace.  A production process has no environment surface at all.
    std::string accepted;
    auto note = [&](const char *name, const char *v) { accepted += std::string(accepted.empty() ? "" : ", ") + name + "=" + v; };
    const char *tuning_env = std::getenv("RB_TUNING_ENV");
    if (tuning_env && std::atoi(tuning_env) == 1) {
        if (const char *v = std::getenv("RB_MERGE")) {  // rb_engine_set_merge is the API
            if (std::atoi(v) >= 0 && std::atoi(v) <= 2) { e->merge_mode = std::atoi(v); note("RB_MERGE", v); }
        }
        if (const char *v = std::getenv("RB_MERGE_MAX_BYTES")) { e->merge_max_bytes = std::strtoull(v, nullptr, 10); note("RB_MERGE_MAX_BYTES", v); }
        if (const char *v = std::getenv("RB_PHASE_MAX_SLICES")) {
            if (std::atoi(v) >= 1 && std::atoi(v) <= 32) { e->phase_max_slices = (uint32_t)std::atoi(v); note("RB_PHASE_MAX_SLICES", v); }
        }
        if (const char *v = std::getenv("RB_PHASE_SLICE_LOG2")) {
            if (std::atoi(v) >= 1 && std::atoi(v) <= 26) { e->phase_slice_log2 = (uint32_t)std::atoi(v); note("RB_PHASE_SLICE_LOG2", v); }
        }
        if (const char *v = std::getenv("RB_PHASE_N_SLICES")) { e->phase_n_slices = (uint32_t)std::max(0, std::atoi(v)); note("RB_PHASE_N_SLICES", v); }
        if (const char *v = std::getenv("RB_PHASE_XCD_SKEW")) { e->phase_xcd_skew = (uint32_t)std::atoi(v) & 3u; note("RB_PHASE_XCD_SKEW", v); }
        if (const char *v = std::getenv("RB_PHASE_TSKEW_DIV")) { if (std::atoi(v) >= 1) { e->phase_tskew_div = (uint32_t)std::atoi(v); note("RB_PHASE_TSKEW_DIV", v); } }
        if (const char *v = std::getenv("RB_MULTI_ONE_WORD")) { e->multi_one_word = std::atoi(v) != 0; note("RB_MULTI_ONE_WORD", v); }
        if (const char *v = std::getenv("RB_MULTI_WIDE")) { e->multi_wide = std::atoi(v) != 0; note("RB_MULTI_WIDE", v); }
        if (const char *v = std::getenv("RB_MULTI_WIDE_SIX")) { e->multi_wide_six = std::atoi(v) != 0; note("RB_MULTI_WIDE_SIX", v); }
        if (const char *v = std::getenv("RB_MULTI_READS")) {  // rb_engine_set_reads_per_wave is the API
            if (std::atoi(v) >= 0 && std::atoi(v) <= 2) { e->multi_reads = (uint32_t)std::atoi(v); note("RB_MULTI_READS", v); }
        }
        if (const char *v = std::getenv("RB_SIX_TILES")) { e->six_tile_kernel = std::atoi(v); note("RB_SIX_TILES", v); }
        if (const char *v = std::getenv("RB_MICRO_COPY_KERNEL_BYTES")) { e->micro_copy_kernel_bytes = std::strtoull(v, nullptr, 10); note("RB_MICRO_COPY_KERNEL_BYTES", v); }
    }
    rb::set_warning(accepted.empty() ? std::string() : "rb_engine_create: environment overrides in effect: " + accepted);
    hipError_t he = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    e->d_parts.resize(e->filters.size());
    const size_t n_aux = std::min<size_t>(3, e->filters.size() - 1);
    for (size_t i = 0; i < n_aux && he == hipSuccess; ++i) {
        hipStream_t s = nullptr;
        he = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        if (he == hipSuccess) e->aux.push_back(s);
        hipEvent_t ev = nullptr;
        if (he == hipSuccess) he = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (he == hipSuccess) e->join_ev.push_back(ev);
    }
    if (he == hipSuccess) he = hipEventCreateWithFlags(&e->fork_ev, hipEventDisableTiming);
    if (he == hipSuccess) he = hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking);
    e->counted_on_device = true;  // (before the failure path below: rb_engine_destroy takes the count back)
    g_engines_on_device[(unsigned)device % kPlacementDevices].fetch_add(1);
    if (he != hipSuccess) { rb_engine_destroy(e); return rb::fail(RB_ERR_HIP, hipGetErrorString(he)); }
    *out = e;
    return RB_OK;
}

void rb_engine_destroy(rb_engine *e)
{
    if (!e) return;
    if (e->counted_on_device) g_engines_on_device[(unsigned)e->device % kPlacementDevices].fetch_sub(1);
    (void)hipSetDevice(e->device);
    if (e->stream) { (void)hipStreamSynchronize(e->stream); (void)hipStreamDestroy(e->stream); }
    for (auto &p : e->ev_ring) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (hipStream_t s : e->aux) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    for (hipEvent_t ev : e->join_ev) (void)hipEventDestroy(ev);
    if (e->fork_ev) (void)hipEventDestroy(e->fork_ev);
    if (e->copy_stream) { (void)hipStreamSynchronize(e->copy_stream); (void)hipStreamDestroy(e->copy_stream); }
    for (hipEvent_t ev : e->copy_ev) (void)hipEventDestroy(ev);
    for (DevBuf &b : e->d_parts) b.release();
    for (MergedGroup *g : e->merged) delete g;
    for (auto &t : e->thr) {
        if (t.d) (void)hipFree(t.d);
        if (t.ready) (void)hipEventDestroy(t.ready);
        t.host.release();
    }
    for (void *r : e->thr_retired_dev) (void)hipFree(r);
    for (PinnedBuf &h : e->thr_retired_host) h.release();
    for (DevBuf *b : {&e->d_split_ws, &e->d_split_tickets, &e->d_done_count, &e->d_efflens, &e->d_prestatus, &e->d_maxcount, &e->d_seqs, &e->d_offsets, &e->d_lens, &e->d_best,
                      &e->d_decision, &e->d_status})
        b->release();
    e->h_in.release();
    e->h_out.release();
    e->h_done.release();
    delete e;
}

int rb_engine_set_column_shard(rb_engine *e, int rank, int world)
{
    if (!e || world < 1 || rank < 0 || rank >= world) return rb::fail(RB_ERR_INVALID_ARG, "bad shard");
    std::lock_guard<std::mutex> lock(e->mu);
    if (world != e->shard_world) e->merged_planned = false;  // a bin-sharded rank never uses merged tables: planned again at the next call
    e->shard_rank = rank;
    e->shard_world = world;
    return RB_OK;
}

int rb_set_placement_tries(int tries)
{
    if (tries < 0 || tries > 8) return rb::fail(RB_ERR_INVALID_ARG, "placement tries: 0 / 1 (off) to 8");
    g_placement_tries.store(tries);
    return RB_OK;
}

int rb_dibf_placement_cost(const rb_dibf *f, double *trial_seconds, double *settle_seconds, uint64_t *peak_bytes, uint32_t *skipped)
{
    if (!f) return rb::fail(RB_ERR_INVALID_ARG, "null filter");
    if (trial_seconds) *trial_seconds = f->placement_trial_s;
    if (settle_seconds) *settle_seconds = f->placement_settle_s;
    if (peak_bytes) *peak_bytes = f->placement_peak_bytes;
    if (skipped) *skipped = f->placement_skipped;
    return RB_OK;
}

int rb_dibf_placement(const rb_dibf *f, uint32_t *tries, double *kept_gbps, double *worst_gbps)
{
    if (!f) return rb::fail(RB_ERR_INVALID_ARG, "null filter");
    if (tries) *tries = f->placement_tries;
    if (kept_gbps) *kept_gbps = f->placement_gbps;
    if (worst_gbps) *worst_gbps = f->placement_worst_gbps;
    return RB_OK;
}

int rb_set_default_revcomp_of_n(uint32_t ordinal)
{
    if (ordinal != 3 && ordinal != 4) return rb::fail(RB_ERR_INVALID_ARG, "the reverse strand's image of N is ordinal 3 (T) or 4 (N)");
    g_default_revcomp_of_n.store(ordinal);
    return RB_OK;
}

int rb_engine_set_revcomp_of_n(rb_engine *e, uint32_t ordinal)
{
    if (!e || (ordinal != 3 && ordinal != 4)) return rb::fail(RB_ERR_INVALID_ARG, "the reverse strand's image of N is ordinal 3 (T) or 4 (N)");
    std::lock_guard<std::mutex> lock(e->mu);
    e->revcomp_of_n = ordinal;
    return RB_OK;
}

static void plan_merged(rb_engine *e);

// forgets every merged copy; the next call plans again
static void drop_merged(rb_engine *e)
{
    if (!e->merged.empty()) (void)hipDeviceSynchronize();  // a queued kernel may still read a merged copy
    for (MergedGroup *g : e->merged) delete g;
    e->merged.clear();
    e->merged_of.clear();
    e->merged_planned = false;
}

int rb_engine_set_merge(rb_engine *e, int mode)
{
    if (!e || mode < 0 || mode > 2) return rb::fail(RB_ERR_INVALID_ARG, "merge mode is 0 (never), 1 (when it pays) or 2 (always)");
    std::lock_guard<std::mutex> lock(e->mu);
    if (mode != e->merge_mode) {
        e->merge_mode = mode;
        int rc = check_device(e->device);
        if (rc != RB_OK) return rc;
        drop_merged(e);
    }
    return RB_OK;
}

int rb_engine_merge_info(rb_engine *e, uint32_t *n_tables, uint32_t *n_filters, uint64_t *copy_bytes)
{
    if (!e) return rb::fail(RB_ERR_INVALID_ARG, "null engine");
    std::lock_guard<std::mutex> lock(e->mu);
    int rc = check_device(e->device);
    if (rc != RB_OK) return rc;
    if (!e->merged_planned) plan_merged(e);
    uint32_t t = 0, f = 0;
    uint64_t b = 0;
    for (const MergedGroup *g : e->merged) {
        if (g->members.empty()) continue;  // dissolved: no room on the device
        t += 1;
        f += (uint32_t)g->members.size();
        b += (g->n_blocks * hbm_stride(g->width) + 8) * 8;
    }
    if (n_tables) *n_tables = t;
    if (n_filters) *n_filters = f;
    if (copy_bytes) *copy_bytes = b;
    return RB_OK;
}

int rb_engine_set_split_threshold(rb_engine *e, uint32_t max_reads)
{
    if (!e) return rb::fail(RB_ERR_INVALID_ARG, "null engine");
    std::lock_guard<std::mutex> lock(e->mu);
    e->split_threshold = max_reads;
    return RB_OK;
}

int rb_engine_set_split_parts(rb_engine *e, uint32_t max_parts, uint32_t max_shares)
{
    if (!e) return rb::fail(RB_ERR_INVALID_ARG, "null engine");
    std::lock_guard<std::mutex> lock(e->mu);
    e->split_max_parts = max_parts;
    e->split_max_sub = max_shares ? max_shares : 1;
    return RB_OK;
}

int rb_engine_set_fold_decide(rb_engine *e, int enabled)
{
    if (!e) return rb::fail(RB_ERR_INVALID_ARG, "null engine");
    std::lock_guard<std::mutex> lock(e->mu);
    e->fold_decide = enabled != 0;
    return RB_OK;
}

int rb_engine_set_completion_word(rb_engine *e, int enabled)
{
    if (!e) return rb::fail(RB_ERR_INVALID_ARG, "null engine");
    std::lock_guard<std::mutex> lock(e->mu);
    e->completion_word = enabled != 0;
    e->completion_sync_every = enabled > 1 ? (uint32_t)enabled : 256u;  // (measurement: another period, or -- INT_MAX -- none)
    return RB_OK;
}

int rb_engine_set_overlap(rb_engine *e, int enabled)
{
    if (!e) return rb::fail(RB_ERR_INVALID_ARG, "null engine");
    std::lock_guard<std::mutex> lock(e->mu);
    e->overlap = enabled != 0;
    return RB_OK;
}

int rb_engine_set_nt_threshold(rb_engine *e, uint64_t table_bytes)
{
    if (!e) return rb::fail(RB_ERR_INVALID_ARG, "null engine");
    std::lock_guard<std::mutex> lock(e->mu);
    e->nt_threshold_bytes = table_bytes;
    return RB_OK;
}

int rb_engine_set_phased(rb_engine *e, uint64_t min_table_bytes, uint64_t max_table_bytes, uint32_t base_ticks,
                         uint32_t ticks_per_mib, uint32_t min_reads)
{
    if (!e) return rb::fail(RB_ERR_INVALID_ARG, "null engine");
    std::lock_guard<std::mutex> lock(e->mu);
    e->phase_min_bytes = min_table_bytes;
    e->phase_max_bytes = max_table_bytes;
    if (base_ticks || ticks_per_mib) {  // an explicit window length replaces the built-in rule
        e->phase_base_ticks = base_ticks;
        e->phase_ticks_per_mib = ticks_per_mib;
        e->phase_explicit = true;
    } else {
        e->phase_base_ticks = 450;
        e->phase_ticks_per_mib = 0;
        e->phase_explicit = false;
    }
    e->phase_min_reads = min_reads;
    e->phase_overrides.clear();
    e->short_read_kernel = !(min_table_bytes == 0 && max_table_bytes == 0 && base_ticks == 0 && ticks_per_mib == 0 && min_reads == 0);
    return RB_OK;
}

int rb_engine_set_phase_slices(rb_engine *e, uint32_t slice_log2, uint32_t max_slices)
{
    if (!e || slice_log2 > 26 || max_slices < 1 || max_slices > 32) return rb::fail(RB_ERR_INVALID_ARG, "slices of 2^0..26 bytes, 1..32 of them");
    std::lock_guard<std::mutex> lock(e->mu);
    e->phase_slice_log2 = slice_log2;
    e->phase_max_slices = max_slices;
    e->phase_overrides.clear();
    return RB_OK;
}

int rb_engine_set_phase_equal_slices(rb_engine *e, uint32_t n_slices)
{
    if (!e || n_slices > 32) return rb::fail(RB_ERR_INVALID_ARG, "0 (the rule) or 1..32 equal-length slices");
    std::lock_guard<std::mutex> lock(e->mu);
    e->phase_n_slices = n_slices;
    e->phase_overrides.clear();
    return RB_OK;
}

int rb_engine_set_early_decision(rb_engine *e, int enabled)
{
    if (!e) return rb::fail(RB_ERR_INVALID_ARG, "null engine");
    std::lock_guard<std::mutex> lock(e->mu);
    e->early_decision = enabled != 0;
    return RB_OK;
}

int rb_engine_set_phase_xcd_skew(rb_engine *e, uint32_t mode)
{
    if (!e || mode > 3) return rb::fail(RB_ERR_INVALID_ARG, "mode 0..3 (bit 0: slice skew, bit 1: time skew)");
    std::lock_guard<std::mutex> lock(e->mu);
    e->phase_xcd_skew = mode;
    e->phase_overrides.clear();
    return RB_OK;
}

int rb_engine_set_reads_per_wave(rb_engine *e, uint32_t reads)
{
    if (!e || (reads & ~16u) > 2) return rb::fail(RB_ERR_INVALID_ARG, "0 (offsets in registers), 1 or 2 reads per wave (+ 16)");
    std::lock_guard<std::mutex> lock(e->mu);
    e->multi_reads = reads & 3u;
    e->multi_no_inv = (reads & 16u) != 0;
    e->phase_overrides.clear();
    return RB_OK;
}

int rb_engine_set_serial_table_bytes(rb_engine *e, uint64_t table_bytes)
{
    if (!e) return rb::fail(RB_ERR_INVALID_ARG, "null engine");
    std::lock_guard<std::mutex> lock(e->mu);
    e->serial_table_bytes = table_bytes;
    return RB_OK;
}

int rb_engine_set_host_slice_bytes(rb_engine *e, uint64_t slice_bytes)
{
    if (!e) return rb::fail(RB_ERR_INVALID_ARG, "null engine");
    std::lock_guard<std::mutex> lock(e->host_mu);
    e->host_slice_bytes = slice_bytes ? slice_bytes : ~0ULL;
    return RB_OK;
}

int rb_engine_set_timing(rb_engine *e, int enabled)
{
    if (!e) return rb::fail(RB_ERR_INVALID_ARG, "null engine");
    std::lock_guard<std::mutex> lock(e->mu);
    e->timing = enabled != 0;
    e->ev_used = 0;
    return RB_OK;
}

int rb_engine_kernel_time(rb_engine *e, double *total_ms, uint64_t *n_calls)
{
    if (!e) return rb::fail(RB_ERR_INVALID_ARG, "null engine");
    std::lock_guard<std::mutex> lock(e->mu);
    int rc = check_device(e->device);
    if (rc != RB_OK) return rc;
    double sum = 0.0;
    for (size_t i = 0; i < e->ev_used; ++i) {
        RB_HIP(hipEventSynchronize(e->ev_ring[i].second));
        float ms = 0.f;
        RB_HIP(hipEventElapsedTime(&ms, e->ev_ring[i].first, e->ev_ring[i].second));
        sum += ms;
    }
    if (total_ms) *total_ms = sum;
    if (n_calls) *n_calls = e->ev_used;
    e->ev_used = 0;
    return RB_OK;
}

}  // extern "C"

// thresholds by read length for every filter at r and r-0.02 (host doubles -> device table); *tab_out / *len_out = the
// table to hand to the decision kernel.  Nothing here waits for the GPU: a new or longer table gets a fresh device block
// and is uploaded asynchronously on `st` from page-locked memory, the block it replaces is parked until the engine
// goes away (or 64 MiB of parked blocks have piled up, which costs one device synchronisation).
static int ensure_thresholds(rb_engine *e, uint32_t max_len, double r, double conf, hipStream_t st, const uint16_t **tab_out,
                             uint32_t *len_out)
{
    // one table row per read length: 16 Mbp is far beyond any read the callers classify whole (chunk prefixes, 1.5 kbp
    // live cut-off) and keeps the table (4 bytes x filters x length) small
    if (max_len > (1u << 24)) return rb::fail(RB_ERR_UNSUPPORTED, "reads longer than 2^24 bases");
    // NormalCDFInverse(1 - alpha/2) throws std::invalid_argument outside (0,1) (IBF.hpp:284-308): no thresholds, no decisions
    {
        const double p = 1.0 - (1.0 - conf) / 2.0;
        if (!(p > 0.0 && p < 1.0)) return rb::fail(RB_ERR_INVALID_ARG, "significance outside the range NormalCDFInverse accepts");
    }
    const uint32_t need = max_len + 1;
    rb_engine::ThrTable *t = nullptr;
    for (auto &c : e->thr)
        if (c.d && c.r == r && c.conf == conf) t = &c;
    if (!t) {  // take the empty or the least recently used slot
        t = &e->thr[0];
        if (e->thr[1].last_use < t->last_use || (!e->thr[1].d && t->d)) t = &e->thr[1];
        t->len = 0;
        t->r = r;
        t->conf = conf;
    }
    t->last_use = ++e->thr_clock;
    if (t->d && t->len >= need) {
        // uploaded on another stream, possibly still queued there: order this stream behind it
        if (t->ready && t->up_stream != st) RB_HIP(hipStreamWaitEvent(st, t->ready, 0));
        *tab_out = (const uint16_t *)t->d;
        *len_out = t->len;
        return RB_OK;
    }
    uint32_t cap = 1024;
    while (cap < need) cap <<= 1;
    const size_t nf = e->filters.size();
    const size_t row = nf * 2;
    // a new, longer host copy (rows [0, t->len) carried over), then only the new rows are computed.  The copies it replaces
    // are parked, not freed: an upload from the old host block or a kernel reading the old device block may still be queued.
    {
        PinnedBuf bigger;
        int rc = bigger.ensure((size_t)cap * row * 2);
        if (rc != RB_OK) return rc;
        if (t->len) std::memcpy(bigger.p, t->host.p, (size_t)t->len * row * 2);
        if (t->host.p) e->thr_retired_host.push_back(t->host);
        t->host = bigger;
    }
    uint16_t *tab = (uint16_t *)t->host.p;
    const double r2 = r - 0.02;  // "conf.error_rate -= 0.02" (adaptive_sampling.hpp:55, classify.hpp:67)
    for (uint32_t len = t->len; len < cap; ++len) {
        for (size_t fi = 0; fi < nf; ++fi) {
            const uint64_t k = e->filters[fi]->geo.kmer_size;
            tab[((size_t)len * nf + fi) * 2 + 0] = threshold_u16(len, k, r, conf);
            tab[((size_t)len * nf + fi) * 2 + 1] = threshold_u16(len, k, r2, conf);
        }
    }
    if (t->d) {
        e->thr_retired_dev.push_back(t->d);
        t->d = nullptr;
    }
    e->thr_retired_bytes += 2 * t->bytes;
    if (e->thr_retired_bytes > ((size_t)64 << 20)) {
        RB_HIP(hipDeviceSynchronize());
        for (void *x : e->thr_retired_dev) (void)hipFree(x);
        for (PinnedBuf &h : e->thr_retired_host) h.release();
        e->thr_retired_dev.clear();
        e->thr_retired_host.clear();
        e->thr_retired_bytes = 0;
    }
    t->bytes = (size_t)cap * row * 2;
    RB_HIP(hipMalloc(&t->d, (size_t)cap * row * 2));
    RB_HIP(hipMemcpyAsync(t->d, tab, (size_t)cap * row * 2, hipMemcpyHostToDevice, st));
    if (!t->ready) RB_HIP(hipEventCreateWithFlags(&t->ready, hipEventDisableTiming));
    RB_HIP(hipEventRecord(t->ready, st));
    t->up_stream = st;
    t->len = cap;
    *tab_out = (const uint16_t *)t->d;
    *len_out = cap;
    return RB_OK;
}

// Kernel geometry of one filter for a batch: the rank's word columns (bin-sharded operation), lanes per block, words
// per lane, counter planes, column slices, and the form of K1 (throughput, or latency with its waves / workgroups per
// read).  false = this rank owns no column of the filter.
static bool plan_geometry(const rb_engine *e, const rb_dibf *f, size_t n_reads, uint32_t max_len, CountLaunch &a)
{
    const uint32_t W = (uint32_t)f->geo.bin_width;
    // bin-sharded operation: contiguous word-column range of every block per rank
    uint32_t per = (W + e->shard_world - 1) / e->shard_world;
    if (e->shard_world > 1 && (per & 1)) ++per;  // keep 16-byte alignment of the slices
    a.col_begin = std::min<uint32_t>(W, per * e->shard_rank);
    a.col_end = std::min<uint32_t>(W, a.col_begin + per);
    const uint32_t Weff = a.col_end - a.col_begin;
    const uint32_t kmers = max_len >= f->geo.kmer_size ? max_len - (uint32_t)f->geo.kmer_size + 1 : 0;
    a.planes = kmers <= 1023 ? 10 : 16;
    a.nt = f->geo.n_blocks * f->stride * 8 > e->nt_threshold_bytes;
    if (Weff == 0) return false;
    if (Weff > 64 && (a.col_begin % 2 == 0)) {  // 16 bytes per lane; odd widths end in a one-column lane
        a.wpl = 2; a.lg = 6;
    } else {
        a.wpl = 1; a.lg = 0;
        while ((1u << a.lg) < std::min<uint32_t>(Weff, 64)) ++a.lg;
    }
    const uint32_t slice_words = (1u << a.lg) * a.wpl;
    a.n_slices = (Weff + slice_words - 1) / slice_words;
    // micro-batches cannot fill 256 CUs with one wave per read: spread each read over a workgroup, or several
    a.split_waves = 0;
    if (e->split_threshold && (uint64_t)n_reads * a.n_slices <= e->split_threshold && f->geo.n_hash == 3)
        a.split_waves = split_waves_limit(a.wpl, a.planes, kmers, a.lg);
    a.phase = PhaseCfg{0, 0, 0, 0, 0};
    a.multi_reads = 0;
    a.multi_inv = 0;
    a.multi_tiles = 0;
    a.short_only = kmers <= 256 ? 1 : kmers <= 512 ? 2 : 0;
    // 257-384 k-mers (360 bp reads): one round of six tiles per strand instead of two rounds of four
    // (profiles/r03/window_sweep.txt: with the bounds-checked gathers and its own window length the six-tile kernel takes 28 %
    // less time than two rounds of four tiles on one-word 10 MB filters, 15 % on a one-word 20 MB table, 22 % on two-word blocks)
    if (a.short_only == 2 && kmers <= 384 && e->six_tile_kernel) a.short_only = 3;
    const uint64_t table_bytes = f->geo.n_blocks * f->stride * 8;
    // three- and four-word blocks: only the both-strands build of the phased kernel (reads of up to 512 k-mers, whole blocks)
    const bool wide_short = a.lg == 2 && a.planes <= 10 && kmers <= 512 && a.col_begin == 0 && (a.col_end == 3 || a.col_end == 4) &&
                            f->stride == 4 && W == a.col_end;
    if (a.split_waves < 2 && f->geo.n_hash == 3 && a.wpl == 1 && (a.lg <= 1 || wide_short) && a.n_slices == 1 && table_bytes < (1ull << 31)) {
        if (wide_short) a.short_only = (kmers <= 256 && e->six_tile_kernel != 2) ? 5 : 4;  // (RB_SIX_TILES=2: experiments without the four-tile build)
        // the phased kernels take a lookup's slice from its byte offset by a shift: block strides that are a power of two
        // only (hbm_stride gives one to every filter narrower than 16 words; a bin-sharded rank can reach lg <= 1 on a wider
        // filter, e.g. 3072 bins over 24 ranks -- stride 48 -- and keeps the plain kernel)
        const bool stride_pow2 = (f->stride & (f->stride - 1)) == 0;
        // the kernel shape this batch takes (CountLaunch::short_only is the kernels' own code for it: 1 / 3 / 2 = at most 256 / 384 /
        // 512 k-mers per read, 5 / 4 = the same for stride-4 blocks held by one lane; three-word blocks have builds of their own)
        PhaseShape shape = PhaseShape::General;
        if (a.planes <= 10) {
            switch (a.short_only) {
            case 1: shape = PhaseShape::FourTiles; break;
            case 2: shape = PhaseShape::Rounds; break;
            case 3: shape = PhaseShape::SixTiles; break;
            case 4: shape = a.col_end == 3 ? PhaseShape::Wide3Rounds : PhaseShape::WideRounds; break;
            case 5: shape = a.col_end == 3 ? PhaseShape::Wide3FourTiles : PhaseShape::WideFourTiles; break;
            default: break;
            }
        }
        a.phase_shape = (int)shape;
        const bool in_rule_range = table_bytes >= phase_shape_min_bytes(shape, a.lg, phase_fill(shape, kmers)) &&
                                   (double)table_bytes <= (double)phase_shape_max_bytes(shape, a.lg) * (phase_fill(shape, kmers) >= 0.8 ? 1.0 : phase_fill(shape, kmers)) &&
                                   n_reads >= phase_min_reads_for(table_bytes);
        if (e->phase_max_bytes && table_bytes >= e->phase_min_bytes && table_bytes <= e->phase_max_bytes &&
            n_reads >= e->phase_min_reads && stride_pow2 && (e->phase_explicit || in_rule_range)) {
            const uint32_t slice_log2 = e->phase_slice_log2 ? e->phase_slice_log2 : phase_slice_log2(shape, a.lg, table_bytes, kmers);
            uint32_t sh = 0;
            while (slice_log2 >= 6 && (f->stride * 8) << (sh + 1) <= (1ull << slice_log2)) ++sh;  // (< 6: as small as max_slices allows)
            while (((f->geo.n_blocks + (1ull << sh) - 1) >> sh) > e->phase_max_slices) ++sh;
            uint32_t n_sl = (uint32_t)((f->geo.n_blocks + (1ull << sh) - 1) >> sh);
            // the builds that keep the offsets in LDS (two-word blocks of up to 2^21 - 1 blocks, reads of up to 384 k-mers): planned here because the
            // window belongs to the build (rb_phase_plan.h, phase_multi_window_ticks)
            // (lg 1: short_only 1 / 3 = at most 256 / 384 k-mers; lg 2, blocks of three and four words: short_only 5 = at most 256, 4 = at most 512 --
            // up to 384 of them fit one round of six tiles)
            const int multi_tiles = a.lg <= 1 ? (a.short_only == 1 ? 4 : a.short_only == 3 ? 6 : 0)
                                    : (a.lg == 2 && wide_short) ? (a.short_only == 5 ? 4 : (a.short_only == 4 && kmers <= 384 && e->multi_wide_six) ? 6 : 0) : 0;
            const bool multi_build = e->multi_reads && multi_tiles && a.planes <= 10 && a.col_begin == 0 &&
                                     f->geo.n_blocks < (1ull << (a.lg == 0 ? 22 : 21)) - 1 &&  // (block numbers of 21 bits; one-word blocks: 22, rb_kernels.hip kPackBits1)
                                     ((a.lg == 1 && a.col_end == 2 && f->stride == 2) || (a.lg == 2 && e->multi_wide && f->stride == 4) ||
                                      (a.lg == 0 && e->multi_one_word && a.col_end == 1 && f->stride == 1 && W == 1));
            // the four-word one-lane builds: slices of equal length, fewer than the 4 MiB ones (rb_phase_plan.h, phase_equal_slices)
            uint64_t blocks_per_slice = 0;
            const bool one_word_rule = f->stride == 1 && a.lg == 0 && phase_equal_slices_one_word(shape, a.lg, slice_log2, table_bytes) != 0;
            // ... and the one- and two-word LDS-offset builds where their rule asks for 4 MiB slices: equal ones SHORTER than an L2 (phase_multi_equal_slices)
            const uint32_t multi_equal = (multi_build && a.lg <= 1 && !e->phase_n_slices) ? phase_multi_equal_slices(shape, a.lg, slice_log2, table_bytes, kmers) : 0;
            if (!e->phase_slice_log2 && ((f->stride == 4 && a.lg == 2) || one_word_rule || multi_equal || e->phase_n_slices)) {
                uint32_t want = (f->stride == 4 && a.lg == 2) ? phase_equal_slices(shape, slice_log2, table_bytes)
                                : one_word_rule ? phase_equal_slices_one_word(shape, a.lg, slice_log2, table_bytes) : multi_equal;  // (one-word tables from 50 MiB on)
                if (e->phase_n_slices) want = e->phase_n_slices;  // (RB_PHASE_N_SLICES, measurements: profiles/r04/slice_count_sweep.txt)
                if (want >= 1 && want <= e->phase_max_slices && (want < n_sl || e->phase_n_slices || multi_equal)) {  // (an explicit count and the two-word rule may also ask for MORE slices than the 4 MiB cut)
                    // the kernels carry blocks-per-slice in 31 bits and the slice's span in BYTES in 31 bits as well (bit 31 is the
                    // flag of this form): a slice of 2 GiB or more keeps the power-of-two cut, whose span is a shift (ADVICE r4 --
                    // a truncated span would read lookups beyond it as "no lookup" and the counts would be silently short)
                    const uint64_t bps = (f->geo.n_blocks + want - 1) / want;
                    if (bps < (1ull << 31) && bps * f->stride * 8 < (1ull << 31)) {
                        blocks_per_slice = bps;
                        n_sl = (uint32_t)((f->geo.n_blocks + blocks_per_slice - 1) / blocks_per_slice);
                    }
                }
            }
            uint64_t ticks = e->phase_explicit ? e->phase_base_ticks + (table_bytes >> 20) * e->phase_ticks_per_mib
                             : (blocks_per_slice && multi_equal) ? phase_multi_equal_ticks(shape, n_sl, table_bytes, kmers)
                             : (blocks_per_slice && one_word_rule && !e->phase_n_slices) ? phase_equal_slices_one_word_ticks(shape, n_sl, table_bytes, kmers)
                             : blocks_per_slice ? phase_equal_slices_ticks(shape, a.lg, n_sl, kmers)
                                                : phase_window_ticks(shape, a.lg, slice_log2, n_sl, kmers);
            a.phase_rule_ticks = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ticks, 100), 2000);
            if (!e->phase_explicit)  // a window measured on this device for exactly this table, shape and slice size (rb_engine_calibrate)
                for (const rb_engine::PhaseOverride &o : e->phase_overrides)
                    if (o.table_bytes == table_bytes && o.stride == f->stride && o.shape == (int)shape && o.lg == a.lg && o.slice_log2 == slice_log2) ticks = o.ticks;
            if (multi_build && !e->phase_explicit && !(blocks_per_slice && multi_equal)) {  // (the equal cut's window is the build's own already)
                const uint64_t scaled = phase_multi_window_ticks(shape, slice_log2, n_sl, kmers, a.phase_rule_ticks);
                if (ticks == a.phase_rule_ticks) ticks = scaled;  // (a window measured by rb_engine_calibrate for this table stays)
                a.phase_rule_ticks = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(scaled, 100), 2000);
            }
            ticks = std::min<uint64_t>(std::max<uint64_t>(ticks, 100), 2000);
            a.phase_slice_log2 = slice_log2;
            a.phase_ticks = (uint32_t)ticks;
            ticks = std::max<uint64_t>(2, ticks * e->wall_clock_khz / 100000);  // 10 ns units -> ticks of this device's clock (>= 2: 2^32 / ticks must fit 32 bits)
            a.phase.shift = blocks_per_slice ? (0x80000000u | (uint32_t)blocks_per_slice) : sh;  // bit 31: the low bits are blocks per slice, any number
            a.phase.n_slices = n_sl;
            a.phase.inv_ticks = (uint32_t)((1ull << 32) / ticks);
            a.phase.xcd_skew = e->phase_xcd_skew & 1u;
            a.phase.tskew = (e->phase_xcd_skew & 2u) ? (uint32_t)(ticks / e->phase_tskew_div) : 0u;  // each XCD's windows start an eighth of a window after the previous one's
            a.phase_slice_bytes = blocks_per_slice ? blocks_per_slice * f->stride * 8 : (f->stride * 8) << sh;
            if (multi_build) {
                a.multi_reads = (int)e->multi_reads;
                a.multi_tiles = multi_tiles;
            }
        } else if ((a.lg == 0 || (shape != PhaseShape::General && a.col_begin == 0 && a.col_end == 2 && f->stride == 2) ||
                    ((shape == PhaseShape::WideFourTiles || shape == PhaseShape::Wide3FourTiles) && phase_fill(shape, kmers) >= 0.8 &&
                     table_bytes < phase_shape_min_bytes(shape, a.lg, 1.0))) && e->short_read_kernel) {
            // blocks outside the phased range still take that kernel for its both-strands-in-one-round path (one-word blocks
            // always; two-word blocks for reads of up to 512 k-mers; small three- and four-word tables for reads of up to 256:
            // 2 MiB 7.2 -> 6.6 ms, 4 MiB 7.7 -> 6.9): one "slice" that holds every offset, no clock, no waiting
            a.phase.shift = 31;
            a.phase.n_slices = 1;
            a.phase.inv_ticks = 0;
        }
    }
    // columns -> members for the one-lane-per-block builds of the phased kernel: a filter on its own is one member
    a.narrow = NarrowMerge{};
    a.narrow.n = 1;
    a.narrow.bit_begin[0] = 0;
    a.narrow.bit_end[0] = (uint32_t)f->geo.n_bins;
    for (uint32_t c = 0; c < 4; ++c)
        a.narrow.col_bits[c] = c >= W ? 0u : (c + 1 == W && (f->geo.n_bins & 63)) ? (uint32_t)(f->geo.n_bins & 63) : 64u;
    a.split_parts = 1;
    a.split_sub = 1;
    if (a.split_waves >= 2)
        a.split_parts = split_parts_plan(a.wpl, a.planes, kmers, a.lg, (uint32_t)n_reads * a.n_slices, e->split_max_parts,
                                         e->split_max_sub, &a.split_waves, &a.split_sub);
    return true;
}

static int ensure_split_ws(rb_engine *e, CountLaunch &a, size_t n_filters, hipStream_t st);

// micro-batches: filters of equal kernel geometry share a launch; beyond that, the filters whose latency form applies
// share ONE launch of the mixed-geometry kernel when the batch is small or their wave counts are close (the wide deplete
// filter then hides the narrow targets: 52 -> 41 us per call on config 4).  With many reads per call a shared workgroup
// size wastes LDS and wave slots on the filters that need few waves, so those keep their own launches.
static int launch_fused_groups(rb_engine *e, const std::vector<CountLaunch> &pending, const std::vector<uint32_t> &pending_fi,
                               uint16_t *maxcount, hipStream_t st)
{
    int lo = 1 << 30, hi = 0;
    for (const CountLaunch &b : pending)
        if (b.split_waves >= 2) { lo = std::min(lo, b.split_waves); hi = std::max(hi, b.split_waves); }
    const bool mix = hi > 0 && (pending[0].n_reads <= 64 || hi <= 2 * lo);
    std::vector<bool> done(pending.size(), false);
    for (size_t i = 0; i < pending.size(); ++i) {
        if (done[i]) continue;
        CountLaunch g = pending[i];
        g.phase.n_slices = 0;  // fused launches keep the plain gathers
        const bool split = g.split_waves >= 2;
        g.n_fused = 0;
        int want_waves = 0, grid_parts = 1;
        bool same = true, multi = false;
        for (size_t j = i; j < pending.size() && g.n_fused < (int)kMaxFused; ++j) {
            const CountLaunch &b = pending[j];
            if (done[j] || (b.split_waves >= 2) != split || b.planes != g.planes || b.f.n_hash != g.f.n_hash) continue;
            const bool same_geometry = b.lg == g.lg && b.wpl == g.wpl && b.nt == g.nt;
            if (!same_geometry && !(split && mix)) continue;
            if (split && same_geometry && (b.split_waves != g.split_waves) && !mix) continue;
            g.fused_f[g.n_fused] = b.f;
            g.fused_col_begin[g.n_fused] = b.col_begin;
            g.fused_col_end[g.n_fused] = b.col_end;
            g.fused_out_offset[g.n_fused] = pending_fi[j];
            g.fused_geom[g.n_fused] = geom_code(b.lg, b.wpl, b.nt);
            g.fused_parts[g.n_fused] = (uint8_t)(b.split_parts > 1 ? b.split_parts : 1);
            g.fused_sub[g.n_fused] = (uint8_t)(b.split_sub > 1 ? b.split_sub : 1);
            same &= same_geometry;
            multi |= b.split_parts > 1;
            want_waves = std::max(want_waves, b.split_waves);
            grid_parts = std::max(grid_parts, b.split_parts);
            ++g.n_fused;
            done[j] = true;
        }
        if (split) {
            // a filter spread over several workgroups was planned for 2 x 4 waves (= kSplitAnyWaves); filters that
            // keep one workgroup adapt to any even wave count
            if (!same) g.split_waves = std::min(want_waves, kSplitAnyWaves);
            else if (multi) g.split_waves = kSplitAnyWaves;
            g.grid_parts = grid_parts;
        }
        g.out = maxcount;  // per-filter column offsets travel in the set
        int rc = ensure_split_ws(e, g, (size_t)g.n_fused, st);
        if (rc != RB_OK) return rc;
        RB_HIP(launch_ibf_count_max(g, st));
    }
    return RB_OK;
}

// workspace of a multi-workgroup latency launch: partial counters + arrival counters (zero between launches; a fresh
// or grown allocation is zeroed on the stream that will use it)
static int ensure_split_ws(rb_engine *e, CountLaunch &a, size_t n_filters, hipStream_t st)
{
    if (a.split_waves < 2) return RB_OK;
    if (a.grid_parts < 1) a.grid_parts = a.split_parts > 1 ? a.split_parts : 1;
    if (a.grid_parts <= 1) return RB_OK;
    const size_t items = n_filters * (size_t)a.n_reads * a.n_slices;
    const size_t np = a.planes <= 10 ? 10 : 16;
    int rc = e->d_split_ws.ensure(items * (size_t)a.grid_parts * 2 * 2 * np * 64 * 8);
    if (rc != RB_OK) return rc;
    const void *old = e->d_split_tickets.p;
    const size_t old_cap = e->d_split_tickets.cap;
    rc = e->d_split_tickets.ensure(items * 4);
    if (rc != RB_OK) return rc;
    if (e->d_split_tickets.p != old || e->d_split_tickets.cap != old_cap || e->tickets_dirty)
        RB_HIP(hipMemsetAsync(e->d_split_tickets.p, 0, e->d_split_tickets.cap, st));
    e->tickets_dirty = true;  // until the call that uses them returns RB_OK
    a.split_ws = (uint64_t *)e->d_split_ws.p;
    a.split_tickets = (uint32_t *)e->d_split_tickets.p;
    return RB_OK;
}

// what the decision needs besides the raw maxima (the threshold table is made on `st` when this (max_len, r, conf) is new)
static int decide_params(rb_engine *e, uint32_t max_len, double r, double conf, hipStream_t st, uint16_t *maxcount_copy,
                         uint32_t n_parts, uint64_t part_stride, DecideParams *out, uint32_t *done_flag = nullptr, uint32_t done_seq = 0,
                         size_t n_reads = 0)
{
    DecideParams P{};
    int rc = ensure_thresholds(e, max_len, r, conf, st, &P.thr, &P.thr_len);
    if (rc != RB_OK) return rc;
    P.nd = e->nd;
    P.nt = e->nt;
    for (size_t i = 0; i < e->filters.size(); ++i) P.k[i] = (uint32_t)e->filters[i]->geo.kmer_size;
    P.max_len = max_len;
    P.maxcount_copy = maxcount_copy;
    P.n_parts = n_parts ? n_parts : 1;
    P.part_stride = part_stride;
    if (done_flag) {
        P.done_flag = done_flag;
        P.done_seq = done_seq;
        if (n_reads > 256) {  // a decision kernel of several workgroups counts its arrivals
            const void *old = e->d_done_count.p;
            if ((rc = e->d_done_count.ensure(64)) != RB_OK) return rc;
            if (e->d_done_count.p != old || e->done_dirty) RB_HIP(hipMemsetAsync(e->d_done_count.p, 0, 64, st));
            e->done_dirty = true;  // until the call returns RB_OK
            P.done_count = (uint32_t *)e->d_done_count.p;
        }
    }
    *out = P;
    return RB_OK;
}

static int run_decide(rb_engine *e, const uint16_t *d_maxcount, const uint32_t *d_lens, const uint8_t *d_pre_status,
                      size_t n_reads, uint32_t max_len, double r, double conf, int mode, int32_t *d_best,
                      uint8_t *d_decision, uint8_t *d_status, hipStream_t st, uint16_t *maxcount_copy = nullptr,
                      uint32_t n_parts = 1, uint64_t part_stride = 0, uint32_t *done_flag = nullptr, uint32_t done_seq = 0)
{
    DecideParams P{};
    int rc = decide_params(e, max_len, r, conf, st, maxcount_copy, n_parts, part_stride, &P, done_flag, done_seq, n_reads);
    if (rc != RB_OK) return rc;
    RB_HIP(launch_decide(P, d_maxcount, d_lens, d_pre_status, (uint32_t)n_reads, mode, d_best, d_decision, d_status, st));
    return RB_OK;
}

extern "C" {

int rb_host_alloc(size_t bytes, void **out)
{
    if (!out) return rb::fail(RB_ERR_INVALID_ARG, "null out");
    *out = nullptr;
    RB_HIP(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return RB_OK;
}

void rb_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int rb_classify_batch_device(rb_engine *e, const void *d_seqs, const void *d_offsets, const void *d_lens, size_t n_reads,
                             uint32_t max_len, double error_rate, double significance, int mode, void *d_maxcount,
                             void *d_best_target, void *d_decision, void *d_status, void *stream)
{
    rb_batch_desc desc;
    std::memset(&desc, 0, sizeof desc);
    desc.d_seqs = d_seqs;
    desc.d_offsets = d_offsets;
    desc.d_lens = d_lens;
    desc.n_items = n_reads;
    desc.max_len = max_len;
    return rb_classify_batch_device_ex(e, &desc, error_rate, significance, mode, d_maxcount, d_best_target, d_decision,
                                       d_status, stream);
}

}  // extern "C"

// K1 time estimates behind the "when it pays" rule of plan_merged, in ms per 1 M reads of 250 bp (238 k-mers, three hash
// functions), from profiles/r03/slice_size.txt and merged_tables.txt: what a table of `mib` MiB costs
//  - with the plain gathers (also the merged kernel's): L2-resident up to 4 MiB, then the hit rate falls with 4 MiB / table
//    until the kernel sits at the chip's fabric-request wall (~56 G requests/s) from about 64 MiB on, whatever the block width;
//  - with the clock-phased gathers (one- and two-word blocks, 6-128 MiB): 6.6 + 0.125 per MiB (two-word 7.0 + 0.15).
static double est_plain_ms(double mib)
{
    static const double pts[][2] = {{0, 6.1}, {4, 6.5}, {6, 8.7}, {8, 12.0}, {12, 15.8}, {16, 17.9}, {24, 19.9}, {32, 21.5}, {48, 23.2}, {64, 24.0}, {128, 25.0}};
    const int n = (int)(sizeof(pts) / sizeof(pts[0]));
    if (mib >= pts[n - 1][0]) return pts[n - 1][1];
    int i = 1;
    while (pts[i][0] < mib) ++i;
    return pts[i - 1][1] + (pts[i][1] - pts[i - 1][1]) * (mib - pts[i - 1][0]) / (pts[i][0] - pts[i - 1][0]);
}

static double est_filter_ms(const rb_engine *e, const rb_dibf *f)
{
    const uint64_t bytes = f->geo.n_blocks * f->stride * 8;
    const double mib = (double)bytes / 1048576.0;
    const bool phased = f->geo.n_hash == 3 && f->geo.bin_width <= 2 && e->phase_max_bytes && bytes >= e->phase_min_bytes &&
                        bytes <= std::min<uint64_t>(e->phase_max_bytes, phase_shape_max_bytes(PhaseShape::FourTiles, f->geo.bin_width == 1 ? 0 : 1));
    // three- and four-word blocks (stride 4) in the both-strands build, 4.5-48 MiB: 8 MiB 9.4, 16 MiB 11.4, 24 MiB 12.7, 40 MiB 17.2
    if (f->geo.n_hash == 3 && f->stride == 4 && e->phase_max_bytes && bytes >= (9ull << 19) && bytes <= (48ull << 20)) return 8.0 + 0.23 * mib;
    if (!phased) return est_plain_ms(mib);
    return f->geo.bin_width == 1 ? 6.3 + 0.115 * mib : 6.3 + 0.125 * mib;  // (session 55: 8 MiB 7.0 / 6.7, 32 MiB 9.5 / 10.5, 64 MiB 14.1 / 13.8)
}

// Which filters share a merged table.  Candidates: three hash functions, blocks of at most 8 words, equal noOfBlocks and k; at
// most 16 words (one 128-byte line) per merged block.  "When it pays" (mode 1): when the members one after the other are
// estimated to take longer than ONE pass of the merged kernel over the merged table (+ 0.3 ms per member for its maxima):
//   README shape (three one-word targets of 10 MiB + a two-word deplete filter of 20 MiB): 33.6 against 25.5 -> merged (measured
//   29.9 -> 40.8 M reads/s); three one-word filters of 10 MiB: 23.7 against 23.6 -> apart; two filters of 12 or 24 MB: apart
//   (measured 0.88 x merged); two filters beyond the phased range (each at the request wall on its own): merged (1.9-2.0 x
//   measured at 48-540 MB per table, 3.0 x for three); two filters of 1-2 MB whose merged copy still fits an L2: merged (1.65 x).
static void plan_merged(rb_engine *e)
{
    drop_merged(e);
    e->merged_planned = true;
    e->merged_of.assign(e->filters.size(), -1);
    if (e->merge_mode == 0 || e->shard_world != 1) return;
    for (size_t i = 0; i < e->filters.size(); ++i) {
        if (e->merged_of[i] >= 0) continue;
        const rb_ibf_info &gi = e->filters[i]->geo;
        if (gi.n_hash != 3 || gi.bin_width > 8) continue;
        std::vector<uint32_t> members{(uint32_t)i};
        uint64_t width = gi.bin_width;
        for (size_t j = i + 1; j < e->filters.size() && members.size() < kMaxMerged; ++j) {
            const rb_ibf_info &gj = e->filters[j]->geo;
            if (e->merged_of[j] >= 0 || gj.n_hash != 3 || gj.bin_width > 8 || gj.n_blocks != gi.n_blocks || gj.kmer_size != gi.kmer_size ||
                width + gj.bin_width > 16)
                continue;
            members.push_back((uint32_t)j);
            width += gj.bin_width;
        }
        if (members.size() < 2) continue;
        // Bit-packed layout: when the members' bins, put side by side bit to bit, need fewer word columns than their whole words do AND
        // that brings the block down to the four words one lane can hold, the merged table is a "filter" of that width for the
        // both-strands builds of the phased kernel (README shape: 2 + 1 + 1 + 1 = 5 words -> 243 bits = 4 words: a 40 MB table in ten
        // L2-sized slices instead of an 80 MB one at the fabric's line rate).  Per-bin counting does not care where a member's bins
        // start; the per-member maxima are taken over bit ranges (NarrowMerge / MergeMap).
        uint64_t bins_total = 0;
        for (uint32_t m : members) bins_total += e->filters[m]->geo.n_bins;
        const uint64_t packed_width = (bins_total + 63) / 64;
        const bool packed = packed_width < width && packed_width <= 4 && members.size() <= kMaxNarrow;
        if (packed) width = packed_width;
        double apart = 0.0;
        for (uint32_t m : members) apart += est_filter_ms(e, e->filters[m]);
        // (a merged table of two to four words is served by the phased kernel like a filter of that width)
        const double merged_mib = (double)(gi.n_blocks * hbm_stride(width) * 8) / 1048576.0;
        const bool narrow_phased = e->phase_max_bytes && ((width == 2 && merged_mib >= 1.25 && merged_mib <= 96.0) ||
                                                          ((width == 3 || width == 4) && merged_mib <= 48.0));
        const double together = (narrow_phased ? (width == 2 ? 6.3 + 0.125 * merged_mib : 8.0 + 0.23 * merged_mib) : est_plain_ms(merged_mib)) +
                                0.3 * (double)members.size();
        if (e->merge_mode == 1 && apart <= 1.05 * together) continue;
        if ((gi.n_blocks * hbm_stride(width) + 8) * 8 > e->merge_max_bytes) continue;
        MergedGroup *g = new (std::nothrow) MergedGroup();
        if (!g) return;
        g->members = members;
        g->width = width;
        g->packed = packed;
        uint32_t at = 0;
        for (uint32_t m : members) {
            g->bit_begin.push_back(at);
            at += packed ? (uint32_t)e->filters[m]->geo.n_bins : (uint32_t)e->filters[m]->geo.bin_width * 64u;
        }
        g->n_blocks = gi.n_blocks;
        for (uint32_t m : members) e->merged_of[m] = (int)e->merged.size();
        e->merged.push_back(g);
    }
}

// the merged copy of a group, made (or made again after a member changed) on `st`.  kMergeNoMemory: the device has no room
// for the copy -- the caller dissolves the group and its members are served one by one as before.
static constexpr int kMergeNoMemory = -1000;
static bool merged_table_fresh(const rb_engine *e, const MergedGroup *g)
{
    const MergedTable *t = g->tab.get();
    if (!t || !t->d_words || t->versions.size() != g->members.size() || g->dev.words != t->d_words) return false;
    for (size_t i = 0; i < g->members.size(); ++i)
        if (t->versions[i] != e->filters[g->members[i]]->version.load()) return false;
    return true;
}

// caller holds merged_rw(e->device) EXCLUSIVELY: no call of any engine on this device is between its freshness check and its launch
static int ensure_merged_table(rb_engine *e, MergedGroup *g, hipStream_t st)
{
    if (!g->tab) {
        std::vector<const rb_dibf *> key;
        for (uint32_t m : g->members) key.push_back(e->filters[m]);
        g->tab = merged_table_for(e->device, key, g->bit_begin, g->width);
        if (!g->tab) return kMergeNoMemory;
    }
    MergedTable *t = g->tab.get();
    // this engine's view of the block layout
    g->map = MergeMap{};
    g->map.n = (uint32_t)g->members.size();
    g->map.width = (uint32_t)g->width;
    for (size_t i = 0; i < g->members.size(); ++i) {
        g->map.bit_begin[i] = g->bit_begin[i];
        g->map.bit_end[i] = g->bit_begin[i] + (uint32_t)e->filters[g->members[i]]->geo.n_bins;
        g->map.out_offset[i] = g->members[i];
    }
    bool fresh = t->d_words != nullptr && t->versions.size() == g->members.size();
    for (size_t i = 0; fresh && i < g->members.size(); ++i) fresh = t->versions[i] == e->filters[g->members[i]]->version.load();
    if (!fresh) {
        if (!t->d_words) {
            t->stride = hbm_stride(g->width);
            t->n_blocks = g->n_blocks;
            const bool twin = ((t->stride == 2 && g->width == 2) || (t->stride == 4 && g->width >= 3 && g->width <= 4)) && t->n_blocks < (1ull << 21) - 1;
            if (hipMalloc((void **)&t->d_words, (t->n_blocks * t->stride + 8) * 8 * (twin ? 2 : 1)) != hipSuccess) {
                (void)hipGetLastError();
                t->d_words = nullptr;
                g->tab.reset();
                return kMergeNoMemory;
            }
            t->d_inv = twin ? t->d_words + (t->n_blocks * t->stride + 8) : nullptr;
        } else {
            RB_HIP(hipDeviceSynchronize());  // made again: a kernel of an earlier call (of any engine, on any stream) may still read the old copy
        }
        RB_HIP(hipMemsetAsync(t->d_words, 0, (t->n_blocks * t->stride + 8) * 8, st));
        t->versions.assign(g->members.size(), 0);
        for (size_t i = 0; i < g->members.size(); ++i) {
            rb_dibf *f = e->filters[g->members[i]];
            t->versions[i] = f->version.load();
            RB_HIP(launch_merge_bits(f->d_words, (uint32_t)f->stride, (uint32_t)f->geo.bin_width, (uint32_t)f->geo.n_bins, t->d_words,
                                     (uint32_t)t->stride, g->bit_begin[i], t->n_blocks, st));
        }
        if (t->d_inv) RB_HIP(launch_invert_words(t->d_words, t->d_inv, t->n_blocks * t->stride + 8, st));
        rb_ibf_info geo = e->filters[g->members[0]]->geo;  // noOfBlocks, k, h of the members
        geo.bin_width = g->width;
        geo.n_bins = g->width * 64;
        int rc = make_dev_desc(geo, t->d_words, t->stride, &t->dev);
        if (rc != RB_OK) return rc;
        // made once (and again when a member changed): wait here, so that calls on other streams and of other engines find it complete
        RB_HIP(hipStreamSynchronize(st));
    }
    g->stride = t->stride;
    g->dev = t->dev;
    return RB_OK;
}

// host_maxcount: optional pinned host destination for a copy of the maxcount rows, written by the decision kernel
static int classify_device_impl(rb_engine *e, const rb_batch_desc *desc, double error_rate, double significance, int mode,
                                void *d_maxcount, void *d_best_target, void *d_decision, void *d_status, void *stream,
                                uint16_t *host_maxcount, uint32_t *done_flag = nullptr, uint32_t done_seq = 0)
{
    if (!e || !desc) return rb::fail(RB_ERR_INVALID_ARG, "null engine or descriptor");
    if (mode != RB_MODE_CHECK_UNBLOCK && mode != RB_MODE_CLASSIFY_CHUNK && mode != RB_MODE_CLASSIFY_ANY)
        return rb::fail(RB_ERR_INVALID_ARG, "unknown mode");
    const size_t n_reads = desc->n_items;
    if (n_reads >= (1ULL << 31)) return rb::fail(RB_ERR_INVALID_ARG, "batch too large");
    if (n_reads == 0) return RB_OK;
    const void *d_seqs = desc->d_seqs, *d_offsets = desc->d_offsets;
    if (!d_seqs || !d_offsets || !desc->d_lens) return rb::fail(RB_ERR_INVALID_ARG, "null input buffer");
    if ((desc->d_nmask == nullptr) != (desc->d_nmask_offsets == nullptr))
        return rb::fail(RB_ERR_INVALID_ARG, "packed input needs both the N bitmap and its offsets");
    std::lock_guard<std::mutex> lock(e->mu);
    int rc = check_device(e->device);
    if (rc != RB_OK) return rc;
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    const size_t nf = e->filters.size();
    // on-GPU chunking / read indirection: effective per-item lengths (+ the bad-chunk status) are made on the device
    const bool chunked = desc->chunk_start != 0 || desc->chunk_length != 0 || desc->d_read_ids != nullptr;
    const void *d_lens = desc->d_lens;
    const uint8_t *d_pre_status = nullptr;
    uint32_t max_len = desc->max_len;
    if (chunked) {
        rc = e->d_efflens.ensure(n_reads * 4);
        if (rc == RB_OK) rc = e->d_prestatus.ensure(n_reads);
        if (rc != RB_OK) return rc;
        RB_HIP(launch_chunk_prep((const uint32_t *)desc->d_lens, (const uint32_t *)desc->d_read_ids, (uint32_t)n_reads,
                                 desc->chunk_start, desc->chunk_length, (uint32_t *)e->d_efflens.p,
                                 (uint8_t *)e->d_prestatus.p, st));
        d_lens = e->d_efflens.p;
        d_pre_status = (const uint8_t *)e->d_prestatus.p;
        if (desc->chunk_length && desc->chunk_length < max_len) max_len = desc->chunk_length;
    }

    uint16_t *maxcount = (uint16_t *)d_maxcount;
    if (!maxcount) {
        rc = e->d_maxcount.ensure(n_reads * nf * 2);
        if (rc != RB_OK) return rc;
        maxcount = (uint16_t *)e->d_maxcount.p;
    }
    std::pair<hipEvent_t, hipEvent_t> *evp = nullptr;
    if (e->timing && e->ev_used < ((size_t)1 << 16)) {  // bounded: a caller that never collects stops being timed
        if (e->ev_used == e->ev_ring.size()) {
            hipEvent_t a = nullptr, b = nullptr;
            RB_HIP(hipEventCreate(&a));
            RB_HIP(hipEventCreate(&b));
            e->ev_ring.emplace_back(a, b);
        }
        evp = &e->ev_ring[e->ev_used++];
        RB_HIP(hipEventRecord(evp->first, st));
    }
    // fork/join over auxiliary streams costs ~20-40 us of event traffic per call (measured): worth it for large
    // batches only; micro-batches queue their few short kernels on the one stream
    // opt-in early decision: the count kernels of the plain throughput form read the decision kernel's threshold table (made here, on the
    // call's stream, before any count kernel -- the auxiliary streams wait for the fork event below)
    const uint16_t *early_thr = nullptr;
    uint32_t early_thr_len = 0;
    if (e->early_decision && mode == RB_MODE_CHECK_UNBLOCK && !d_maxcount && !host_maxcount && e->shard_world == 1 && n_reads > e->split_threshold &&
        (d_decision || d_status || d_best_target)) {
        rc = ensure_thresholds(e, max_len, error_rate, significance, st, &early_thr, &early_thr_len);
        if (rc != RB_OK) return rc;
    }
    const bool fan_out = e->overlap && nf > 1 && !e->aux.empty() && n_reads > e->split_threshold;
    if (fan_out) {
        RB_HIP(hipEventRecord(e->fork_ev, st));
        for (size_t k = 0; k < e->aux.size(); ++k) RB_HIP(hipStreamWaitEvent(e->aux[k], e->fork_ev, 0));
    }
    // filters of one hash geometry that share a merged table: one launch per group (throughput form only; micro-batches
    // keep the latency kernels, which already put every filter of a call into one launch)
    if (!e->merged_planned) plan_merged(e);
    const bool use_merged = !e->merged.empty() && n_reads > e->split_threshold && e->shard_world == 1;
    if (use_merged) {
        for (size_t gi = 0; gi < e->merged.size(); ++gi) {
            MergedGroup *g = e->merged[gi];
            if (g->members.empty()) continue;  // dissolved
            // shared: from "the copy is fresh" to "the kernel that reads it is queued"; a stale (or missing) copy is made under the
            // exclusive lock, after which the check is repeated (bounded: a filter that is inserted into without pause does not
            // stall its readers, they go on with the copy of the moment)
            std::shared_lock<std::shared_mutex> reading(merged_rw(e->device));
            rc = RB_OK;
            for (int attempt = 0; attempt < 4 && !merged_table_fresh(e, g); ++attempt) {
                reading.unlock();
                {
                    std::unique_lock<std::shared_mutex> writing(merged_rw(e->device));
                    rc = ensure_merged_table(e, g, st);
                }
                reading.lock();
                if (rc != RB_OK) break;
            }
            if (rc == kMergeNoMemory || (rc == RB_OK && !g->tab)) {
                for (uint32_t m : g->members) e->merged_of[m] = -1;
                g->members.clear();
                continue;
            }
            if (rc != RB_OK) return rc;
            CountLaunch a{};
            a.src.seqs = (const uint8_t *)d_seqs;
            a.src.offsets = (const uint64_t *)d_offsets;
            a.src.lens = (const uint32_t *)d_lens;
            a.src.nmask = (const uint8_t *)desc->d_nmask;
            a.src.nmask_offsets = (const uint64_t *)desc->d_nmask_offsets;
            a.src.ids = (const uint32_t *)desc->d_read_ids;
            a.src.base_off = desc->chunk_start;
            a.src.max_len = max_len;
            a.n_reads = (uint32_t)n_reads;
            if (g->width <= 4) {
                // a merged block of two to four words is held by ONE lane of the both-strands builds of the phased kernel: the
                // merged table is planned like a filter of that width (slices, windows), the kernel writes one maximum per member
                rb_dibf as_filter;
                as_filter.device = e->device;
                as_filter.geo = e->filters[g->members[0]]->geo;
                as_filter.geo.bin_width = g->width;
                as_filter.geo.n_bins = g->width * 64;
                as_filter.stride = g->stride;
                CountLaunch p = a;
                const bool planned = plan_geometry(e, &as_filter, n_reads, max_len, p);
                const bool one_lane = (p.lg == 1 && p.short_only >= 1 && p.short_only <= 3) || (p.lg == 2 && (p.short_only == 4 || p.short_only == 5));
                if (planned && p.phase.n_slices && p.split_waves < 2 && p.planes <= 10 && one_lane && g->map.n <= kMaxNarrow) {
                    p.f = g->dev;
                    p.f.comp_n = e->revcomp_of_n;
                    if (p.multi_reads && g->tab && g->tab->d_inv && !e->multi_no_inv) {  // the complemented twin of the copy: the OR form of the multi-read build
                        p.f.words = g->tab->d_inv;
                        p.multi_inv = 1;
                    }
                    p.narrow = NarrowMerge{};
                    p.narrow.n = g->map.n;
                    for (uint32_t c = 0; c < 4; ++c) {  // bins of a column are its low bits in both layouts: up to the end of the last member in it
                        uint32_t top = 0;
                        for (uint32_t m = 0; m < g->map.n; ++m)
                            if (g->map.bit_begin[m] < (c + 1) * 64u && g->map.bit_end[m] > c * 64u) top = std::max(top, std::min(g->map.bit_end[m], (c + 1) * 64u) - c * 64u);
                        p.narrow.col_bits[c] = c < g->width ? top : 0u;
                    }
                    for (uint32_t m = 0; m < g->map.n; ++m) {
                        p.narrow.bit_begin[m] = g->map.bit_begin[m];
                        p.narrow.bit_end[m] = g->map.bit_end[m];
                        p.narrow.out_offset[m] = g->map.out_offset[m];
                    }
                    p.out = maxcount;
                    p.out_read_stride = (uint32_t)nf;
                    p.out_slice_stride = 0;
                    RB_HIP(launch_ibf_count_max(p, st));
                    continue;
                }
            }
            a.f = g->dev;
            a.f.comp_n = e->revcomp_of_n;
            a.src.seqs = (const uint8_t *)d_seqs;
            a.src.offsets = (const uint64_t *)d_offsets;
            a.src.lens = (const uint32_t *)d_lens;
            a.src.nmask = (const uint8_t *)desc->d_nmask;
            a.src.nmask_offsets = (const uint64_t *)desc->d_nmask_offsets;
            a.src.ids = (const uint32_t *)desc->d_read_ids;
            a.src.base_off = desc->chunk_start;
            a.src.max_len = max_len;
            a.n_reads = (uint32_t)n_reads;
            a.wpl = 1;
            a.lg = 0;
            while ((1u << a.lg) < g->width) ++a.lg;
            const uint32_t kmers = max_len >= g->dev.k ? max_len - g->dev.k + 1 : 0;
            a.planes = kmers <= 1023 ? 10 : 16;
            a.nt = g->n_blocks * g->stride * 8 > e->nt_threshold_bytes;
            a.out = maxcount;
            a.out_read_stride = (uint32_t)nf;
            RB_HIP(launch_ibf_count_max_merged(a, g->map, st));
        }
    }
    std::vector<CountLaunch> pending;
    std::vector<uint32_t> pending_fi;
    // may the count kernel of this call decide as well?  (one filter, in the latency form with one column slice: settled below)
    // (a call that announces its completion through the host word folds only when it has ONE read: the deciding thread then holds the
    // call's last result; with more reads the decision kernel, one workgroup up to 256 reads, announces)
    bool fold_ok = e->fold_decide && nf == 1 && e->shard_world == 1 && (d_best_target || d_decision || d_status) && !fan_out && !use_merged &&
                   n_reads <= e->split_threshold && n_reads <= e->fold_max_reads && (!done_flag || n_reads == 1);
    bool folded = false;
    FoldJob job{};
    // Which filters may run side by side?  A table of a few tens of MB lives partly in the 4 MiB L2 of each XCD (hit rate
    // about 4 MiB / table); two such tables gathered at once halve each other's share, so narrow filters take turns on the
    // call's stream (measured on the README shape, four filters of 10-20 MB: 13.3 -> 16.5 M reads/s).  Tables far beyond
    // that have no L2 share to lose and overlap on the auxiliary streams as before.
    auto l2_sensitive = [&](const rb_dibf *f) { return f->geo.n_blocks * f->stride * 8 <= e->serial_table_bytes; };
    size_t n_big = 0, n_small = 0;
    for (const rb_dibf *f : e->filters) (l2_sensitive(f) ? n_small : n_big) += 1;
    size_t next_aux = 0;
    bool big_on_main = false;
    for (size_t fi = 0; fi < nf; ++fi) {
        if (use_merged && e->merged_of[fi] >= 0) continue;  // counted by its group's launch above
        const rb_dibf *f = e->filters[fi];
        hipStream_t fs = st;
        if (fan_out && !l2_sensitive(f)) {
            if (n_small == 0 && !big_on_main) big_on_main = true;  // no narrow filter wants the main stream: the first big one takes it
            else fs = e->aux[next_aux++ % e->aux.size()];
        }
        CountLaunch a{};
        a.f = f->dev;
        a.f.comp_n = e->revcomp_of_n;
        a.src.seqs = (const uint8_t *)d_seqs;
        a.src.offsets = (const uint64_t *)d_offsets;
        a.src.lens = (const uint32_t *)d_lens;
        a.src.nmask = (const uint8_t *)desc->d_nmask;
        a.src.nmask_offsets = (const uint64_t *)desc->d_nmask_offsets;
        a.src.ids = (const uint32_t *)desc->d_read_ids;
        a.src.base_off = desc->chunk_start;
            a.src.max_len = max_len;
        a.n_reads = (uint32_t)n_reads;
        if (!plan_geometry(e, f, n_reads, max_len, a)) {
            // this rank holds no column of this filter: its partial maxima are 0
            fold_ok = false;
            RB_HIP(hipMemset2DAsync(maxcount + fi, nf * 2, 0, 2, n_reads, fs));
            continue;
        }
        if (a.n_slices != 1 || a.split_waves < 2) fold_ok = false;
        // (a target filter's count also picks best_target -- the strictly-greater argmax of IBFClassify.cpp:262-273 -- so target filters stop
        // early only when the caller does not ask for best_target; a deplete filter's count is seen through the two predicates alone)
        if (early_thr && a.split_waves < 2 && (fi < e->nd || !d_best_target)) {
            a.early_thr = early_thr;
            a.early_thr_len = early_thr_len;
            a.early_nf = (uint32_t)nf;
            a.early_fi = (uint32_t)fi;
        }
        if (a.n_slices == 1) {
            a.out = maxcount + fi;
            a.out_read_stride = (uint32_t)nf;
            a.out_slice_stride = 0;
            if (!fan_out && nf > 1 && n_reads <= e->split_threshold) {
                pending.push_back(a);  // micro-batch: fused below with the filters of equal kernel geometry
                pending_fi.push_back((uint32_t)fi);
            } else {
                if (fold_ok) {  // the engine's only filter in the latency form: this launch decides
                    if ((rc = decide_params(e, max_len, error_rate, significance, fs, host_maxcount, 1, 0, &job.P, done_flag, done_seq, n_reads)) != RB_OK) return rc;
                    job.on = 1;
                    job.mode = mode;
                    job.maxcount = maxcount;
                    job.lens = (const uint32_t *)d_lens;
                    job.pre_status = d_pre_status;
                    job.best_target = (int32_t *)d_best_target;
                    job.decision = (uint8_t *)d_decision;
                    job.status = (uint8_t *)d_status;
                    a.fold = &job;
                    folded = true;
                }
                if ((rc = ensure_split_ws(e, a, 1, fs)) != RB_OK) return rc;
                RB_HIP(launch_ibf_count_max(a, fs));
            }
        } else {
            rc = e->d_parts[fi].ensure((size_t)a.n_slices * n_reads * 2);
            if (rc != RB_OK) return rc;
            a.out = (uint16_t *)e->d_parts[fi].p;
            a.out_read_stride = 1;
            a.out_slice_stride = (uint32_t)n_reads;
            if ((rc = ensure_split_ws(e, a, 1, fs)) != RB_OK) return rc;
            RB_HIP(launch_ibf_count_max(a, fs));
            RB_HIP(launch_reduce_slices(a.out, a.n_slices, (uint32_t)n_reads, maxcount, (uint32_t)nf, (uint32_t)fi, fs));
        }
    }
    if ((rc = launch_fused_groups(e, pending, pending_fi, maxcount, st)) != RB_OK) return rc;
    if (fan_out) {
        for (size_t k = 0; k < std::min(e->aux.size(), next_aux); ++k) {
            RB_HIP(hipEventRecord(e->join_ev[k], e->aux[k]));
            RB_HIP(hipStreamWaitEvent(st, e->join_ev[k], 0));
        }
    }
    if (evp) RB_HIP(hipEventRecord(evp->second, st));
    if (!folded && e->shard_world == 1 && (d_best_target || d_decision || d_status)) {
        rc = run_decide(e, maxcount, (const uint32_t *)d_lens, d_pre_status, n_reads, max_len, error_rate, significance,
                        mode, (int32_t *)d_best_target, (uint8_t *)d_decision, (uint8_t *)d_status, st, host_maxcount, 1, 0, done_flag, done_seq);
        if (rc != RB_OK) return rc;
    } else if (!folded && done_flag) {
        return rb::fail(RB_ERR_INVALID_ARG, "completion word without a decision");
    }
    if (!stream) RB_HIP(hipStreamSynchronize(st));
    e->tickets_dirty = false;  // every launch of this call was accepted (and, on the engine's own stream, has finished)
    e->done_dirty = false;
    return RB_OK;
}

// what the engine would launch for filter `filter_index` on a batch of n_reads reads of at most max_len bases
extern "C" int rb_engine_plan(rb_engine *e, size_t filter_index, size_t n_reads, uint32_t max_len, rb_plan_info *out)
{
    if (!e || !out || filter_index >= e->filters.size()) return rb::fail(RB_ERR_INVALID_ARG, "rb_engine_plan: bad argument");
    std::lock_guard<std::mutex> lock(e->mu);
    int rc = check_device(e->device);
    if (rc != RB_OK) return rc;
    std::memset(out, 0, sizeof *out);
    if (!e->merged_planned) plan_merged(e);
    const rb_dibf *f = e->filters[filter_index];
    rb_dibf as_filter;
    const MergedGroup *g = nullptr;
    if (!e->merged.empty() && n_reads > e->split_threshold && e->shard_world == 1 && e->merged_of[filter_index] >= 0 &&
        !e->merged[e->merged_of[filter_index]]->members.empty())
        g = e->merged[e->merged_of[filter_index]];
    if (g) {  // the table the lookups really go to: the group's merged copy, planned like a filter of its width
        out->merged_members = (uint32_t)g->members.size();
        as_filter.device = e->device;
        as_filter.geo = f->geo;
        as_filter.geo.bin_width = g->width;
        as_filter.geo.n_bins = g->width * 64;
        as_filter.stride = hbm_stride(g->width);
        f = &as_filter;
    }
    CountLaunch a{};
    a.n_reads = (uint32_t)std::min<size_t>(n_reads, 0x7FFFFFFFu);
    const bool planned = plan_geometry(e, f, n_reads, max_len, a);
    out->table_bytes = f->geo.n_blocks * f->stride * 8;
    out->block_words = (uint32_t)f->geo.bin_width;
    out->stride_words = (uint32_t)f->stride;
    if (!planned) return RB_OK;  // this rank owns no column of the filter
    const bool one_lane = (a.lg == 1 && a.short_only >= 1 && a.short_only <= 3) || (a.lg == 2 && (a.short_only == 4 || a.short_only == 5));
    const bool merged_plain = g && !(g->width <= 4 && a.phase.n_slices && a.split_waves < 2 && a.planes <= 10 && one_lane);
    out->lanes_per_block_log2 = (uint32_t)a.lg;
    out->words_per_lane = (uint32_t)a.wpl;
    out->column_slices = a.n_slices;
    out->counter_planes = (uint32_t)a.planes;
    out->nontemporal = (uint32_t)a.nt;
    out->split_waves = (uint32_t)(a.split_waves >= 2 ? a.split_waves : 0);
    const char *form = "ibf_count_max_kernel";
    if (a.split_waves >= 2) form = "ibf_count_max_split_kernel";
    else if (merged_plain) form = "ibf_count_max_merged_kernel";
    else if (a.phase.n_slices) form = a.multi_reads ? "ibf_count_max_phased_multi_kernel" : "ibf_count_max_phased_kernel";
    std::snprintf(out->kernel, sizeof out->kernel, "%s", form);
    out->reserved0 = (uint32_t)a.multi_reads;  // reads per wave of the multi build (0: the one-read build)
    if (!merged_plain && a.split_waves < 2 && a.phase.n_slices) {
        out->phase_shape = (uint32_t)a.phase_shape;
        std::snprintf(out->phase_shape_name, sizeof out->phase_shape_name, "%s", phase_rule((PhaseShape)a.phase_shape, a.lg).name);
        out->phase_slices = a.phase.n_slices;
        if (a.phase.inv_ticks) {  // clock-phased (one slice and no clock: the both-strands round of that kernel without waiting)
            out->phased = 1;
            out->phase_slice_log2 = a.phase_slice_log2;
            out->phase_slice_bytes = a.phase_slice_bytes;
            out->phase_window_ticks = a.phase_ticks;
            out->phase_rule_ticks = a.phase_rule_ticks;
        }
    }
    return RB_OK;
}

// Fits the windows of the clock-phased gathers to THIS device: the planner's table was measured on one box;
// clocks, firmware and compilers move the optima.  For every table the engine would serve with the phased form on a batch of
// n_reads reads of read_len bases, K1 of the whole engine is timed on synthetic reads with the rule's window and with the window
// x 0.7 / 0.85 / 1.2 / 1.45 (same slice size); the fastest -- if it beats the rule by more than 2 % -- replaces the rule for that
// table, kernel shape and slice size.  Results never depend on it.  Call it on an idle engine.
extern "C" int rb_engine_calibrate(rb_engine *e, size_t n_reads, uint32_t read_len, double max_ms, uint32_t *n_tables, uint32_t *n_changed)
{
    if (!e || n_reads == 0 || n_reads >= (1ULL << 28) || read_len == 0 || read_len > 100000) return rb::fail(RB_ERR_INVALID_ARG, "rb_engine_calibrate: bad argument");
    if (n_tables) *n_tables = 0;
    if (n_changed) *n_changed = 0;
    int rc = check_device(e->device);
    if (rc != RB_OK) return rc;
    const auto t_begin = std::chrono::steady_clock::now();
    auto elapsed_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    // which tables are phased, and with what: the engine's own answer per filter (merged groups answer for all their members)
    struct Key { uint64_t table_bytes; uint32_t stride, slice_log2, rule_ticks; int shape, lg; };
    std::vector<Key> keys;
    for (size_t fi = 0; fi < e->filters.size(); ++fi) {
        rb_plan_info pl;
        if ((rc = rb_engine_plan(e, fi, n_reads, read_len, &pl)) != RB_OK) return rc;
        if (!pl.phased) continue;
        Key k{pl.table_bytes, pl.stride_words, pl.phase_slice_log2, pl.phase_rule_ticks, (int)pl.phase_shape, (int)pl.lanes_per_block_log2};
        bool seen = false;
        for (const Key &o : keys) seen |= o.table_bytes == k.table_bytes && o.stride == k.stride && o.shape == k.shape && o.lg == k.lg && o.slice_log2 == k.slice_log2;
        if (!seen) keys.push_back(k);
    }
    if (n_tables) *n_tables = (uint32_t)keys.size();
    if (keys.empty()) return RB_OK;
    // synthetic batch on the device: uniform ACGT (the lookups of a read are uniform over the table whatever its bases are)
    DevBuf d_reads, d_off, d_len, d_max;
    const size_t nf = e->filters.size();
    if ((rc = d_reads.ensure(n_reads * (size_t)read_len)) != RB_OK || (rc = d_off.ensure(n_reads * 8)) != RB_OK || (rc = d_len.ensure(n_reads * 4)) != RB_OK ||
        (rc = d_max.ensure(n_reads * nf * 2)) != RB_OK) {
        d_reads.release(); d_off.release(); d_len.release(); d_max.release();
        return rc;
    }
    hipError_t he = launch_fill_reads((uint8_t *)d_reads.p, (uint64_t *)d_off.p, (uint32_t *)d_len.p, n_reads, read_len, 0x5eedULL, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    auto cleanup = [&] { d_reads.release(); d_off.release(); d_len.release(); d_max.release(); };
    if (he != hipSuccess) { cleanup(); return rb::fail(RB_ERR_HIP, std::string("calibrate: ") + hipGetErrorString(he)); }
    rb_batch_desc desc;
    std::memset(&desc, 0, sizeof desc);
    desc.d_seqs = d_reads.p;
    desc.d_offsets = d_off.p;
    desc.d_lens = d_len.p;
    desc.n_items = n_reads;
    desc.max_len = read_len;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (hipEventCreate(&ev0) != hipSuccess || hipEventCreate(&ev1) != hipSuccess) { cleanup(); return rb::fail(RB_ERR_HIP, "calibrate: events"); }
    auto k1_ms = [&](double *out) -> int {  // one untimed launch, then the MEDIAN of five, each timed on its own: single launches of the
                                            // two-word shapes can land 10-15 % off their usual time (a mean of two picked such outliers)
        double ms[5];
        for (int it = 0; it < 6; ++it) {
            (void)hipEventRecord(ev0, e->stream);
            const int r = classify_device_impl(e, &desc, 0.1, 0.95, RB_MODE_CHECK_UNBLOCK, d_max.p, nullptr, nullptr, nullptr, (void *)e->stream, nullptr);
            if (r != RB_OK) return r;
            (void)hipEventRecord(ev1, e->stream);
            if (hipEventSynchronize(ev1) != hipSuccess) return rb::fail(RB_ERR_HIP, "calibrate: sync");
            float t = 0.f;
            (void)hipEventElapsedTime(&t, ev0, ev1);
            if (it > 0) ms[it - 1] = t;
        }
        std::sort(ms, ms + 5);
        *out = ms[2];
        return RB_OK;
    };
    // Windows around the rule's, in order.  Two-word and wide blocks show narrow dips and cliffs along the window length (a neighbour
    // of the best point can be 30 % slower), and where they lie moves with the batch size: the winner is the best point of the
    // SMOOTHED curve (half the point, a quarter of each neighbour), it has to beat the rule's smoothed time by 4 %, and a second
    // measurement of rule and winner has to confirm 3 % -- otherwise the rule stays.
    static const double kFactors[] = {0.7, 0.85, 1.0, 1.2, 1.45};
    constexpr int kN = 5, kRule = 2;
    auto set_trial = [&](const Key &k, uint32_t ticks, bool install) {
        std::lock_guard<std::mutex> lock(e->mu);
        auto &ov = e->phase_overrides;
        ov.erase(std::remove_if(ov.begin(), ov.end(), [&](const rb_engine::PhaseOverride &o) {
                     return o.table_bytes == k.table_bytes && o.stride == k.stride && o.shape == k.shape && o.lg == k.lg && o.slice_log2 == k.slice_log2; }), ov.end());
        if (install) ov.push_back(rb_engine::PhaseOverride{k.table_bytes, k.stride, k.slice_log2, ticks, k.shape, k.lg});
    };
    uint32_t changed = 0;
    for (const Key &k : keys) {
        double t[kN] = {0, 0, 0, 0, 0};
        uint32_t ticks[kN];
        bool complete = true;
        for (int i = 0; i < kN && rc == RB_OK; ++i) {
            ticks[i] = (uint32_t)std::min(2000.0, std::max(100.0, k.rule_ticks * kFactors[i]));
            if (max_ms > 0 && elapsed_ms() > max_ms) { complete = false; break; }
            set_trial(k, ticks[i], true);
            rc = k1_ms(&t[i]);
        }
        int best = kRule;
        if (rc == RB_OK && complete) {
            double sm[kN];
            for (int i = 0; i < kN; ++i) sm[i] = 0.5 * t[i] + 0.25 * t[i > 0 ? i - 1 : i] + 0.25 * t[i + 1 < kN ? i + 1 : i];
            for (int i = 0; i < kN; ++i)
                if (sm[i] < sm[best]) best = i;
            if (best != kRule && !(sm[best] < 0.96 * sm[kRule])) best = kRule;
            if (best != kRule && ticks[best] != ticks[kRule]) {  // confirm on fresh measurements
                double t_rule = 0.0, t_best = 0.0;
                set_trial(k, ticks[kRule], true);
                rc = k1_ms(&t_rule);
                if (rc == RB_OK) {
                    set_trial(k, ticks[best], true);
                    rc = k1_ms(&t_best);
                }
                if (rc != RB_OK || !(t_best < 0.97 * t_rule)) best = kRule;
            } else {
                best = kRule;
            }
        }
        set_trial(k, ticks[best < kN ? best : kRule], rc == RB_OK && best != kRule);
        if (rc == RB_OK && best != kRule) ++changed;
        if (rc != RB_OK) break;
    }
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    cleanup();
    if (n_changed) *n_changed = changed;
    return rc;
}

extern "C" {

int rb_classify_batch_device_ex(rb_engine *e, const rb_batch_desc *desc, double error_rate, double significance, int mode,
                                void *d_maxcount, void *d_best_target, void *d_decision, void *d_status, void *stream)
{
    return classify_device_impl(e, desc, error_rate, significance, mode, d_maxcount, d_best_target, d_decision, d_status,
                                stream, nullptr);
}

int rb_decide_device_parts(rb_engine *e, const void *d_maxcount, uint32_t n_parts, uint64_t part_stride, const void *d_lens,
                           size_t n_reads, uint32_t max_len, double error_rate, double significance, int mode,
                           void *d_best_target, void *d_decision, void *d_status, void *stream)
{
    if (!e || !d_maxcount || !d_lens) return rb::fail(RB_ERR_INVALID_ARG, "null argument");
    if (n_parts == 0 || (n_parts > 1 && part_stride < n_reads * e->filters.size()))
        return rb::fail(RB_ERR_INVALID_ARG, "partial tables overlap");
    if (mode != RB_MODE_CHECK_UNBLOCK && mode != RB_MODE_CLASSIFY_CHUNK && mode != RB_MODE_CLASSIFY_ANY)
        return rb::fail(RB_ERR_INVALID_ARG, "unknown mode");
    if (n_reads == 0) return RB_OK;
    std::lock_guard<std::mutex> lock(e->mu);
    int rc = check_device(e->device);
    if (rc != RB_OK) return rc;
    hipStream_t st = stream ? (hipStream_t)stream : e->stream;
    rc = run_decide(e, (const uint16_t *)d_maxcount, (const uint32_t *)d_lens, nullptr, n_reads, max_len, error_rate,
                    significance, mode, (int32_t *)d_best_target, (uint8_t *)d_decision, (uint8_t *)d_status, st, nullptr,
                    n_parts, part_stride);
    if (rc != RB_OK) return rc;
    if (!stream) RB_HIP(hipStreamSynchronize(st));
    return RB_OK;
}

int rb_decide_device(rb_engine *e, const void *d_maxcount, const void *d_lens, size_t n_reads, uint32_t max_len,
                     double error_rate, double significance, int mode, void *d_best_target, void *d_decision,
                     void *d_status, void *stream)
{
    return rb_decide_device_parts(e, d_maxcount, 1, 0, d_lens, n_reads, max_len, error_rate, significance, mode, d_best_target,
                                  d_decision, d_status, stream);
}

int rb_classify_batch(rb_engine *e, const char *seqs, const uint64_t *offsets, const uint32_t *lens, size_t n_reads,
                      double error_rate, double significance, int mode, uint16_t *out_maxcount, int32_t *out_best_target,
                      uint8_t *out_decision, uint8_t *out_status)
{
    if (!e) return rb::fail(RB_ERR_INVALID_ARG, "null engine");
    if (n_reads == 0) return RB_OK;
    if (!seqs || !offsets || !lens) return rb::fail(RB_ERR_INVALID_ARG, "null input buffer");
    int rc = check_device(e->device);
    if (rc != RB_OK) return rc;
    uint64_t hi = 0, lo = ~0ULL, sum_len = 0;
    uint32_t max_len = 0;
    for (size_t i = 0; i < n_reads; ++i) {
        hi = std::max<uint64_t>(hi, offsets[i] + lens[i]);
        lo = std::min<uint64_t>(lo, offsets[i]);
        sum_len += lens[i];
        max_len = std::max(max_len, lens[i]);
    }
    const size_t nf = e->filters.size();
    const size_t n = n_reads;
    hipStream_t st = e->stream;
    std::lock_guard<std::mutex> host_lock(e->host_mu);  // the staging buffers below are per engine
    const bool sharded = e->shard_world != 1;

    if (sum_len + 16 * n <= ((uint64_t)8 << 20)) {
        // ---- micro-batch path: one pinned staging block each way -> one H2D copy in; results are written into the pinned
        // output block by the decision kernel itself.
        // in : u64 offsets[n] | u32 lens[n] | compacted read bytes      out: i32 best[n] | u16 maxcount[n*nf] | u8 decision[n] | u8 status[n]
        const size_t in_bytes = 12 * n + (size_t)sum_len + 1;
        const size_t out_bytes = 4 * n + 2 * nf * n + 2 * n;
        {
            std::lock_guard<std::mutex> lock(e->mu);
            if ((rc = e->h_in.ensure(in_bytes + 16)) != RB_OK) return rc;  // (+16: the copy kernel moves whole 16-byte units)
            if ((rc = e->h_out.ensure(out_bytes)) != RB_OK) return rc;
            if ((rc = e->d_seqs.ensure(in_bytes + 16)) != RB_OK) return rc;
            if ((rc = e->d_maxcount.ensure(out_bytes)) != RB_OK) return rc;
        }
        uint64_t *ho = (uint64_t *)e->h_in.p;
        uint32_t *hl = (uint32_t *)(ho + n);
        char *hs = (char *)(hl + n);
        uint64_t pos = 0;
        for (size_t i = 0; i < n; ++i) {
            ho[i] = pos;
            hl[i] = lens[i];
            std::memcpy(hs + pos, seqs + offsets[i], lens[i]);
            pos += lens[i];
        }
        char *din = (char *)e->d_seqs.p;
        char *dout = (char *)e->d_maxcount.p;
        char *hout = (char *)e->h_out.p;
        // up to 1 MiB the batch is fetched by a kernel of this stream: the runtime's copy is a blit kernel of its own that costs about 5 us
        // more from 64 reads on (64 reads 69.8 -> 64.8 us host to host, 256: 138.4 -> 133.6, 512: 240.6 -> 235.7; equal at one read and
        // from 1 024 reads on: profiles/r05/micro_copy_ab.txt)
        if (in_bytes <= e->micro_copy_kernel_bytes) RB_HIP(launch_copy_from_host(e->h_in.p, din, in_bytes, st));
        else RB_HIP(hipMemcpyAsync(din, e->h_in.p, in_bytes, hipMemcpyHostToDevice, st));
        rb_batch_desc desc;
        std::memset(&desc, 0, sizeof desc);
        desc.d_seqs = din + 12 * n;
        desc.d_offsets = din;
        desc.d_lens = din + 8 * n;
        desc.n_items = n;
        desc.max_len = max_len;
        uint16_t *d_max = (uint16_t *)(dout + 4 * n);
        if (e->word_calls >= e->completion_sync_every) {
            RB_HIP(hipStreamSynchronize(st));
            e->word_calls = 0;
        }
        // the call's last kernel announces the results itself (completion word), the host spins on the word: 4 us less than the stream's wait
        uint32_t *done = nullptr;
        uint32_t seq = 0;
        if (e->completion_word && !sharded && n <= e->completion_max_reads) {
            std::lock_guard<std::mutex> lock(e->mu);
            if ((rc = e->h_done.ensure(64)) != RB_OK) return rc;
            done = (uint32_t *)e->h_done.p;
            if (++e->done_seq == 0) e->done_seq = 1;
            seq = e->done_seq;
            __atomic_store_n(done, 0u, __ATOMIC_RELAXED);
        }
        if (!sharded) {
            // the decision kernel writes its results (and a copy of the maxcount rows) straight into the pinned block:
            // posted PCIe writes, no device-to-host copy command behind the kernels (one dependent launch less per call)
            rc = classify_device_impl(e, &desc, error_rate, significance, mode, d_max, hout, hout + 4 * n + 2 * nf * n,
                                      hout + 4 * n + 2 * nf * n + n, (void *)st, (uint16_t *)(hout + 4 * n), done, seq);
            if (rc != RB_OK) return rc;
        } else {
            rc = classify_device_impl(e, &desc, error_rate, significance, mode, d_max, nullptr, nullptr, nullptr, (void *)st,
                                      nullptr);
            if (rc != RB_OK) return rc;
            RB_HIP(hipMemcpyAsync(hout + 4 * n, d_max, 2 * nf * n, hipMemcpyDeviceToHost, st));
        }
        bool arrived = false;
        if (done) {
            const auto t0 = std::chrono::steady_clock::now();
            for (uint32_t spins = 0;; ++spins) {
                if (__atomic_load_n(done, __ATOMIC_ACQUIRE) == seq) { arrived = true; break; }
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#endif
                if ((spins & 1023u) == 1023u &&
                    std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > (long long)e->completion_spin_us)
                    break;  // overdue (or a kernel has failed): the stream says which
            }
        }
        if (!arrived) {
            RB_HIP(hipStreamSynchronize(st));
            e->word_calls = 0;
        } else {
            ++e->word_calls;
        }
        if (out_maxcount) std::memcpy(out_maxcount, hout + 4 * n, 2 * nf * n);
        if (!sharded) {
            if (out_best_target) std::memcpy(out_best_target, hout, 4 * n);
            if (out_decision) std::memcpy(out_decision, hout + 4 * n + 2 * nf * n, n);
            if (out_status) std::memcpy(out_status, hout + 4 * n + 2 * nf * n + n, n);
        }
        return RB_OK;
    }

    // ---- large batches: the spanned byte range goes over as it is (offsets stay valid relative to a shifted base), in
    // slices of consecutive reads: slice i+1 crosses PCIe on the copy stream while slice i is counted.  A pageable
    // source makes hipMemcpyAsync block the host while it stages, which is exactly when the GPU works on the slice before.
    // Offsets and lengths go through the engine's page-locked input block (one asynchronous copy each, ahead of the first
    // slice), and -- up to 4 M reads per call -- the decision kernel writes its results and a copy of the maxcount rows
    // straight into the page-locked output block, like the micro-batch path: no device-to-host copy command behind the
    // kernels, no pageable copy that would block the calling thread four times per call while other host threads wait for
    // the GPU (the CLI's classifier threads: one call per 32-64 k reads each).
    const uint64_t span = hi - lo;
    const bool direct_out = !sharded && n <= ((size_t)1 << 22);
    const size_t out_bytes = 4 * n + 2 * nf * n + 2 * n;
    {
        std::lock_guard<std::mutex> lock(e->mu);
        if ((rc = e->d_seqs.ensure(span ? span : 1)) != RB_OK) return rc;
        if ((rc = e->d_offsets.ensure(n * 8)) != RB_OK) return rc;
        if ((rc = e->d_lens.ensure(n * 4)) != RB_OK) return rc;
        if ((rc = e->d_maxcount.ensure(n * nf * 2)) != RB_OK) return rc;
        if ((rc = e->h_in.ensure(n * 12)) != RB_OK) return rc;
        if (direct_out) {
            if ((rc = e->h_out.ensure(out_bytes)) != RB_OK) return rc;
        } else {
            if ((rc = e->d_best.ensure(n * 4)) != RB_OK) return rc;
            if ((rc = e->d_decision.ensure(n)) != RB_OK) return rc;
            if ((rc = e->d_status.ensure(n)) != RB_OK) return rc;
        }
    }
    char *d_seqs = (char *)e->d_seqs.p;
    uint64_t *d_off = (uint64_t *)e->d_offsets.p;
    uint32_t *d_len = (uint32_t *)e->d_lens.p;
    uint16_t *d_max = (uint16_t *)e->d_maxcount.p;
    char *hout = (char *)e->h_out.p;
    const char *d_base = d_seqs - lo;  // device address of the caller's seqs[0]
    hipStream_t cs = e->copy_stream;
    std::memcpy(e->h_in.p, offsets, n * 8);
    std::memcpy((char *)e->h_in.p + n * 8, lens, n * 4);
    RB_HIP(hipMemcpyAsync(d_off, e->h_in.p, n * 8, hipMemcpyHostToDevice, cs));
    RB_HIP(hipMemcpyAsync(d_len, (char *)e->h_in.p + n * 8, n * 4, hipMemcpyHostToDevice, cs));
    size_t i0 = 0, slice = 0;
    while (i0 < n) {
        uint64_t bytes = 0, s_lo = ~0ULL, s_hi = 0;
        size_t i1 = i0;
        while (i1 < n && (i1 == i0 || bytes + lens[i1] <= e->host_slice_bytes)) {
            bytes += lens[i1];
            s_lo = std::min<uint64_t>(s_lo, offsets[i1]);
            s_hi = std::max<uint64_t>(s_hi, offsets[i1] + lens[i1]);
            ++i1;
        }
        const size_t cnt = i1 - i0;
        // a wait binds to the record that precedes it, so a small ring of events can be re-recorded by later slices
        const size_t evi = slice % 64;
        if (evi == e->copy_ev.size()) {
            hipEvent_t ev = nullptr;
            RB_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            e->copy_ev.push_back(ev);
        }
        // reads that alias or interleave across slices are copied again to the same place: same bytes, no hazard for the
        // kernels already reading them
        if (s_hi > s_lo) RB_HIP(hipMemcpyAsync(d_seqs + (s_lo - lo), seqs + s_lo, s_hi - s_lo, hipMemcpyHostToDevice, cs));
        RB_HIP(hipEventRecord(e->copy_ev[evi], cs));
        RB_HIP(hipStreamWaitEvent(st, e->copy_ev[evi], 0));
        rb_batch_desc desc;
        std::memset(&desc, 0, sizeof desc);
        desc.d_seqs = d_base;
        desc.d_offsets = d_off + i0;
        desc.d_lens = d_len + i0;
        desc.n_items = cnt;
        desc.max_len = max_len;
        if (direct_out)
            rc = classify_device_impl(e, &desc, error_rate, significance, mode, d_max + i0 * nf, hout + 4 * i0, hout + 4 * n + 2 * nf * n + i0,
                                      hout + 4 * n + 2 * nf * n + n + i0, (void *)st, (uint16_t *)(hout + 4 * n) + i0 * nf);
        else
            rc = classify_device_impl(e, &desc, error_rate, significance, mode, d_max + i0 * nf, sharded ? nullptr : (int32_t *)e->d_best.p + i0,
                                      sharded ? nullptr : (uint8_t *)e->d_decision.p + i0, sharded ? nullptr : (uint8_t *)e->d_status.p + i0, (void *)st,
                                      nullptr);
        if (rc != RB_OK) {
            (void)hipStreamSynchronize(cs);
            (void)hipStreamSynchronize(st);
            return rc;
        }
        i0 = i1;
        ++slice;
    }
    if (direct_out) {
        RB_HIP(hipStreamSynchronize(st));
        if (out_maxcount) std::memcpy(out_maxcount, hout + 4 * n, 2 * nf * n);
        if (out_best_target) std::memcpy(out_best_target, hout, 4 * n);
        if (out_decision) std::memcpy(out_decision, hout + 4 * n + 2 * nf * n, n);
        if (out_status) std::memcpy(out_status, hout + 4 * n + 2 * nf * n + n, n);
        return RB_OK;
    }
    if (out_maxcount) RB_HIP(hipMemcpyAsync(out_maxcount, d_max, n * nf * 2, hipMemcpyDeviceToHost, st));
    if (!sharded) {
        if (out_best_target) RB_HIP(hipMemcpyAsync(out_best_target, e->d_best.p, n * 4, hipMemcpyDeviceToHost, st));
        if (out_decision) RB_HIP(hipMemcpyAsync(out_decision, e->d_decision.p, n, hipMemcpyDeviceToHost, st));
        if (out_status) RB_HIP(hipMemcpyAsync(out_status, e->d_status.p, n, hipMemcpyDeviceToHost, st));
    }
    RB_HIP(hipStreamSynchronize(st));
    return RB_OK;
}

// pointer-array form: read i = seq_ptrs[i][0 .. lens[i]) -- what a basecaller hands over (one buffer per read)
int rb_classify_batch_ptrs(rb_engine *e, const char *const *seq_ptrs, const uint32_t *lens, size_t n_reads,
                           double error_rate, double significance, int mode, uint16_t *out_maxcount,
                           int32_t *out_best_target, uint8_t *out_decision, uint8_t *out_status)
{
    if (!e) return rb::fail(RB_ERR_INVALID_ARG, "null engine");
    if (n_reads == 0) return RB_OK;
    if (!seq_ptrs || !lens) return rb::fail(RB_ERR_INVALID_ARG, "null input buffer");
    // the reads are gathered once on the host; rb_classify_batch then stages them (pinned, one copy) or copies the block
    uint64_t sum = 0;
    for (size_t i = 0; i < n_reads; ++i) {
        if (lens[i] && !seq_ptrs[i]) return rb::fail(RB_ERR_INVALID_ARG, "null read pointer");
        sum += lens[i];
    }
    std::vector<char> flat((size_t)sum + 1);
    std::vector<uint64_t> offsets(n_reads);
    uint64_t pos = 0;
    for (size_t i = 0; i < n_reads; ++i) {
        offsets[i] = pos;
        if (lens[i]) std::memcpy(flat.data() + pos, seq_ptrs[i], lens[i]);
        pos += lens[i];
    }
    return rb_classify_batch(e, flat.data(), offsets.data(), lens, n_reads, error_rate, significance, mode, out_maxcount,
                             out_best_target, out_decision, out_status);
}

}  // extern "C"

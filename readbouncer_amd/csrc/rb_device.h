// rb_device.h -- kernel argument structs and launcher declarations (HIP translation units only)
#pragma once
#include <hip/hip_runtime.h>

#include "rb_internal.h"

namespace rb {

constexpr unsigned kMaxFilters = 16;

// by-value kernel argument describing one HBM-resident IBF
struct IbfDev {
    const uint64_t *words;  // block-major bit matrix, reference layout
    uint64_t magic;         // floor(2^64 / n_blocks) for the Barrett reduction
    uint64_t precalc[rbspec::kMaxHash];
    uint32_t n_blocks;
    uint32_t pow2_mask;  // n_blocks - 1 when n_blocks is a power of two, else 0xFFFFFFFF
    uint32_t n_bins;
    uint32_t bin_width;
    uint32_t stride;  // words from one block to the next in HBM (>= bin_width; padded layout)
    uint32_t k;
    uint32_t n_hash;
    uint32_t comp_n;  // ordinal the reverse strand holds for an N of the read (rbspec::kRevCompOfN unless the engine was told otherwise)
};

// where the bases of a batch live (by-value kernel argument)
struct ReadSrc {
    const uint8_t *seqs;            // ASCII bytes, or the 2-bit payload when nmask != nullptr
    const uint64_t *offsets;        // per READ: byte offset of its bases / of its packed payload
    const uint32_t *lens;           // per ITEM: effective length (whole read, or the chunk after chunk_prep)
    const uint8_t *nmask;           // packed input only: N bitmap payload
    const uint64_t *nmask_offsets;  // per READ: byte offset of its N bitmap
    const uint32_t *ids;            // optional item -> read indirection (nullptr = identity)
    uint32_t base_off;              // bases skipped at the start of every read (chunk start)
    uint32_t max_len;               // declared upper bound of the item lengths: K1 never looks at more bases of an item than this (a longer
                                    // item is answered with RB_ERR_INVALID_ARG by the decision kernel; 0 = no bound given)
};

// Clock-phased gathers for tables of a few L2 sizes (narrow filters; rb_kernels.hip, "phased form"): the table is cut into
// n_slices runs of 2^shift blocks, and window number (wall clock * inv_ticks) >> 32 tells the whole chip which slice to
// gather from.  n_slices == 0: off.
struct PhaseCfg {
    uint32_t shift;      // log2 blocks per slice; with bit 31 set the low bits are blocks per slice, any number (stride-4 one-lane builds)
    uint32_t n_slices;   // ceil(n_blocks / 2^shift), <= 32 (the engine plans <= 8)
    uint32_t inv_ticks;  // floor(2^32 / window length in 10 ns ticks)
    uint32_t skew;       // added to the window number: 0, or this wave's XCD number when xcd_skew is set (experiment)
    uint32_t xcd_skew;   // 1: every XCD works on a different slice at any time (slice = (window + XCD) mod n_slices)
    uint32_t tskew;      // wall-clock ticks by which each XCD's windows start later than the previous XCD's (the kernel multiplies by its XCD
                         // number): the XCDs then refill their L2s one after the other, not all in the same instant.  0: one clock for the chip
};

constexpr unsigned kMaxFused = 8;
constexpr int kSplitAnyWaves = 8;  // waves per workgroup of the mixed-geometry latency kernel (built for 512 threads)

// by-value kernel argument: up to 8 filters served by one launch (blockIdx.y picks one).  The throughput kernel takes
// filters of one kernel geometry; the latency kernel takes any mix and reads the geometry per filter.
struct FilterSet {
    uint32_t n;
    uint32_t col_begin[kMaxFused], col_end[kMaxFused];
    uint32_t out_offset[kMaxFused];  // element offset of the filter's column in the output
    uint32_t geom[kMaxFused];        // latency kernel: lg | (wpl == 2 ? 8 : 0) | (nt ? 16 : 0)
    uint32_t parts[kMaxFused];       // latency kernel: workgroups per (read, slice) that work for this filter (<= grid parts)
    uint32_t sub[kMaxFused];         // latency kernel: shares per macro tile
    IbfDev f[kMaxFused];
};

// Filters of one hash geometry merged into ONE table (rb_engine.hip, MergedGroup): which BITS of a merged block belong to which
// filter.  Member g owns the bins [bit_begin[g], bit_end[g]) of every block -- word-aligned when every member keeps whole word
// columns, packed bit to bit when that saves a word column (README shape: 122 + 43 + 29 + 49 bins = 243 bits = four words instead
// of five, which takes the table to the one-lane-per-block builds of the phased kernel).  By-value kernel argument of
// ibf_count_max_merged_kernel.
constexpr unsigned kMaxMerged = 16;
struct MergeMap {
    uint32_t n;                       // filters in the merged table
    uint32_t bit_begin[kMaxMerged];   // first bin of filter g in a merged block
    uint32_t bit_end[kMaxMerged];     // one past its last bin
    uint32_t out_offset[kMaxMerged];  // element offset of filter g's column in a row of the output
    uint32_t width;                   // word columns of a merged block
};

// bins of member [begin, end) that lie in word column c of a merged block, as a mask of that word
__host__ __device__ inline uint64_t member_mask(uint32_t begin, uint32_t end, uint32_t c)
{
    const uint32_t lo = c * 64u, hi = lo + 64u;
    const uint32_t b = begin > lo ? begin : lo, e = end < hi ? end : hi;
    if (b >= e) return 0ULL;
    const uint64_t upto_e = (e - lo) >= 64u ? ~0ULL : ((1ULL << (e - lo)) - 1);
    const uint64_t upto_b = (1ULL << (b - lo)) - 1;  // b - lo < 64
    return upto_e & ~upto_b;
}

// Merged tables of at most four word columns (two to eight narrow filters of one hash geometry, rb_engine.hip) are served by the
// both-strands builds of the phased kernel, which hold a whole block in one lane: which bins belong to which member (bit ranges as
// in MergeMap), how many of a column's 64 bits are bins at all (0: the column does not exist), and where each member's maximum
// goes in a row of the output.  A filter on its own is the case n = 1.
constexpr unsigned kMaxNarrow = 8;
struct NarrowMerge {
    uint32_t n;                       // members (1..8)
    uint32_t bit_begin[kMaxNarrow];   // member -> first bin in a block
    uint32_t bit_end[kMaxNarrow];     // member -> one past its last bin
    uint32_t col_bits[4];             // word column -> bins in it (64: all; 0: no such column)
    uint32_t out_offset[kMaxNarrow];  // member -> element offset in a row of the output
};

struct CountLaunch {
    IbfDev f;
    ReadSrc src;
    uint32_t n_reads;               // number of work items
    uint32_t col_begin, col_end;  // word columns of every block this launch covers
    uint32_t n_slices;            // column slices of 2^lg * wpl words
    int lg, wpl, planes;
    int nt;                       // non-temporal table gathers (tables beyond the Infinity Cache)
    PhaseCfg phase;               // throughput form on narrow filters: clock-phased gathers (n_slices == 0: off)
    NarrowMerge narrow;           // phased form on two- to four-word blocks: columns -> members (the engine fills it; n = 1: one filter)
    int phase_shape;              // engine bookkeeping (rbplan::PhaseShape of the planner's row for this launch; the kernels do not read it)
    uint32_t phase_slice_log2, phase_ticks;  // ... the slice size and window length in effect (10 ns ticks), for rb_engine_plan
    uint32_t phase_rule_ticks;               // ... and what the planner's table alone would give (rb_engine_calibrate may have replaced it)
    uint64_t phase_slice_bytes;              // ... and the slice length in effect (2^phase_slice_log2, or the equal-length slices of the four-word builds)
    int short_only;               // 1 / 3 / 2: the declared max_len gives at most 256 / 384 / 512 k-mers per read; 0: more
    int multi_reads;              // phased form, two-word blocks, short_only 1: reads per wave of ibf_count_max_phased_multi_kernel (0: the one-read build)
    int multi_tiles;              // ... tiles per strand of that build: 4 (reads of up to 256 k-mers) or 6 (up to 384)
    int multi_inv;                // ... and f.words is the COMPLEMENTED twin of a merged copy (the build that ORs instead of masking and ANDing)
    int split_waves;              // >= 2: latency form, workgroups of split_waves waves
    int split_parts, split_sub;   // latency form: workgroups per (read, slice) and shares per macro tile (0/1 = one workgroup)
    int grid_parts;               // latency form: workgroups launched per (read, slice) = max parts of the fused filters
    uint64_t *split_ws;           // parts > 1: partial counters [filter][item][grid part][strand][2][planes][64]
    uint32_t *split_tickets;      // parts > 1: arrival counters [filter][item], zero between launches
    uint16_t *out;
    uint32_t out_read_stride, out_slice_stride;
    // latency form only: n_fused > 0 = several filters in one launch (then f/col_begin/col_end above describe the first)
    int n_fused;
    IbfDev fused_f[kMaxFused];
    uint32_t fused_col_begin[kMaxFused], fused_col_end[kMaxFused], fused_out_offset[kMaxFused];
    uint8_t fused_geom[kMaxFused], fused_parts[kMaxFused], fused_sub[kMaxFused];  // latency form, see FilterSet
    const struct FoldJob *fold;   // latency form: the launch also makes the decisions (nullptr: it only counts)
    // opt-in early decision (plain throughput form, RB_MODE_CHECK_UNBLOCK without raw maxima): the decision kernel's threshold table, so that
    // a wave can stop counting once a bin has reached the larger of the read's two thresholds for this filter (nullptr: count everything)
    const uint16_t *early_thr;
    uint32_t early_thr_len, early_nf, early_fi;
};

inline uint8_t geom_code(int lg, int wpl, int nt) { return (uint8_t)(lg | (wpl == 2 ? 8 : 0) | (nt ? 16 : 0)); }

struct DecideParams {
    uint32_t nd, nt;
    uint32_t k[kMaxFilters];
    const uint16_t *thr;  // [thr_len][nf][2]: thresholds at r and at r-0.02 by read length
    uint32_t thr_len;
    uint32_t max_len;  // declared upper bound of the read lengths of this batch
    uint16_t *maxcount_copy;  // optional second destination of the maxcount rows (pinned host memory of the micro-batch path)
    // bin-sharded operation: maxcount holds n_parts partial tables (one per rank, all-gathered), part_stride elements apart;
    // the raw maximum of a (read, filter) is the max over them.  1 = a plain table.
    uint32_t n_parts;
    uint64_t part_stride;
    // Completion word of the host micro-batch path (nullptr: none): when every result of the call is written, done_seq is stored -- behind a
    // system-scope release -- into this word of page-locked host memory, and the host, which spins on it, does not wait for the stream
    // (hipStreamSynchronize returns 4 us later than the word arrives: profiles/r05/graph_launch_probe.txt).  Decision kernels of more than one
    // workgroup count their arrivals in done_count (zero between calls).
    uint32_t *done_flag;
    uint32_t done_seq;
    uint32_t *done_count;
};

// The decision of a micro-batch made by the latency kernel itself, for engines with ONE filter: the workgroup that writes a read's raw
// maximum runs the decision for that read, and the dependent launch of the decision kernel goes away (1.1-1.8 us of a 40-120 us call
// up to 256 reads; with several filters the reads would need arrival counters of their own, and one more agent-scope release per
// (read, filter) costs more than the launch did: profiles/r05/negative_results.md, entry 10).  By-value kernel argument.
struct FoldJob {
    uint32_t on;  // 0: the kernel only counts
    int mode;
    const uint16_t *maxcount;  // the table [n_reads][1] the launch writes into
    const uint32_t *lens;
    const uint8_t *pre_status;
    int32_t *best_target;
    uint8_t *decision;
    uint8_t *status;
    DecideParams P;
};

hipError_t launch_ibf_count_max(const CountLaunch &a, hipStream_t st);
hipError_t launch_ibf_count_max_merged(const CountLaunch &a, const MergeMap &map, hipStream_t st);
// block b of a filter (width words at stride s_src, n_bins bins) -> bits [dst_bit, dst_bit + n_bins) of block b of dst (ORed in: dst starts zeroed)
hipError_t launch_invert_words(const uint64_t *src, uint64_t *dst, uint64_t n_words, hipStream_t st);
hipError_t launch_merge_bits(const uint64_t *src, uint32_t s_src, uint32_t width, uint32_t n_bins, uint64_t *dst, uint32_t s_dst, uint32_t dst_bit,
                             uint64_t n_blocks, hipStream_t st);
int split_waves_limit(int wpl, int planes, uint32_t max_kmers, int lg);
int split_waves_cap(int wpl, int planes, int lg);
int split_parts_plan(int wpl, int planes, uint32_t max_kmers, int lg, uint32_t n_items, uint32_t max_parts, uint32_t max_sub,
                     int *nw, int *sub);
hipError_t launch_reduce_slices(const uint16_t *part, uint32_t n_slices, uint32_t n_reads, uint16_t *maxcount,
                                uint32_t nf, uint32_t fidx, hipStream_t st);
hipError_t launch_decide(const DecideParams &P, const uint16_t *maxcount, const uint32_t *lens, const uint8_t *pre_status,
                         uint32_t n_reads, int mode, int32_t *best_target, uint8_t *decision, uint8_t *status,
                         hipStream_t st);
// micro-batches: pinned host block -> device staging by a kernel of the call's own stream (16-byte units; both blocks hold whole units)
hipError_t launch_copy_from_host(const void *h_src, void *d_dst, size_t bytes, hipStream_t st);
hipError_t launch_chunk_prep(const uint32_t *lens, const uint32_t *ids, uint32_t n_items, uint32_t chunk_start,
                             uint32_t chunk_len, uint32_t *eff_lens, uint8_t *pre_status, hipStream_t st);
hipError_t launch_insert(const IbfDev &f, uint64_t *words, const uint8_t *seq, const uint64_t *starts,
                         const uint64_t *ends, const uint64_t *bins, const uint64_t *kmer_prefix,
                         uint32_t n_fragments, uint64_t total_kmers, hipStream_t st);
hipError_t launch_restride_blocks(const uint64_t *src, uint32_t s_src, uint64_t *dst, uint32_t s_dst, uint32_t w_copy,
                                  uint64_t n_blocks, hipStream_t st);
hipError_t launch_compare_bits(const uint64_t *a, const uint64_t *b, uint64_t n_words, uint64_t *out3, hipStream_t st);
hipError_t launch_fill_reads(uint8_t *seqs, uint64_t *offsets, uint32_t *lens, size_t n_reads, uint32_t read_len, uint64_t seed, hipStream_t st);
hipError_t launch_fill_synth(uint64_t *words, uint64_t used_words, uint32_t bin_width, uint32_t stride_words,
                             uint64_t last_mask, uint64_t seed, hipStream_t st);

}  // namespace rb

// rb_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the IBF classify path.
//
//   K1 ibf_count_max : seqan::count(filter, seq) + seqan::count(filter, TSeqRevComp(seq)) +
//                      the max over bins/strands of Read::max_matches
//                      (src/IBF/IBFClassify.cpp:149-150, 48-71); fuses the (Dna5String) conversion
//                      of src/main/classify.hpp:272.
//   K2 decide        : threshold + argmax + decision (IBFClassify.cpp:154-162, 262-273, 299-365;
//                      src/main/adaptive_sampling.hpp:35-113; src/main/classify.hpp:58-111,275-292)
//   K4 ibf_insert    : seqan::insertKmer (src/IBF/IBFBuild.cpp:190)
//   fill_synth       : benchmark filler
//
// Design (DESIGN.md has the long form).  The path is integer/bitwise and HBM-gather bound: per read
// and filter 2*(L-k+1)*h random blocks of 8*W bytes.  One 64-lane wave owns one (read, column slice):
//   * phase A: every lane hashes its own k-mer of a 64-k-mer tile (base-5 value, h multiplicative
//     hashes, Barrett reduction mod noOfBlocks) -- the block numbers stay in registers;
//   * phase B: the wave walks the tile; a group of LPB = 2^LG lanes covers one block with one
//     8-byte (or 16-byte) word column per lane, so every wave-level load instruction is 64/LPB
//     fully coalesced block gathers; block numbers travel by ds_bpermute / v_readlane;
//   * per-bin counters are BIT-SLICED: lane-private 64-bit planes, one bit per bin, fed by a
//     Harley-Seal carry-save tree (7 CSAs per 8 gathered words) -- no atomics, no LDS traffic,
//     10 or 16 planes (16 planes wrap at 65536 exactly like the reference's uint16_t counters);
//   * the max over bins is taken on the bit-sliced form with 64-wide ballots, MSB plane first.
// Forms that share that body (count_strand): the throughput form (one wave per read and column slice, both strands in
// sequence; slices of a read sit side by side in a workgroup), its MERGED variant (several narrow filters of one hash geometry
// in one table, one gather per lookup for all of them), the PHASED variant for narrow filters on their own (clock-phased
// slices of the table, bounds-checked buffer gathers, both strands in one round) and the latency form for micro-batches (one or several
// workgroups per read; their waves split strands and k-mer tiles, add their bit-sliced counters through LDS and -- across
// workgroups -- through a workspace and an arrival counter; one launch serves filters of different geometries).  Gathers are
// issued in batches of 12-24 per wave with no control flow around them and a schedule fence before the first use;
// filters beyond the Infinity Cache are read with non-temporal loads.  Inputs may be ASCII or 2-bit packed reads, whole
// or chunked on the device (BaseSrc / ReadSrc).
// No MFMA anywhere: there is no multiply-accumulate structure in this path.
#include <hip/hip_runtime.h>

#include <atomic>

#include "rb_device.h"

namespace rb {

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t shfl32(uint32_t v, int src_lane)
{
    return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);
}
__device__ __forceinline__ uint64_t shfl64(uint64_t v, int src_lane)
{
    uint32_t lo = shfl32((uint32_t)v, src_lane), hi = shfl32((uint32_t)(v >> 32), src_lane);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t readlane32(uint32_t v, int uniform_lane)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)v, uniform_lane);
}

// Table gathers.  NT = non-temporal: for a table far beyond the 256 MiB Infinity Cache every block is touched
// once and caching it only evicts something else (measured on config 3: +2.4 %); a table that fits the cache
// keeps the default policy (config 2 loses 1.9 % with NT).  The engine picks per filter by table size.
typedef unsigned long long rb_u64x2 __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ uint64_t load_word(const uint64_t *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT>
__device__ __forceinline__ rb_u64x2 load_word2(const uint64_t *p)
{
    if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const rb_u64x2 *>(p));
    else return *reinterpret_cast<const rb_u64x2 *>(p);
}

// number of the XCD this wave runs on (0-7): a placement fact, used for speed only
__device__ __forceinline__ uint32_t xcc_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xFu;
}

// carry-save adder on bit planes: (h, l) = a + b + c per bit position
#define RB_CSA(h, l, a, b, c)                \
    {                                        \
        const uint64_t u__ = (a) ^ (b);      \
        const uint64_t c__ = (c);            \
        (h) = ((a) & (b)) | (u__ & c__);     \
        (l) = u__ ^ c__;                     \
    }

template <int NP>
struct Planes {
    uint64_t p[NP];  // p[i] holds bit i of 64 per-bin counters
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int i = 0; i < NP; ++i) p[i] = 0;
    }
    // fold 8 one-bit-per-bin words into the counters (Harley-Seal)
    __device__ __forceinline__ void add8(const uint64_t x[8])
    {
        uint64_t twosA, twosB, foursA, foursB, eights;
        RB_CSA(twosA, p[0], p[0], x[0], x[1]);
        RB_CSA(twosB, p[0], p[0], x[2], x[3]);
        RB_CSA(foursA, p[1], p[1], twosA, twosB);
        RB_CSA(twosA, p[0], p[0], x[4], x[5]);
        RB_CSA(twosB, p[0], p[0], x[6], x[7]);
        RB_CSA(foursB, p[1], p[1], twosA, twosB);
        RB_CSA(eights, p[2], p[2], foursA, foursB);
        uint64_t carry = eights;
#pragma unroll
        for (int i = 3; i < NP; ++i) {
            const uint64_t t = p[i] & carry;
            p[i] ^= carry;
            carry = t;
        }
    }
    // counters += counters held by lane (lane ^ xor_mask)
    __device__ __forceinline__ void add_from_lane_xor(int lane, int xor_mask)
    {
        uint64_t carry = 0;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const uint64_t o = shfl64(p[i], lane ^ xor_mask);
            uint64_t h, l;
            RB_CSA(h, l, p[i], o, carry);
            p[i] = l;
            carry = h;
        }
    }
};

// 64 x 64 bit-matrix transpose across the wave: bit j of lane i <-> bit i of lane j (six block swaps)
__device__ __forceinline__ uint64_t wave_transpose64(uint64_t x, int lane)
{
    constexpr uint64_t kMask[6] = {0x00000000FFFFFFFFULL, 0x0000FFFF0000FFFFULL, 0x00FF00FF00FF00FFULL,
                                   0x0F0F0F0F0F0F0F0FULL, 0x3333333333333333ULL, 0x5555555555555555ULL};
    // first swap (32 x 32 blocks): the high words of lanes 0-31 change places with the low words of lanes 32-63 -- one v_permlane32_swap
    // (gfx950) instead of two ds_bpermute and the selects around them
    {
        const auto r = __builtin_amdgcn_permlane32_swap((uint32_t)x, (uint32_t)(x >> 32), false, false);
        x = ((uint64_t)r[1] << 32) | r[0];
    }
#pragma unroll
    for (int t = 1; t < 6; ++t) {
        const int sft = 32 >> t;
        const uint64_t m = kMask[t];
        const uint64_t o = shfl64(x, lane ^ sft);
        x = (lane & sft) ? ((x & ~m) | ((o & ~m) >> sft)) : ((x & m) | ((o & m) << sft));
    }
    return x;
}

// per-bin counts of 4 one-bit-per-bin words per lane, summed over the 64 lanes: lane b returns the count of bin b.
// The lane-local sum (0..4 per bin) is three bit planes; each plane is transposed across the wave, so that lane b holds
// bit b of every lane's plane word, and a popcount finishes the sum -- 36 cross-lane moves instead of the 120 of a
// butterfly over ten counter planes.
__device__ __forceinline__ uint32_t wave_bin_counts4(uint64_t x0, uint64_t x1, uint64_t x2, uint64_t x3, int lane)
{
    uint64_t h, l;
    RB_CSA(h, l, x0, x1, x2);
    const uint64_t c = l & x3;
    const uint64_t p0 = l ^ x3, p1 = h ^ c, p2 = h & c;
    return (uint32_t)__popcll(wave_transpose64(p0, lane)) + 2u * (uint32_t)__popcll(wave_transpose64(p1, lane)) +
           4u * (uint32_t)__popcll(wave_transpose64(p2, lane));
}

// the same for 6 one-bit-per-bin words per lane (sums 0..6: still three planes, three transposes)
__device__ __forceinline__ uint32_t wave_bin_counts6(uint64_t x0, uint64_t x1, uint64_t x2, uint64_t x3, uint64_t x4,
                                                     uint64_t x5, int lane)
{
    uint64_t h1, l1, h2, l2, p2, p1;
    RB_CSA(h1, l1, x0, x1, x2);
    RB_CSA(h2, l2, x3, x4, x5);
    const uint64_t p0 = l1 ^ l2, c = l1 & l2;
    RB_CSA(p2, p1, h1, h2, c);
    return (uint32_t)__popcll(wave_transpose64(p0, lane)) + 2u * (uint32_t)__popcll(wave_transpose64(p1, lane)) +
           4u * (uint32_t)__popcll(wave_transpose64(p2, lane));
}

template <int T>
__device__ __forceinline__ uint32_t wave_bin_counts(const uint64_t *x, int lane)
{
    if constexpr (T == 6) return wave_bin_counts6(x[0], x[1], x[2], x[3], x[4], x[5], lane);
    else if constexpr (T == 2)  // sums 0..2: two planes, two transposes
        return (uint32_t)__popcll(wave_transpose64(x[0] ^ x[1], lane)) + 2u * (uint32_t)__popcll(wave_transpose64(x[0] & x[1], lane));
    else if constexpr (T == 3) {  // sums 0..3: one carry-save step, two planes
        uint64_t h, l;
        RB_CSA(h, l, x[0], x[1], x[2]);
        return (uint32_t)__popcll(wave_transpose64(l, lane)) + 2u * (uint32_t)__popcll(wave_transpose64(h, lane));
    }
    else return wave_bin_counts4(x[0], x[1], x[2], x[3], lane);
}

// max over all bins held by the wave (WPL plane sets per lane), MSB plane first
template <int NP, int WPL>
__device__ __forceinline__ uint32_t planes_max(const Planes<NP> (&pl)[WPL], const uint64_t (&valid)[WPL])
{
    uint64_t cand[WPL];
#pragma unroll
    for (int w = 0; w < WPL; ++w) cand[w] = valid[w];
    uint32_t res = 0;
#pragma unroll
    for (int i = NP - 1; i >= 0; --i) {
        uint64_t t[WPL];
        bool any = false;
#pragma unroll
        for (int w = 0; w < WPL; ++w) {
            t[w] = cand[w] & pl[w].p[i];
            any |= (t[w] != 0);
        }
        if (__ballot(any) != 0ULL) {  // wave-uniform
            res |= 1u << i;
#pragma unroll
            for (int w = 0; w < WPL; ++w) cand[w] = t[w];
        }
    }
    return res;
}

// steps of the plain kernels whose gathers go out together (8-byte lanes / 16-byte lanes).  The wide kernels are HBM bound
// and do not care: 4 waves per SIMD with half the loads in flight run at the same 0.864-0.866 of 8 TB/s on config 3
// (profiles/r03/window_sweep.txt, session 19)
#ifndef RB_HALF1
#define RB_HALF1 8
#endif
#ifndef RB_HALF2
#define RB_HALF2 4
#endif

// ---------------------------------------------------------------------------------------------
// K1.  LG: log2(lanes per block); WPL: 64-bit word columns per lane (1 or 2); NP: counter planes;
// H: compile-time number of hash functions, 0 = run-time (slow path, hashes in phase B).
constexpr int kWavesPerBlock = 4;
constexpr int kMaxTiles = 8;                                      // tiles per macro tile when LG < 3
constexpr int kStageBytes = 64 * kMaxTiles + rbspec::kMaxKmer;    // bases staged per macro tile

template <int LG>
struct TileShape {
    static constexpr int LPB = 1 << LG;                 // lanes that cover one block
    static constexpr int NG = 64 >> LG;                 // blocks gathered per wave instruction
    static constexpr int SPT = LPB;                     // phase-B steps per 64-k-mer tile
    static constexpr int J = SPT >= 8 ? 1 : 8 / SPT;    // tiles per macro tile (so that steps come in eights)
    static constexpr int STEPS = SPT * J;
    static constexpr int ITEMS = 64 * J;                // k-mers per macro tile
};

// per-lane view of the column slice a wave works on
template <int WPL>
struct LaneCols {
    uint64_t valid[WPL];        // bins of this lane's word(s) that exist (padding and foreign columns masked)
    const uint64_t *lane_base;  // f.words + first word column of this lane
    const uint64_t *safe_base;  // lane_base, or f.words for a lane that owns no (complete) column: always loadable
    bool colok;                 // lane owns at least one existing column
    bool col_full;              // WPL == 2: both words exist, a 16-byte load is allowed
};

template <int LG, int WPL>
__device__ __forceinline__ LaneCols<WPL> make_lane_cols(const IbfDev &f, int lane, uint32_t col_begin, uint32_t col_end,
                                                        uint32_t slice)
{
    constexpr int LPB = 1 << LG;
    const int c = lane & (LPB - 1);
    const uint32_t W = f.bin_width;
    const uint32_t col0 = col_begin + slice * (uint32_t)(LPB * WPL) + (uint32_t)(c * WPL);
    LaneCols<WPL> lc;
    lc.colok = false;
#pragma unroll
    for (int w = 0; w < WPL; ++w) {
        const uint32_t col = col0 + w;
        const bool ok = col < col_end;
        const uint32_t rem = f.n_bins & 63u;
        lc.valid[w] = !ok ? 0ULL : (col == W - 1 && rem) ? ((1ULL << rem) - 1) : ~0ULL;
        lc.colok |= ok;
    }
    lc.col_full = (col0 + WPL) <= col_end;
    lc.lane_base = f.words + col0;
    // WPL == 2 with an odd column count: the last lane owns one column; its 16-byte load also fetches the word
    // after it (next column of the block, or the first word past the block -- still inside the filter thanks to the
    // 256 metadata bits at the tail) and valid[1] == 0 masks it.  Blocks of an odd-width filter are only 8-byte
    // aligned: gfx950 executes 16-byte global loads at 8-byte alignment (checked on the device).
    lc.safe_base = lc.colok ? lc.lane_base : f.words;
    return lc;
}

// Bases of one read as the kernel sees them: ASCII bytes (nm == nullptr) or the packed form of SURVEY 8f.4 --
// 2 bits per base (A0 C1 G2 T3; base i in bits 2*(i&3) of byte i>>2) plus an N bitmap (bit i&7 of byte i>>3) that
// turns a base into Dna5 ordinal 4.  `first` = bases skipped at the start of the read (on-GPU chunking).
struct BaseSrc {
    const uint8_t *bytes;  // ASCII: already advanced to the first base; packed: start of the read's 2-bit payload
    const uint8_t *nm;     // packed: start of the read's N bitmap; ASCII: nullptr
    uint32_t first;        // packed: index of the first base of the chunk inside the read
    __device__ __forceinline__ uint32_t ord(uint32_t i) const
    {
        if (nm == nullptr) return rbspec::dna5_ord(bytes[i]);
        const uint32_t b = first + i;
        const uint32_t code = (bytes[b >> 2] >> ((b & 3u) << 1)) & 3u;
        return ((nm[b >> 3] >> (b & 7u)) & 1u) ? 4u : code;
    }
};

// Batch sizes of the phased window loop (phased_window_loop below): k-mers of a lane whose gathers go out together (B: one-word blocks, three 8-byte loads per k-mer; KB: two-word blocks,
// three 16-byte loads per k-mer).  Every load in flight holds its destination registers, so the batch size sets the
// occupancy: with ALL lookups of a window in flight (8 k-mers, 48 registers) the 250 bp one-word kernel has five waves per
// SIMD, with two k-mers per batch six (73 VGPRs) -- four waits per window instead of one, and 7 % less time per read; the
// six-tile kernel goes from four to five waves (93 VGPRs), the two-word 250 bp kernel from four to five (86).  Measured
// per shape in profiles/r03/window_sweep.txt; the window lengths of rb_engine.hip belong to these values.
#ifndef RB_GATHER_B1
#define RB_GATHER_B1 2
#endif
#ifndef RB_GATHER_KB1
#define RB_GATHER_KB1 1
#endif
#ifndef RB_GATHER_B3
#define RB_GATHER_B3 2
#endif
#ifndef RB_GATHER_KB3
#define RB_GATHER_KB3 2
#endif
#ifndef RB_WIDE_WAVES  // waves per SIMD the four-tile build for THREE-word blocks is compiled for (94 registers, no scratch; the
                       // four-word build would spill 104 bytes at five waves and stays at four: 116 registers)
#define RB_WIDE_WAVES 5
#endif
#ifndef RB_WIDE_TILES  // tiles per strand and round of the three- / four-word build for reads of 257-512 k-mers
#define RB_WIDE_TILES 3
#endif
#ifndef RB_GATHER_BG  // per-strand tiles of the general build (reads of more than 512 k-mers, blocks of 3-8 words)
#define RB_GATHER_BG 2
#endif
typedef unsigned int rb_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int rb_u32x4 __attribute__((ext_vector_type(4)));
constexpr int kBufRsrcWord3 = 0x00020000;  // raw buffer, DATA_FORMAT = 32 bit (gfx9 family)

// Which slice a wave gathers next: the one the wall clock names NOW, if the wave has not served it yet; otherwise the wave
// sleeps until the clock has moved on.  A wave that is ahead of the clock therefore waits for its next window, and a wave
// that has fallen behind does NOT work through the windows it missed -- their slices have left the L2 -- but joins the chip on
// the current slice and picks the missed ones up when they come round again (round 2 walked the slices in a fixed order:
// a window that was too short for the waves then cost a factor of two, profiles/r03/window_sweep.txt).  Bounded: if the
// clock does not move for ~0.1 ms (its 32-bit wrap, a window length of 0) the lowest slice still open is taken.
__device__ __forceinline__ uint32_t phase_next_slice(uint32_t done, const PhaseCfg &ph)
{
    for (;;) {
        const uint32_t wn = (uint32_t)(((uint64_t)((uint32_t)wall_clock64() + ph.tskew) * ph.inv_ticks) >> 32);
        const uint32_t cur = (wn + ph.skew) % ph.n_slices;
        if (!((done >> cur) & 1u)) return cur;
        bool moved = false;
        for (uint32_t guard = 0; guard < 2048; ++guard) {
            __builtin_amdgcn_s_sleep(2);
            const uint32_t w2 = (uint32_t)(((uint64_t)((uint32_t)wall_clock64() + ph.tskew) * ph.inv_ticks) >> 32);
            if (w2 != wn) { moved = true; break; }
        }
        if (!moved) return (uint32_t)__builtin_ctz(~done);  // done has a zero bit below n_slices: the caller loops while it does
    }
}

// The window loop of the phased form -- ONE body for every block width a single lane holds (NW = 1 to 4 words; the three entry
// points below only name the load widths): x*[u] &= the words of the block at byte offset bn[u][*] of `words`, each gathered in the
// window of its slice.  0xFFFFFFFF = no lookup.  The predication is done by the BOUNDS CHECK of a raw buffer descriptor, not by
// exec masks: per window the wave points the descriptor at the slice of the moment (base = slice start, num_records = slice
// bytes) and issues every lookup as buffer_load with the offset (lookup - slice start): a lane whose lookup lies in another
// slice (or that has none) is out of range, makes no memory access and gets 0 back, which an OR with the lane's out-of-range
// mask turns into the neutral all-ones.  No saveexec, no branch around a load, no scalar work per lookup (the exec-masked loads
// of round 2: ~11 instructions per lookup and window, seven of them scalar or branches, every destination register kept
// initialised across the window loop).
// How a table is cut is carried by slice_shift (rb_engine.hip, plan_geometry): a plain value = slices of 2^slice_shift bytes
// (31: one slice that holds every offset -- with ph = {n_slices 1, inv_ticks 0} that is one batch of gathers with no waiting,
// for tables that need no phasing); bit 31 set + a non-zero length in the low bits = slices of that many BYTES, any multiple of
// the block size (equal-length slices, rb_phase_plan.h).  Blocks never straddle a slice.
// Loads per lookup: NW 1 = one 8-byte, 2 = one 16-byte, 3 = 16 + 8 bytes, 4 = two 16-byte (blocks of 3 and 4 words lie at a
// stride of 4 words, 32-byte aligned).  K k-mers of a lane go out together (KB): every load in flight holds its destination
// registers, so the batch size sets the occupancy (see RB_GATHER_* above).
template <int NW, int N, int H, int KB>
__device__ __forceinline__ void phased_window_loop(uint64_t (&x0)[N], uint64_t (&x1)[N], uint64_t (&x2)[N], uint64_t (&x3)[N],
                                                   const uint32_t (&bn)[N][H], const uint64_t *words, uint32_t slice_shift,
                                                   const PhaseCfg ph)
{
    static_assert(NW >= 1 && NW <= 4, "one lane holds blocks of one to four words");
    static_assert(N % KB == 0, "the k-mers of a lane are gathered in batches of KB");
    const uint32_t all = ph.n_slices >= 32 ? ~0u : (1u << ph.n_slices) - 1u;  // one bit per slice
    uint32_t done = 0;
#pragma unroll 1
    while (done != all) {
        const uint32_t cur = phase_next_slice(done, ph);
        done |= 1u << cur;
        const bool any_len = (slice_shift >> 31) != 0 && slice_shift != 0x80000000u;
        const uint32_t span = any_len ? (slice_shift & 0x7FFFFFFFu) : 1u << min(slice_shift, 31u);  // (slice_shift == 31: a single slice, cur == 0)
        const uint32_t start = any_len ? cur * span : cur << min(slice_shift, 31u);
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(reinterpret_cast<const char *>(words)) + start, 0, (int)span, kBufRsrcWord3);
#pragma unroll
        for (int part = 0; part < N / KB; ++part) {
            rb_u32x2 lo2[KB][H], hi2[KB][H];  // NW 1: the block; NW 3: its third word
            rb_u32x4 lo4[KB][H], hi4[KB][H];  // NW 2-4: words 0-1; NW 4: words 2-3
#pragma unroll
            for (int uu = 0; uu < KB; ++uu) {
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    // (the offsets are made opaque at both uses: otherwise `offset - start` of all lookups is computed up front
                    // and kept for the masks below -- 24 registers, a wave per SIMD on the 250 bp one-word kernel)
                    uint32_t off = bn[part * KB + uu][h];
                    asm volatile("" : "+v"(off));
                    if constexpr (NW == 1) lo2[uu][h] = __builtin_amdgcn_raw_buffer_load_b64(rs, off - start, 0, 0);
                    else lo4[uu][h] = __builtin_amdgcn_raw_buffer_load_b128(rs, off - start, 0, 0);
                    // ("no lookup" stays out of range with bit 4 set)
                    if constexpr (NW == 3) hi2[uu][h] = __builtin_amdgcn_raw_buffer_load_b64(rs, (off - start) | 16u, 0, 0);
                    if constexpr (NW == 4) hi4[uu][h] = __builtin_amdgcn_raw_buffer_load_b128(rs, (off - start) | 16u, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int uu = 0; uu < KB; ++uu) {
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    uint32_t off = bn[part * KB + uu][h];
                    asm volatile("" : "+v"(off));
                    const uint32_t out = (off - start) >= span ? 0xFFFFFFFFu : 0u;  // lanes that loaded nothing
                    const int u = part * KB + uu;
                    if constexpr (NW == 1) {
                        x0[u] &= (((uint64_t)(lo2[uu][h].y | out)) << 32) | (lo2[uu][h].x | out);
                    } else {
                        x0[u] &= (((uint64_t)(lo4[uu][h].y | out)) << 32) | (lo4[uu][h].x | out);
                        x1[u] &= (((uint64_t)(lo4[uu][h].w | out)) << 32) | (lo4[uu][h].z | out);
                    }
                    if constexpr (NW == 3) x2[u] &= (((uint64_t)(hi2[uu][h].y | out)) << 32) | (hi2[uu][h].x | out);
                    if constexpr (NW == 4) {
                        x2[u] &= (((uint64_t)(hi4[uu][h].y | out)) << 32) | (hi4[uu][h].x | out);
                        x3[u] &= (((uint64_t)(hi4[uu][h].w | out)) << 32) | (hi4[uu][h].z | out);
                    }
                }
            }
        }
    }
}

// one-word blocks (8-byte gathers): x = the word of each of the lane's N k-mers
template <int N, int H, bool NT, int B = N>
__device__ __forceinline__ void phased_gather(uint64_t (&x)[N], const uint32_t (&bn)[N][H], const uint64_t *words,
                                              uint32_t slice_shift, const PhaseCfg ph)
{
    phased_window_loop<1, N, H, B>(x, x, x, x, bn, words, slice_shift, ph);
}

// two-word blocks held by ONE lane (16-byte gathers): x0 / x1 = the two word columns of the lane's N k-mers
template <int N, int H, int KB>
__device__ __forceinline__ void phased_gather_x2(uint64_t (&x0)[N], uint64_t (&x1)[N], const uint32_t (&bn)[N][H],
                                                 const uint64_t *words, uint32_t slice_shift, const PhaseCfg ph)
{
    phased_window_loop<2, N, H, KB>(x0, x1, x1, x1, bn, words, slice_shift, ph);
}

// three- and four-word blocks (stride 4 words) held by ONE lane: two gathers per lookup
template <int N, int H, int KB, int NW = 4>
__device__ __forceinline__ void phased_gather_x4(uint64_t (&x0)[N], uint64_t (&x1)[N], uint64_t (&x2)[N], uint64_t (&x3)[N],
                                                 const uint32_t (&bn)[N][H], const uint64_t *words, uint32_t slice_shift,
                                                 const PhaseCfg ph)
{
    static_assert(NW == 3 || NW == 4, "narrower blocks have entry points of their own");
    phased_window_loop<NW, N, H, KB>(x0, x1, x2, x3, bn, words, slice_shift, ph);
}

// PH = clock-phased gathers (PhaseCfg): for a table of a few L2 sizes -- one- and two-word blocks, 10-20 MB -- every
// 8-byte gather that misses the XCD's 4 MiB L2 costs a full 128-byte fabric request, and the chip serves about 60 G of
// those per second whatever their useful size (profiles/r02/gather_probe.txt: 76 G gathers/s on 20 MiB against 269 G/s
// from an L2-resident slice).  So the chip is made to work on ONE slice of the table at a time: the block numbers of a
// macro tile stay in registers (they do anyway), the wall clock (s_memrealtime, the same on every CU) names the slice of
// the moment, and each wave gathers only its lookups that fall into that slice, ANDing them into the per-k-mer words it
// keeps in registers; after n_slices windows every lookup has been served once, in whatever window its slice came up.
// Nothing depends on the timing but the speed: a wave that is ahead of the clock sleeps until its next window opens, one
// that is behind never waits, and the result is the same AND of the same words.
// Counts one strand of one read into the wave's bit-sliced counters, visiting the macro tiles
// mt_first, mt_first + mt_step, ... (mt_step = ITEMS walks the whole read; the split kernel interleaves waves) and of
// each macro tile the eight-step blocks [blk_first, blk_end) (all STEPS / 8 of them, or a wave's share in the split kernel).
// EARLY (the opt-in early-decision mode of the throughput form, rb_engine_set_early_decision): after every macro tile the running
// maximum of this wave's counters is compared with `stop_at`; once it is reached the strand is left (the counters are a lower bound of
// the read's, which is all the decision needs: see ibf_count_max_kernel) and true is returned.
template <int LG, int WPL, int NP, int H, bool NT, bool PH = false, bool EARLY = false>
__device__ __forceinline__ bool count_strand(Planes<NP> (&pl)[WPL], const IbfDev &f, const LaneCols<WPL> &lc,
                                             const BaseSrc &seq, uint32_t len, uint32_t n, int strand,
                                             uint32_t mt_first, uint32_t mt_step, int blk_first, int blk_end,
                                             uint8_t *stage, int lane, const PhaseCfg ph = PhaseCfg{0, 0, 0, 0, 0}, uint32_t stop_at = 0xFFFFFFFFu)
{
    using T = TileShape<LG>;
    constexpr int NG = T::NG, SPT = T::SPT, J = T::J, ITEMS = T::ITEMS;
    constexpr int HR = H > 0 ? H : 1;
    const int g = lane >> LG;
    const uint32_t S = f.stride;  // words between consecutive blocks in HBM (>= bin_width, see rb_engine.hip)
    const uint32_t k = f.k;

    for (uint32_t mt = mt_first; mt < n; mt += mt_step) {
        // ---- stage the bases of this macro tile as Dna5 ordinals ((Dna5String) conversion)
        const uint32_t wlen = min((uint32_t)(ITEMS + k - 1), len - mt);
        __builtin_amdgcn_wave_barrier();
        for (uint32_t i = lane; i < wlen; i += 64) stage[i] = (uint8_t)seq.ord(mt + i);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();

        // ---- phase A: one k-mer per lane and tile
        uint32_t idx[J][HR];
        uint64_t kv[J];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const uint32_t p = mt + j * 64 + lane;
            uint64_t v = 0;
            if (p < n) {
                const uint8_t *b = stage + (p - mt);
                if (strand == 0) {
                    for (uint32_t i = 0; i < k; ++i) v = v * 5u + b[i];
                } else {  // k-mer of the reverse complement that covers the same window
                    for (uint32_t i = 0; i < k; ++i) v = v * 5u + rbspec::dna5_comp(b[k - 1 - i], f.comp_n);
                }
            }
            kv[j] = v;
            if constexpr (H > 0) {
#pragma unroll
                for (int h = 0; h < H; ++h)
                    idx[j][h] = rbspec::block_index(v, f.precalc[h], f.n_blocks, f.magic, f.pow2_mask);
            }
        }

        // ---- phase B: gather + count, eight steps at a time.
        // The loads of a batch of steps are issued back to back and consumed afterwards, so that a wave keeps
        // 8*H (WPL = 1) or 4*H (WPL = 2) gathers in flight.  There is no control flow around the loads: steps past
        // the end of the read and lanes without a column read block 0 of the filter (always a valid, cache-resident
        // address) and are masked out of the result -- a branch per step would make the compiler drain the
        // memory pipe after every step (3 loads in flight instead of 24).
#pragma unroll 1
        for (int blk = blk_first; blk < blk_end; ++blk) {
            {
                const int s0 = blk * 8;
                const uint32_t first = mt + (uint32_t)((s0 / SPT) * 64 + (s0 % SPT) * NG);
                if (first >= n) break;  // wave-uniform
            }
            uint64_t x[WPL][8];
            if constexpr (H > 0 && PH) {
                static_assert(WPL == 1, "the phased form serves blocks of at most 8 words");
                // what a lane keeps per lookup is the BYTE offset of its word in the table (32 bits: the engine plans this
                // form for tables far below 4 GiB), so that the gathers take the scalar-base + 32-bit-offset form -- block
                // numbers plus 64-bit addresses would not leave room for three waves per SIMD
                uint32_t bn[8][H];
                const uint32_t col_bytes = (uint32_t)((lc.lane_base - f.words) * 8);
                const uint32_t slice_shift = (ph.shift >> 31) ? (0x80000000u | ((ph.shift & 0x7FFFFFFFu) * (uint32_t)(S * 8)))
                                                              : min(31u, ph.shift + 3u + (31u - (uint32_t)__builtin_clz(S)));  // S (words per block) is a power of two here
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int s = blk * 8 + u;
                    const int j = (SPT >= 8) ? 0 : (u / SPT);
                    const int it = (s % SPT) * NG + g;
                    const uint32_t p = mt + (uint32_t)(j * 64 + it);
                    const bool ok = (p < n) && lc.colok;
                    x[0][u] = ok ? lc.valid[0] : 0ULL;
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        uint32_t b;
                        if constexpr (LG == 0) b = idx[j][h];
                        else b = shfl32(idx[j][h], it);
                        // a lookup that is not to be made gets an offset no slice ever has
                        bn[u][h] = ok ? b * (S * 8u) + col_bytes : 0xFFFFFFFFu;
                    }
                }
                phased_gather<8, H, NT, RB_GATHER_BG>(x[0], bn, f.words, slice_shift, ph);
            } else if constexpr (H > 0) {
                constexpr int HALF = (WPL == 1) ? RB_HALF1 : RB_HALF2;  // steps per load batch
#pragma unroll
                for (int half = 0; half < 8 / HALF; ++half) {
                    uint64_t ld[HALF][H][WPL];
                    bool okv[HALF];
#pragma unroll
                    for (int uu = 0; uu < HALF; ++uu) {
                        const int u = half * HALF + uu;
                        const int s = blk * 8 + u;
                        const int j = (SPT >= 8) ? 0 : (u / SPT);  // compile-time either way
                        const int it = (s % SPT) * NG + g;          // k-mer of this group within tile j
                        const uint32_t p = mt + (uint32_t)(j * 64 + it);
                        const bool ok = (p < n) && lc.colok;
                        okv[uu] = ok;
#pragma unroll
                        for (int h = 0; h < H; ++h) {
                            uint32_t b;
                            if constexpr (LG == 0) b = idx[j][h];
                            else if constexpr (LG == 6) b = readlane32(idx[j][h], it);
                            else b = shfl32(idx[j][h], it);
                            const uint64_t *src = lc.safe_base + (uint64_t)(ok ? b : 0u) * S;
                            if constexpr (WPL == 1) {
                                ld[uu][h][0] = load_word<NT>(src);
                            } else {
                                const rb_u64x2 q = load_word2<NT>(src);
                                ld[uu][h][0] = q.x;
                                ld[uu][h][1] = q.y;
                            }
                        }
                    }
                    // every gather of the batch is issued before the first one is consumed: without this fence the
                    // scheduler sinks loads next to their uses to save registers (seen in the ISA: vmcnt(3) instead
                    // of vmcnt(11..23)) and the wave is back to a few gathers in flight
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int uu = 0; uu < HALF; ++uu) {
#pragma unroll
                        for (int w = 0; w < WPL; ++w) {
                            uint64_t acc = okv[uu] ? lc.valid[w] : 0ULL;
#pragma unroll
                            for (int h = 0; h < H; ++h) acc &= ld[uu][h][w];
                            x[w][half * HALF + uu] = acc;
                        }
                    }
                }
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int s = blk * 8 + u;
                    const int j = (SPT >= 8) ? 0 : (u / SPT);
                    const int it = (s % SPT) * NG + g;
                    const uint32_t p = mt + (uint32_t)(j * 64 + it);
                    const bool ok = (p < n) && lc.colok;
                    uint64_t acc[WPL];
#pragma unroll
                    for (int w = 0; w < WPL; ++w) acc[w] = ok ? lc.valid[w] : 0ULL;
                    const uint64_t v = (LG == 0) ? kv[j] : shfl64(kv[j], it);
                    if (ok) {
                        for (uint32_t h = 0; h < f.n_hash; ++h) {
                            const uint32_t bi = rbspec::block_index(v, f.precalc[h], f.n_blocks, f.magic, f.pow2_mask);
                            const uint64_t *src = lc.lane_base + (uint64_t)bi * S;
#pragma unroll
                            for (int w = 0; w < WPL; ++w) acc[w] &= (lc.valid[w] ? load_word<NT>(src + w) : 0ULL);
                        }
                    }
#pragma unroll
                    for (int w = 0; w < WPL; ++w) x[w][u] = acc[w];
                }
            }
#pragma unroll
            for (int w = 0; w < WPL; ++w) pl[w].add8(x[w]);
        }
        if constexpr (EARLY) {
            // (with several lane groups per block every group holds the counts of ITS k-mers only: a lower bound of a lower bound, still
            // sufficient; the butterfly below is skipped when the wave leaves here)
            if (planes_max<NP, WPL>(pl, lc.valid) >= stop_at) return true;
        }
    }

    // ---- sum the partial counters of the NG groups (butterfly): every group then holds the total
    if constexpr (NG > 1) {
#pragma unroll
        for (int m = T::LPB; m < 64; m <<= 1) {
#pragma unroll
            for (int w = 0; w < WPL; ++w) pl[w].add_from_lane_xor(lane, m);
        }
    }
    return false;
}

__device__ __forceinline__ BaseSrc make_base_src(const ReadSrc &r, uint32_t item, uint32_t *len_out)
{
    const uint32_t rid = r.ids ? r.ids[item] : item;
    const uint32_t len = r.lens[item];  // effective length of this work item (chunked or whole read)
    // never beyond the declared bound: lengths above it are the caller's error (the decision kernel says so per read), and lengths
    // that are not lengths at all -- a buffer the caller has not finished writing -- must not send K1 reading gigabytes away
    *len_out = (r.max_len && len > r.max_len) ? r.max_len : len;
    BaseSrc b;
    if (r.nmask) {
        b.bytes = r.seqs + r.offsets[rid];
        b.nm = r.nmask + r.nmask_offsets[rid];
        b.first = r.base_off;
    } else {
        b.bytes = r.seqs + r.offsets[rid] + r.base_off;
        b.nm = nullptr;
        b.first = 0;
    }
    return b;
}

// throughput form: one wave per (read, column slice), both strands in sequence
#ifndef RB_WAVES_PLAIN  // waves per SIMD the plain throughput kernel is compiled for (one word per lane, ten planes)
#define RB_WAVES_PLAIN 3
#endif
// EARLY: the opt-in early-decision mode (rb_engine_set_early_decision; RB_MODE_CHECK_UNBLOCK calls that do not ask for the raw maxima).
// check_unblock (src/main/adaptive_sampling.hpp:35-113) looks at a filter's count only through "count >= threshold at r" and "count >=
// threshold at r - 0.02" (the rescan of :55-56); both are decided for good the moment some bin of the filter reaches the larger of the
// two thresholds on either strand -- the reference counts on (and counts again for the rescan).  A wave that gets there writes the
// maximum it has, a lower bound >= both thresholds, and stops: the decision kernel's predicates come out as with the full count.  Work is
// SKIPPED in this mode: its reads/s are a product figure, never a roofline figure (bench.py reports it as `c3_early`, apart).
struct EarlyCfg {
    const uint16_t *thr;  // [thr_len][nf][2] thresholds at r and r - 0.02 by read length (the decision kernel's table)
    uint32_t thr_len, nf;
    uint32_t fi[kMaxFused];  // blockIdx.y -> filter index in the table
};

template <int LG, int WPL, int NP, int H, bool NT, bool EARLY = false>
__global__ __launch_bounds__(64 * kWavesPerBlock) __attribute__((amdgpu_waves_per_eu((WPL == 1 && NP == 10 && H == 3) ? RB_WAVES_PLAIN : 3, 8))) void ibf_count_max_kernel(
    FilterSet set, ReadSrc src, uint32_t n_reads, uint32_t n_slices, uint16_t *__restrict__ out_base,
    uint32_t out_read_stride, uint32_t out_slice_stride, EarlyCfg early)
{
    __shared__ uint8_t s_stage[kWavesPerBlock][kStageBytes];
    const IbfDev &f = set.f[blockIdx.y];  // filters of equal kernel geometry may share a launch (micro-batches)
    const uint32_t col_begin = set.col_begin[blockIdx.y], col_end = set.col_end[blockIdx.y];
    uint16_t *__restrict__ out = out_base + set.out_offset[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // work item = (read, column slice), slice fastest: the waves of a workgroup gather neighbouring parts of the same
    // blocks at about the same time (DRAM page locality for wide filters)
    const uint64_t item = (uint64_t)blockIdx.x * kWavesPerBlock + wave;
    const uint32_t read = (uint32_t)(item / n_slices);
    const uint32_t slice = (uint32_t)(item - (uint64_t)read * n_slices);
    if (read >= n_reads) return;  // wave-uniform; there are no block-level barriers below

    const LaneCols<WPL> lc = make_lane_cols<LG, WPL>(f, lane, col_begin, col_end, slice);
    uint32_t len;
    const BaseSrc seq = make_base_src(src, read, &len);
    const uint32_t n = len >= f.k ? len - f.k + 1 : 0;

    uint32_t stop_at = 0xFFFFFFFFu;
    if constexpr (EARLY) {
        // the larger of the read's two thresholds for this filter (a negative int16 threshold arrives as 65 5xx: never reached, as in K2),
        // and at least one hit (a threshold of 0 is met by every read, but "classified" also needs a count above 0)
        const uint32_t tl = len < early.thr_len ? len : early.thr_len - 1;
        const uint16_t *t = early.thr + ((size_t)tl * early.nf + early.fi[blockIdx.y]) * 2;
        stop_at = max(max((uint32_t)t[0], (uint32_t)t[1]), 1u);
    }
    uint32_t best = 0;
    for (int strand = 0; strand < 2; ++strand) {
        Planes<NP> pl[WPL];
#pragma unroll
        for (int w = 0; w < WPL; ++w) pl[w].clear();
        const bool left = count_strand<LG, WPL, NP, H, NT, false, EARLY>(pl, f, lc, seq, len, n, strand, 0u, (uint32_t)TileShape<LG>::ITEMS, 0,
                                                                         TileShape<LG>::STEPS / 8, s_stage[wave], lane, PhaseCfg{0, 0, 0, 0, 0}, stop_at);
        const uint32_t m = planes_max<NP, WPL>(pl, lc.valid);
        best = m > best ? m : best;
        if (EARLY && left) break;  // wave-uniform
    }
    if (lane == 0) out[(size_t)read * out_read_stride + (size_t)slice * out_slice_stride] = (uint16_t)best;
}

// Throughput form over a MERGED table: several narrow filters of one hash geometry (same noOfBlocks, k and h -- every filter
// the reference builds with one fragment_size has them: noOfBits = BinSizeBits x 64 x binWidth, so noOfBlocks = BinSizeBits
// whatever the bin count, src/IBF/IBFBuild.cpp:404-413) hash a k-mer to the SAME block number, so their blocks can sit side
// by side in one table and ONE gather per (k-mer, hash function) serves all of them.  The path is bound by requests, not
// bytes: 1 deplete + 3 target filters of 2 + 1 + 1 + 1 words become one 64-byte gather instead of four lookups.  Counting is
// unchanged (one word column per lane, bit-sliced planes); the max over bins is taken per filter, over the lanes that hold
// that filter's columns, and lane g of the wave carries filter g's result across the strands.
template <int LG, int NP, bool NT>
__global__ __launch_bounds__(64 * kWavesPerBlock) void ibf_count_max_merged_kernel(IbfDev f, MergeMap map, ReadSrc src, uint32_t n_reads,
                                                                                   uint16_t *__restrict__ out, uint32_t out_read_stride)
{
    __shared__ uint8_t s_stage[kWavesPerBlock][kStageBytes];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t read = blockIdx.x * kWavesPerBlock + wave;
    if (read >= n_reads) return;  // wave-uniform; no block-level barriers below
    LaneCols<1> lc = make_lane_cols<LG, 1>(f, lane, 0u, f.bin_width, 0u);
    // which bins of this lane's word column exist at all: the union of the members' bit ranges (word-aligned or packed)
    const uint32_t col = (uint32_t)lane & ((1u << LG) - 1u);
    uint64_t valid = 0;
    for (uint32_t g = 0; g < map.n; ++g) valid |= member_mask(map.bit_begin[g], map.bit_end[g], col);
    if (col >= map.width) valid = 0;
    lc.valid[0] = valid;
    lc.colok = valid != 0;
    lc.safe_base = lc.colok ? lc.lane_base : f.words;
    uint32_t len;
    const BaseSrc seq = make_base_src(src, read, &len);
    const uint32_t n = len >= f.k ? len - f.k + 1 : 0;
    uint32_t best = 0;  // lane g: filter g
    for (int strand = 0; strand < 2; ++strand) {
        Planes<NP> pl[1];
        pl[0].clear();
        count_strand<LG, 1, NP, 3, NT>(pl, f, lc, seq, len, n, strand, 0u, (uint32_t)TileShape<LG>::ITEMS, 0, TileShape<LG>::STEPS / 8,
                                       s_stage[wave], lane);
        for (uint32_t g = 0; g < map.n; ++g) {
            const uint64_t mine[1] = {col < map.width ? member_mask(map.bit_begin[g], map.bit_end[g], col) : 0ULL};
            const uint32_t m = planes_max<NP, 1>(pl, mine);
            if ((uint32_t)lane == g) best = m > best ? m : best;
        }
    }
    if ((uint32_t)lane < map.n) out[(size_t)read * out_read_stride + map.out_offset[lane]] = (uint16_t)best;
}

// merged table: block b of a filter (width words at stride s_src, n_bins bins) -> bits [dst_bit, dst_bit + n_bins) of block b of dst.
// One thread per (block, source word): the word is masked to the bins that exist and ORed into the one or two destination words it
// lands in (atomically: members that share a destination word are merged by separate launches on one stream, but two source words
// of ONE member can meet in a destination word when dst_bit is not a multiple of 64).
__global__ void merge_bits_kernel(const uint64_t *__restrict__ src, uint32_t s_src, uint32_t width, uint32_t n_bins, uint64_t *__restrict__ dst,
                                  uint32_t s_dst, uint32_t dst_bit, uint64_t n_blocks)
{
    const uint64_t total = n_blocks * width;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const uint64_t b = i / width;
        const uint32_t c = (uint32_t)(i - b * width);
        uint64_t v = src[b * s_src + c] & member_mask(0u, n_bins, c);
        if (v == 0) continue;
        const uint32_t at = dst_bit + c * 64u;
        const uint32_t w = at >> 6, sh = at & 63u;
        atomicOr(reinterpret_cast<unsigned long long *>(dst + b * s_dst + w), (unsigned long long)(v << sh));
        if (sh && (v >> (64u - sh))) atomicOr(reinterpret_cast<unsigned long long *>(dst + b * s_dst + w + 1), (unsigned long long)(v >> (64u - sh)));
    }
}

// throughput form with clock-phased gathers (see count_strand): narrow filters of a few L2 sizes, one column slice
// (three waves per SIMD: the lookups a CU holds in registers are what a window has to work with).  SHORT != 0: the engine
// knows that no read of the batch has more k-mers than one of the both-strands shapes below takes, and only that path is
// compiled in:
//   SHORT 1  <= 256 k-mers (the reference's default 250 bp chunk): ONE round of 4 tiles per strand -- 93 VGPRs, five waves
//            per SIMD for one-word blocks;
//   SHORT 3  <= 384 k-mers (360 bp reads, the length the reference recommends): ONE round of 6 tiles per strand.  Two rounds
//            of 4 tiles left the second one 36 % dense (92 of 256 k-mers per strand) while every round costs a full turn
//            of windows; 6 tiles are 91 % dense and the per-bin sums of six words still fit three bit planes;
//   SHORT 2  <= 512 k-mers: two rounds of 4 tiles (the general build, SHORT 0, takes this path too and falls back to
//            the per-strand tiles of count_strand for longer reads).
// bins of a word column as a mask (NarrowMerge::col_bits)
__device__ __forceinline__ uint64_t col_bits_mask(uint32_t bits) { return bits >= 64 ? ~0ULL : ((1ULL << bits) - 1); }

// The end of the one-lane-per-block rounds: lane b holds, per word column c, the larger of the two strands' counts of bin 64 c + b;
// every member of the (possibly merged) table gets the maximum over ITS bins -- the bit range [bit_begin, bit_end) of the block,
// word-aligned or packed -- and the wave.
template <int NC>
__device__ __forceinline__ void write_member_maxima(const uint32_t (&colmax)[NC], const NarrowMerge &nm, int lane, uint16_t *out)
{
    for (uint32_t g = 0; g < nm.n; ++g) {
        uint32_t m = 0;
        // bin 64 c + lane lies in [begin, end)  <=>  (lane - begin) + 64 c < end - begin in unsigned arithmetic: one subtraction per member
        // (the two-sided compare per column cost the two-word 250 bp build its 72nd register: 73 -> six instead of seven waves per SIMD,
        // 9.3 -> 10.8 ms per 1 M reads on a 20 MB table)
        const uint32_t rel = (uint32_t)lane - nm.bit_begin[g], span = nm.bit_end[g] - nm.bit_begin[g];
#pragma unroll
        for (int c = 0; c < NC; ++c) m = ((rel + (uint32_t)(c * 64)) < span && colmax[c] > m) ? colmax[c] : m;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const uint32_t o = shfl32(m, lane ^ sft);
            m = o > m ? o : m;
        }
        if (lane == 0) out[nm.out_offset[g]] = (uint16_t)m;
    }
}

// k-mer values from staged TRIPLES of bases, for k <= 13 (the reference's default; 5^13 < 2^31, so a value fits 32 bits).  Beside the
// staged Dna5 ordinals ord[i] a read's staging area holds f3[i] = 25 ord[i] + 5 ord[i+1] + ord[i+2] and r3[i] = the same of the
// complemented bases in reverse order; the forward value of the window at p is then k/3 steps `v = 125 v + f3[..]` (v_mad_u32_u24: every
// step starts below 5^10 < 2^24, the 24-bit multiply is exact) plus k%3 single bases, the reverse-complement value the single bases from
// the right end first and then the triples from the right -- ~20 VALU instructions per window position and both strands where thirteen
// 64-bit multiply-adds per strand took ~180.  The narrow kernels are three quarters VALU-busy (profiles/r05/pmc_summary.csv, targets3:
// SQ_INSTS_VALU 4 207 per read x 4 cycles over 1 024 SIMDs = 7.3 of 9.5 ms), and the hashing was the largest part of it.
constexpr uint32_t kTripleMaxK = 13;
__device__ __forceinline__ void stage_triples(const uint8_t *ord, uint8_t *f3, uint8_t *r3, uint32_t len, uint32_t comp_n, int lane)
{
    for (uint32_t i = lane; i + 2 < len; i += 64) {
        const uint32_t a = ord[i], b = ord[i + 1], c = ord[i + 2];
        f3[i] = (uint8_t)(a * 25u + b * 5u + c);
        r3[i] = (uint8_t)(rbspec::dna5_comp(c, comp_n) * 25u + rbspec::dna5_comp(b, comp_n) * 5u + rbspec::dna5_comp(a, comp_n));
    }
}
// values of the k-mer at p and of its reverse complement; small_k: the triples are staged (k <= kTripleMaxK), q3 = k / 3, s1 = k % 3
__device__ __forceinline__ void kmer_values_both(const uint8_t *ord, const uint8_t *f3, const uint8_t *r3, uint32_t p, uint32_t k, uint32_t q3,
                                                 uint32_t s1, uint32_t comp_n, bool small_k, uint64_t &vf, uint64_t &vr)
{
    if (small_k) {
        uint32_t a = 0, b = 0;
        for (uint32_t t = 0; t < q3; ++t) a = __umul24(a, 125u) + f3[p + 3u * t];
        for (uint32_t t = 0; t < s1; ++t) a = a * 5u + ord[p + 3u * q3 + t];
        for (uint32_t t = 0; t < s1; ++t) b = b * 5u + rbspec::dna5_comp(ord[p + k - 1u - t], comp_n);
        for (uint32_t t = q3; t-- > 0;) b = __umul24(b, 125u) + r3[p + 3u * t];
        vf = a;
        vr = b;
    } else {
        const uint8_t *bs = ord + p;
        vf = 0;
        vr = 0;
        for (uint32_t i = 0; i < k; ++i) vf = vf * 5u + bs[i];
        for (uint32_t i = 0; i < k; ++i) vr = vr * 5u + rbspec::dna5_comp(bs[k - 1 - i], comp_n);
    }
}
// the same for ONE strand (rc: the reverse complement's k-mer over the window at p)
__device__ __forceinline__ uint64_t kmer_value_one(const uint8_t *ord, const uint8_t *f3, const uint8_t *r3, uint32_t p, uint32_t k, uint32_t q3,
                                                   uint32_t s1, uint32_t comp_n, bool small_k, bool rc)
{
    if (small_k) {
        uint32_t a = 0;
        if (!rc) {
            for (uint32_t t = 0; t < q3; ++t) a = __umul24(a, 125u) + f3[p + 3u * t];
            for (uint32_t t = 0; t < s1; ++t) a = a * 5u + ord[p + 3u * q3 + t];
        } else {
            for (uint32_t t = 0; t < s1; ++t) a = a * 5u + rbspec::dna5_comp(ord[p + k - 1u - t], comp_n);
            for (uint32_t t = q3; t-- > 0;) a = __umul24(a, 125u) + r3[p + 3u * t];
        }
        return a;
    }
    const uint8_t *bs = ord + p;
    uint64_t v = 0;
    if (!rc) {
        for (uint32_t i = 0; i < k; ++i) v = v * 5u + bs[i];
    } else {
        for (uint32_t i = 0; i < k; ++i) v = v * 5u + rbspec::dna5_comp(bs[k - 1 - i], comp_n);
    }
    return v;
}
// staging area of the one-lane-per-block register builds: the ordinals of a read of up to 512 + k - 1 bases, then the two triple arrays
constexpr int kTripleStride = 64 * kMaxTiles + rbspec::kMaxKmer;  // = kStageBytes: bytes per staged array
constexpr int kStageBytes3 = 3 * kTripleStride;

// Waves per SIMD each build of the phased kernel is compiled for (amdgpu_waves_per_eu minimum = the register budget the compiler
// works against).  Left to itself it stops at the first allocation that fits its default target; told to aim higher it finds
// 56 instead of 73 registers for the one-word 250 bp build (8 waves), 76 instead of 93 for the one-word 360 bp build (6 waves), 72
// instead of 86 for the two-word 250 bp build (7 waves) -- all without scratch (the two-word 360 bp build spills 96 bytes at five
// waves and stays at four).  More reads per cycle: 5-15 % less time per read at the (longer) windows that go with it
// (profiles/r03/slice_size.txt, session 54).
#ifndef RB_WAVES_0_1
#define RB_WAVES_0_1 7
#endif
#ifndef RB_WAVES_0_3
#define RB_WAVES_0_3 6
#endif
#ifndef RB_WAVES_1_1
#define RB_WAVES_1_1 6
#endif
#ifndef RB_WAVES_1_3
#define RB_WAVES_1_3 3
#endif
#ifndef RB_TILES_ROUNDS  // tiles per strand and round of the one-word build for reads of 385-512 k-mers: three (63 registers, eight waves per
                        // SIMD; two rounds of four tiles took 111 registers, four waves: session 61, 7-13 % slower); two-word blocks keep four
#define RB_TILES_ROUNDS 3
#endif
#ifndef RB_WIDE3_ROUNDS  // a three-word build for the rounds of three tiles as well (94 registers at five waves per SIMD; session 59: 3-10 % faster)
#define RB_WIDE3_ROUNDS 1
#endif
#ifndef RB_WAVES_0_2
#define RB_WAVES_0_2 7
#endif
#ifndef RB_WAVES_1_2
#define RB_WAVES_1_2 3
#endif
#ifndef RB_WAVES_2_2
#define RB_WAVES_2_2 3
#endif
#ifndef RB_WAVES_2_2_NW3
#define RB_WAVES_2_2_NW3 5
#endif
#ifndef RB_WAVES_GEN
#define RB_WAVES_GEN 3
#endif
constexpr int phased_min_waves(int lg, int shrt, int nw)
{
    if (lg == 0) return shrt == 1 ? RB_WAVES_0_1 : shrt == 3 ? RB_WAVES_0_3 : shrt ? RB_WAVES_0_2 : RB_WAVES_GEN;
    if (lg == 1) return shrt == 1 ? RB_WAVES_1_1 : shrt == 3 ? RB_WAVES_1_3 : shrt ? RB_WAVES_1_2 : RB_WAVES_GEN;
    if (lg == 2 && shrt == 1) return nw == 3 ? RB_WIDE_WAVES : 4;
    if (lg == 2) return nw == 3 ? RB_WAVES_2_2_NW3 : RB_WAVES_2_2;
    return 3;
}

template <int LG, int NP, int SHORT, int NW = 4>  // NW: words per block the one-lane build for stride-4 blocks holds (3: no fourth column)
__global__ __launch_bounds__(64 * kWavesPerBlock) __attribute__((amdgpu_waves_per_eu(phased_min_waves(LG, SHORT, NW), 8))) void ibf_count_max_phased_kernel(
    IbfDev f, uint32_t col_begin, uint32_t col_end, ReadSrc src, uint32_t n_reads, PhaseCfg ph, uint16_t *__restrict__ out,
    uint32_t out_read_stride, NarrowMerge nm)
{
    __shared__ uint8_t s_stage[kWavesPerBlock][kStageBytes3];  // (ordinals + the two triple arrays of stage_triples)
    if (ph.xcd_skew) ph.skew = xcc_id();
    if (ph.tskew) ph.tskew *= xcc_id();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t read = blockIdx.x * kWavesPerBlock + wave;
    if (read >= n_reads) return;  // wave-uniform; no block-level barriers below
    const LaneCols<1> lc = make_lane_cols<LG, 1>(f, lane, col_begin, col_end, 0);
    uint32_t len;
    const BaseSrc seq = make_base_src(src, read, &len);
    const uint32_t n = len >= f.k ? len - f.k + 1 : 0;
    uint32_t best = 0;
    constexpr int T = SHORT == 3 ? 6 : (SHORT == 2 && LG == 0) ? RB_TILES_ROUNDS : 4;  // 64-k-mer tiles per strand and round
    constexpr uint32_t kRound = 64u * T;                       // k-mers per strand and round
    constexpr bool kOneRound = SHORT == 1 || SHORT == 3;       // no loop state (registers)
    if constexpr (LG == 0) {
        // One-word blocks, reads of up to 512 k-mers: a 512-k-mer macro tile per strand would leave more than half of the 24
        // lookups a lane can keep in flight unused on the reference's default 250 bp chunk (238 k-mers) -- and the windows
        // of the phased form live on lookups held in registers.  So both strands share a macro tile: tiles 0..T-1 are
        // forward k-mers, tiles T..2T-1 the k-mers of the reverse complement over the same windows (separate counts, as in
        // the reference).
        constexpr uint32_t kBothMax = SHORT == 1 ? 256u : SHORT == 3 ? 384u : 512u;  // the general build takes two rounds as well
        if (n <= kBothMax) {  // len <= 512 + k - 1 <= kStageBytes: the whole read is staged once
            uint8_t *stage = s_stage[wave];
            for (uint32_t i = lane; i < len; i += 64) stage[i] = (uint8_t)seq.ord(i);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const uint32_t k = f.k;
            // (the build with loop state, rounds of three tiles, keeps the 64-bit Horner: the triples' extra state costs it its eighth wave -- 72
            // registers and a spill)
            const bool small_k = !(SHORT == 2) && k <= kTripleMaxK;
            const uint32_t q3 = k / 3u, s1 = k - 3u * q3;
            if (small_k) {
                stage_triples(stage, stage + kTripleStride, stage + 2 * kTripleStride, len, f.comp_n, lane);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
            const uint32_t col_bytes = (uint32_t)((lc.lane_base - f.words) * 8);
            const uint32_t slice_shift = (ph.shift >> 31) ? (0x80000000u | ((ph.shift & 0x7FFFFFFFu) * (f.stride * 8u)))
                                                          : min(31u, ph.shift + 3u + (31u - (uint32_t)__builtin_clz(f.stride)));
            uint32_t cf = 0, cr = 0;  // lane b: count of bin b, forward / reverse complement
            // the counts of a round are summed across the wave at once, so no counter planes are carried
#pragma unroll 1
            for (uint32_t base = 0; base < (kOneRound ? 1u : n); base += kRound) {
            uint32_t bn[2 * T][3];
            uint64_t x[2 * T];
#pragma unroll
            for (int j = 0; j < 2 * T; ++j) {  // slots 0 .. T-1: forward k-mers; T .. 2T-1: those of the reverse complement over the same windows
                const uint32_t p = base + (uint32_t)((j % T) * 64 + lane);
                const bool ok = (p < n) && lc.colok;
                uint64_t v = 0;
                if (ok) v = kmer_value_one(stage, stage + kTripleStride, stage + 2 * kTripleStride, p, k, q3, s1, f.comp_n, small_k, j >= T);
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    const uint32_t blk = rbspec::block_index(v, f.precalc[h], f.n_blocks, f.magic, f.pow2_mask);
                    bn[j][h] = ok ? blk * (f.stride * 8u) + col_bytes : 0xFFFFFFFFu;
                }
                if constexpr (T == 6) __builtin_amdgcn_sched_barrier(0);  // one k-mer's hash chains at a time (registers)
            }
#pragma unroll
            for (int j = 0; j < 2 * T; ++j) x[j] = bn[j][0] != 0xFFFFFFFFu ? lc.valid[0] : 0ULL;  // after the hashing: 16 registers less there
            // (six tiles per strand: the 36 gathers of a window go out in two batches of 18)
            phased_gather<2 * T, 3, false, T == 6 ? RB_GATHER_B3 : (SHORT == 1 ? RB_GATHER_B1 : (T == 4 ? 8 : 2))>(x, bn, f.words, slice_shift, ph);
            cf += wave_bin_counts<T>(x, lane);
            cr += wave_bin_counts<T>(x + T, lane);
            }
            uint32_t m = cf > cr ? cf : cr;  // bins beyond noOfBins count 0: the gathered words were masked with lc.valid
#pragma unroll
            for (int sft = 1; sft < 64; sft <<= 1) {
                const uint32_t o = shfl32(m, lane ^ sft);
                m = o > m ? o : m;
            }
            if (lane == 0) out[(size_t)read * out_read_stride] = (uint16_t)m;
            return;
        }
    }
    if constexpr (LG == 1) {
        // Two-word blocks (65-128 bins), reads of up to 512 k-mers: the same both-strands tile with ONE lane per block and
        // 16-byte gathers -- twice the lookups a wave holds per round of windows compared with two lanes per block, and
        // a 20 MB table has to cross the fabric once per round whatever a wave asks of it.
        // (two rounds only in the build that has nothing else in it: next to the per-strand path the loop state spills)
        constexpr uint32_t kBothMax = SHORT == 2 ? 512u : SHORT == 3 ? 384u : 256u;
        if (n <= kBothMax && col_begin == 0 && col_end == 2 && f.stride == 2) {
            uint8_t *stage = s_stage[wave];
            for (uint32_t i = lane; i < len; i += 64) stage[i] = (uint8_t)seq.ord(i);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const uint32_t k = f.k;
            const bool small_k = !(SHORT == 2) && k <= kTripleMaxK;  // (not in the builds with loop state: registers)
            const uint32_t q3 = k / 3u, s1 = k - 3u * q3;
            if (small_k) {
                stage_triples(stage, stage + kTripleStride, stage + 2 * kTripleStride, len, f.comp_n, lane);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
            const uint64_t valid0 = col_bits_mask(nm.col_bits[0]), valid1 = col_bits_mask(nm.col_bits[1]);
            const uint32_t slice_shift = (ph.shift >> 31) ? (0x80000000u | ((ph.shift & 0x7FFFFFFFu) * 16u)) : min(31u, ph.shift + 4u);
            uint32_t cf = 0, cr = 0;  // lane b: counts of bins b (low half) and 64 + b (high half), forward / reverse complement
#pragma unroll 1
            for (uint32_t base = 0; base < (SHORT == 2 ? n : 1u); base += kRound) {
            uint32_t bn[2 * T][3];
            uint64_t x0[2 * T], x1[2 * T];
#pragma unroll
            for (int j = 0; j < 2 * T; ++j) {
                const uint32_t p = base + (uint32_t)((j % T) * 64 + lane);
                const bool ok = p < n;
                uint64_t v = 0;
                if (ok) v = kmer_value_one(stage, stage + kTripleStride, stage + 2 * kTripleStride, p, k, q3, s1, f.comp_n, small_k, j >= T);
#pragma unroll
                for (int h = 0; h < 3; ++h) {
                    const uint32_t blk = rbspec::block_index(v, f.precalc[h], f.n_blocks, f.magic, f.pow2_mask);
                    bn[j][h] = ok ? blk * 16u : 0xFFFFFFFFu;
                }
                // six tiles: one k-mer's hash chains at a time (interleaved, the twelve of them spill 60 registers)
                if constexpr (T == 6) __builtin_amdgcn_sched_barrier(0);
            }
            // the AND accumulators start as "every existing bin" for the k-mers that exist (set up after the hashing: 32
            // registers less while the hash chains are in flight)
#pragma unroll
            for (int j = 0; j < 2 * T; ++j) {
                const bool ok = bn[j][0] != 0xFFFFFFFFu;
                x0[j] = ok ? valid0 : 0ULL;
                x1[j] = ok ? valid1 : 0ULL;
            }
            // (the builds with loop state or six tiles send two k-mers' gathers at a time: with four the state does not fit
            // three waves per SIMD)
            phased_gather_x2<2 * T, 3, SHORT == 1 ? RB_GATHER_KB1 : (SHORT == 3 ? RB_GATHER_KB3 : 2)>(x0, x1, bn, f.words, slice_shift, ph);
            cf += wave_bin_counts<T>(x0, lane) | (wave_bin_counts<T>(x1, lane) << 16);
            cr += wave_bin_counts<T>(x0 + T, lane) | (wave_bin_counts<T>(x1 + T, lane) << 16);
            }
            const uint32_t colmax[2] = {max(cf & 0xFFFFu, cr & 0xFFFFu), max(cf >> 16, cr >> 16)};  // at most 512 each: no carry between halves
            write_member_maxima<2>(colmax, nm, lane, out + (size_t)read * out_read_stride);
            return;
        }
    }
    if constexpr (LG == 2 && (SHORT == 2 || SHORT == 1)) {
        // Three- and four-word blocks (129-256 bins, stride 4 words), reads of up to 512 k-mers: ONE lane per block with two
        // 16-byte gathers, both strands in rounds of two 64-k-mer tiles each (a lane's AND accumulators are 4 words x 4 k-mers =
        // 32 registers; four tiles per strand would take 64 and leave three waves per SIMD)
        // SHORT 1: one round of four tiles per strand (<= 256 k-mers; 116 registers, four waves per SIMD); otherwise rounds of three
        // tiles (113 registers, four waves; 360 bp reads take two rounds, 91 % dense -- rounds of two tiles: three rounds, 11-17 %
        // instead of 30-50 % over the plain kernel; four tiles with the loop state: 163 registers, three waves, slower still)
        constexpr int T2 = SHORT == 1 ? 4 : RB_WIDE_TILES;
        if (n <= (SHORT == 1 ? 256u : 512u) && col_begin == 0 && (col_end == (uint32_t)NW || (NW == 4 && col_end == 3)) && f.stride == 4) {
            uint8_t *stage = s_stage[wave];
            for (uint32_t i = lane; i < len; i += 64) stage[i] = (uint8_t)seq.ord(i);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            __builtin_amdgcn_wave_barrier();
            const uint32_t k = f.k;
            const bool small_k = SHORT == 1 && k <= kTripleMaxK;  // (not in the builds with loop state: registers)
            const uint32_t q3 = k / 3u, s1 = k - 3u * q3;
            if (small_k) {
                stage_triples(stage, stage + kTripleStride, stage + 2 * kTripleStride, len, f.comp_n, lane);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                __builtin_amdgcn_wave_barrier();
            }
            const uint64_t valid0 = col_bits_mask(nm.col_bits[0]), valid1 = col_bits_mask(nm.col_bits[1]);
            const uint64_t valid2 = col_bits_mask(nm.col_bits[2]), valid3 = col_bits_mask(nm.col_bits[3]);
            const uint32_t slice_shift = (ph.shift >> 31) ? (0x80000000u | ((ph.shift & 0x7FFFFFFFu) * 32u)) : min(31u, ph.shift + 5u);
            uint32_t c01f = 0, c23f = 0, c01r = 0, c23r = 0;  // lane b: counts of bins b | 64 + b << 16, and 128 + b | 192 + b << 16
#pragma unroll 1
            for (uint32_t base = 0; base < (SHORT == 1 ? 1u : n); base += 64u * T2) {
                uint32_t bn[2 * T2][3];
                uint64_t x0[2 * T2], x1[2 * T2], x2[2 * T2], x3[2 * T2];
#pragma unroll
                for (int j = 0; j < 2 * T2; ++j) {
                    const uint32_t p = base + (uint32_t)((j % T2) * 64 + lane);
                    const bool ok = p < n;
                    uint64_t v = 0;
                    if (ok) v = kmer_value_one(stage, stage + kTripleStride, stage + 2 * kTripleStride, p, k, q3, s1, f.comp_n, small_k, j >= T2);
#pragma unroll
                    for (int h = 0; h < 3; ++h) {
                        const uint32_t blk = rbspec::block_index(v, f.precalc[h], f.n_blocks, f.magic, f.pow2_mask);
                        bn[j][h] = ok ? blk * 32u : 0xFFFFFFFFu;
                    }
                }
#pragma unroll
                for (int j = 0; j < 2 * T2; ++j) {
                    const bool ok = bn[j][0] != 0xFFFFFFFFu;
                    x0[j] = ok ? valid0 : 0ULL;
                    x1[j] = ok ? valid1 : 0ULL;
                    x2[j] = ok ? valid2 : 0ULL;
                    x3[j] = (ok && NW == 4) ? valid3 : 0ULL;
                }
                phased_gather_x4<2 * T2, 3, 1, NW>(x0, x1, x2, x3, bn, f.words, slice_shift, ph);
                c01f += wave_bin_counts<T2>(x0, lane) | (wave_bin_counts<T2>(x1, lane) << 16);
                c01r += wave_bin_counts<T2>(x0 + T2, lane) | (wave_bin_counts<T2>(x1 + T2, lane) << 16);
                if constexpr (NW == 4) {
                    c23f += wave_bin_counts<T2>(x2, lane) | (wave_bin_counts<T2>(x3, lane) << 16);
                    c23r += wave_bin_counts<T2>(x2 + T2, lane) | (wave_bin_counts<T2>(x3 + T2, lane) << 16);
                } else {
                    c23f += wave_bin_counts<T2>(x2, lane);
                    c23r += wave_bin_counts<T2>(x2 + T2, lane);
                }
            }
            const uint32_t colmax[4] = {max(c01f & 0xFFFFu, c01r & 0xFFFFu), max(c01f >> 16, c01r >> 16),
                                        max(c23f & 0xFFFFu, c23r & 0xFFFFu), max(c23f >> 16, c23r >> 16)};  // at most 512 each: no carry
            write_member_maxima<4>(colmax, nm, lane, out + (size_t)read * out_read_stride);
            return;
        }
    }
    if constexpr (!SHORT) {
        for (int strand = 0; strand < 2; ++strand) {
            Planes<NP> pl[1];
            pl[0].clear();
            count_strand<LG, 1, NP, 3, false, true>(pl, f, lc, seq, len, n, strand, 0u, (uint32_t)TileShape<LG>::ITEMS, 0,
                                                    TileShape<LG>::STEPS / 8, s_stage[wave], lane, ph);
            const uint32_t m = planes_max<NP, 1>(pl, lc.valid);
            best = m > best ? m : best;
        }
    }
    // SHORT: a read with more k-mers than promised (256 / 384 / 512) writes 0 here; the decision kernel turns a length above the
    // declared max_len into RB_ERR_INVALID_ARG, so the value is never used
    if (lane == 0) {
        out[(size_t)read * out_read_stride + nm.out_offset[0]] = (uint16_t)best;
        for (uint32_t g = 1; g < nm.n; ++g) out[(size_t)read * out_read_stride + nm.out_offset[g]] = 0;  // (merged tables: SHORT builds only)
    }
}

// ---------------------------------------------------------------------------------------------
// SEVERAL READS PER WAVE through one pass of the clock-phased windows (round 6; two-word blocks, reads of up to 256 k-mers: the
// build ibf_count_max_phased_kernel<1,10,1,4> serves with one read per wave).  Why: a pass of the chip over all slices reloads the
// table once per XCD, whatever the number of reads that ride along, so the fabric's share of a read's time falls with the reads a CU
// holds per pass -- and those are bounded by on-chip state: per read 8 slots x 64 lanes x (16 bytes of AND accumulator + three
// offsets).  The shipped build keeps both in registers (72 VGPRs, seven waves per SIMD: 28 reads per CU).  Here the accumulators of R
// reads stay in registers and the offsets live in LDS, three block numbers of 21 bits packed into one 64-bit word per (read, slot,
// lane) -- 4 KiB per read, read back once per window (ds_read_b64, lane-consecutive: conflict-free) and unpacked with four VALU
// operations; recomputing them per window from the staged bases instead (3 x 64-bit multiply + xor-shift + Barrett per k-mer) would
// cost more VALU time than a window has.  R = 2 at five waves per SIMD: 40 reads per CU, 160 KiB of LDS -- all of it, which is why
// a read's bases are staged inside its own last slot's region (the slot is hashed into registers before it is overwritten) and a
// workgroup is ONE wave (LDS is granted per workgroup: 8 KiB units pack where 32 KiB units would not).
// Packed block numbers: tables of at most 2^21 - 1 blocks (32 MiB of two-word blocks); 0x1FFFFF = no lookup, out of range of every
// slice because a slice's descriptor ends with the table.  Same windows, same slices, same counting as the one-read build; results
// are identical by construction (the same AND of the same words).
constexpr uint32_t kPackBits = 21;
constexpr uint32_t kPackMask = (1u << kPackBits) - 1u;
// One-word blocks (8 bytes each: a table of 2^21 of them is only 16 MiB) pack 22-bit block numbers: the first two whole, the low 20 bits
// of the third, and its top two bits in a per-lane SPILL word held in a register (two bits per slot, at most twelve slots) -- tables of
// up to 2^22 - 2 blocks (32 MiB), which takes in the 64-bin filter of the reference's own test data (2.47 M blocks).  Two VALU operations
// more per slot and window; 0x3FFFFF = no lookup.
constexpr uint32_t kPackBits1 = 22;
constexpr uint32_t kPackMask1 = (1u << kPackBits1) - 1u;
#ifndef RB_MULTI_WAVES
#define RB_MULTI_WAVES 5
#endif
#ifndef RB_MULTI_WAVES_W4  // the four-word build of four tiles (64 accumulator registers)
#define RB_MULTI_WAVES_W4 5
#endif

// three packed block numbers of one k-mer (PB = kPackBits each, or kPackBits1 with the third one's top two bits returned in hi2)
template <int PB>
__device__ __forceinline__ uint64_t pack_lookups(uint64_t v, const IbfDev &f, uint32_t &hi2)
{
    const uint64_t b0 = rbspec::block_index(v, f.precalc[0], f.n_blocks, f.magic, f.pow2_mask);
    const uint64_t b1 = rbspec::block_index(v, f.precalc[1], f.n_blocks, f.magic, f.pow2_mask);
    const uint64_t b2 = rbspec::block_index(v, f.precalc[2], f.n_blocks, f.magic, f.pow2_mask);
    hi2 = PB == (int)kPackBits ? 0u : (uint32_t)(b2 >> 20);
    return b0 | (b1 << PB) | (b2 << (2 * PB));  // (22 bits: the shift drops the third number's top two bits)
}

// Phase A of the multi-read build for ONE read: its k-mers' packed block numbers go to slots[j * 64 + lane], j = 0..3 the forward
// k-mers at p = 64 j + lane, j = 4..7 the k-mers of the reverse complement over the same windows (~0: no such k-mer).  Returns the
// number of k-mers (0: no such read, a read shorter than k, or one longer than the build was promised -- it counts 0 like the
// one-read build).  The read's bases are staged INSIDE the regions of its own slots 6 and 7 (1 KiB; the wave has no other LDS), so
// those two slots are hashed into registers and stored after every lane is through with the staged bases.
// k <= 13 (the reference's default; 5^13 < 2^31): the k-mer values are 32-bit and come from staged TRIPLES of bases -- f3[i] =
// 25 b[i] + 5 b[i+1] + b[i+2], r3[i] the same of the complemented bases in reverse order -- so a value is four v_mad_u32_u24 by 125
// and one step by 5 instead of thirteen 64-bit multiply-adds per strand (~20 instead of ~180 VALU instructions per window position
// and both strands).  The count kernels of the narrow filters are three quarters VALU-busy (profiles/r05/pmc_summary.csv, targets3:
// SQ_INSTS_VALU 4 207 per read, x 4 cycles over 1024 SIMDs = 7.3 of the kernel's 9.5 ms).  Longer k: 64-bit Horner over the bases.
// T: 64-k-mer tiles per strand (4: reads of up to 256 k-mers; 6: up to 384, the 360 bp prefixes the reference recommends); 2 T slots
// per lane.  The staging area is the region of the last NST slots (T 4: 1 KiB for up to 268 bases and their two triple arrays; T 6:
// 1.5 KiB for up to 396), which are reverse-strand slots of the last NST tiles.
template <int T>
struct MultiShape {
    static constexpr int S = 2 * T;                       // slots per lane
    static constexpr int NST = T == 4 ? 2 : 3;            // slot regions that hold the staged bases
    static constexpr uint32_t kMaxKmers = 64u * T;
    static constexpr uint32_t kArr = T == 4 ? 272u : 400u;  // bytes per staged array: ord, then f3, then r3
    static_assert(3 * kArr <= NST * 512u, "the staged arrays fit the slot regions they borrow");
};

template <int T, int PB>
__device__ __forceinline__ void multi_hash_staged(uint64_t *slots, const IbfDev &f, uint32_t len, uint32_t n, int lane, uint32_t &spill);
template <int T, int PB = (int)kPackBits>
__device__ __forceinline__ uint32_t multi_hash_read(uint64_t *slots, const IbfDev &f, const ReadSrc &src, uint32_t rid, uint32_t n_reads, int lane, uint32_t &spill)
{
    using M = MultiShape<T>;
    const uint32_t k = f.k;
    uint32_t len = 0, n = 0;
    uint8_t *ord = reinterpret_cast<uint8_t *>(slots + (M::S - M::NST) * 64);
    if (rid < n_reads) {
        const BaseSrc seq = make_base_src(src, rid, &len);
        n = len >= k ? len - k + 1 : 0;
        if (n > M::kMaxKmers) n = 0;
        if (n)
            for (uint32_t i = lane; i < len; i += 64) ord[i] = (uint8_t)seq.ord(i);
    }
    multi_hash_staged<T, PB>(slots, f, len, n, lane, spill);
    return n;
}

// the second half: the read's Dna5 ordinals are staged at the start of the staging area (by this wave; no fence taken yet)
// (spill: PB = 22 only -- bits 2 u, 2 u + 1 = the top two bits of slot u's third block number; all ones where a slot has no k-mer)
template <int T, int PB>
__device__ __forceinline__ void multi_hash_staged(uint64_t *slots, const IbfDev &f, uint32_t len, uint32_t n, int lane, uint32_t &spill)
{
    using M = MultiShape<T>;
    const uint32_t k = f.k;
    uint8_t *ord = reinterpret_cast<uint8_t *>(slots + (M::S - M::NST) * 64);
    uint8_t *f3 = ord + M::kArr, *r3 = ord + 2 * M::kArr;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const bool small_k = k <= kTripleMaxK;
    if (small_k && n) {
        stage_triples(ord, f3, r3, len, f.comp_n, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    const uint32_t q3 = k / 3u, s1 = k - 3u * q3;  // k = 3 q3 + s1
    uint64_t held_r[M::NST];  // the last NST tiles: their reverse slots hold the staged bases
#pragma unroll
    for (int i = 0; i < M::NST; ++i) held_r[i] = ~0ULL;
#pragma unroll
    for (int j = 0; j < T; ++j) {
        const uint32_t p = (uint32_t)(j * 64 + lane);
        uint64_t pf = ~0ULL, pr = ~0ULL;
        if (p < n) {
            uint64_t vf, vr;
            uint32_t hf, hr;
            kmer_values_both(ord, f3, r3, p, k, q3, s1, f.comp_n, small_k, vf, vr);
            pf = pack_lookups<PB>(vf, f, hf);
            pr = pack_lookups<PB>(vr, f, hr);
            if constexpr (PB != (int)kPackBits)
                spill = (spill & ~((3u << (2 * j)) | (3u << (2 * (j + T))))) | (hf << (2 * j)) | (hr << (2 * (j + T)));
        }
        slots[j * 64 + lane] = pf;
        if (j < T - M::NST) slots[(j + T) * 64 + lane] = pr;
        else held_r[j - (T - M::NST)] = pr;
        __builtin_amdgcn_sched_barrier(0);  // one window position's hash chains at a time (registers)
    }
    // every lane has hashed its k-mers by now: the last slots take over the staging area
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < M::NST; ++i) slots[(M::S - M::NST + i) * 64 + lane] = held_r[i];
}

// The accumulators of one read: x0 / x1 = the two word columns of every slot, combined over the three lookups.
// !INV: x = AND of the table words (a lane that loaded nothing got 0 from the bounds check and ORs in its all-ones mask first).
// INV: the table holds the COMPLEMENT of the filter's bits (the engine's merged copy has such a twin, rb_engine.hip), x = OR of
// the loaded words -- the bounds check's 0 is then neutral by itself: no compare, no mask, four ORs per lookup instead of a
// compare, a select, four ORs and four ANDs, and the relative offsets need not outlive the loads (six registers).  A bin is hit
// where x stays 0; slots without a k-mer start all-ones, bits beyond a column's bins as well.
template <bool INV, int T, int NW>
__device__ __forceinline__ void multi_init(uint64_t (&x)[NW][2 * T], uint32_t n_kmers, const NarrowMerge &nm, int lane)
{
#pragma unroll
    for (int c = 0; c < NW; ++c) {
        const uint64_t valid = col_bits_mask(nm.col_bits[c]);
#pragma unroll
        for (int j = 0; j < 2 * T; ++j) {
            const bool ok = (uint32_t)((j % T) * 64 + lane) < n_kmers;
            x[c][j] = INV ? (ok ? ~valid : ~0ULL) : (ok ? valid : 0ULL);
        }
    }
}

// One pass of the windows over the slices of the table for the R reads of a wave (their packed block numbers in s_off[r][slot][lane]).
// NW: words per block (2: 16-byte blocks, one 16-byte gather per lookup; 4: blocks of three or four words at a stride of four, two
// gathers).  (Measured and left out, profiles/r06/negative_results.md: cache-policy bits on the gathers -- sc0 / sc1 change nothing, nt
// keeps the lines out of the L2 and costs a factor of 2.4 --, two slots per batch of gathers.)
template <int R, bool INV, int T, int NW>
__device__ __forceinline__ void multi_windows(uint64_t (&x)[R][NW][2 * T], const uint64_t (*s_off)[2 * T][64], const IbfDev &f, const PhaseCfg &ph, int lane,
                                              const uint32_t (&spill)[R])
{
    static_assert(NW == 1 || NW == 2 || NW == 4, "one- and two-word blocks, or the stride-4 layout of three- and four-word blocks");
    constexpr int S = 2 * T;
    constexpr uint32_t kBlockShift = NW == 1 ? 3u : NW == 2 ? 4u : 5u;  // log2 bytes from one block to the next
    const uint32_t slice_shift = (ph.shift >> 31) ? (0x80000000u | ((ph.shift & 0x7FFFFFFFu) << kBlockShift)) : min(31u, ph.shift + kBlockShift);
    const uint32_t table_bytes = f.n_blocks << kBlockShift;  // (< 2^26: the launcher checked)
    const uint32_t all = ph.n_slices >= 32 ? ~0u : (1u << ph.n_slices) - 1u;
    uint32_t done = 0;
    // reads whose gathers of a slot go out together: all of them where the registers allow (OR form of the two-word build: no offsets
    // kept), else one by one
    constexpr int RB = (INV && NW == 2) ? R : 1;
    constexpr int G = NW == 1 ? 1 : NW / 2;  // gathers per lookup (8 bytes for one-word blocks, else 16)
#pragma unroll 1
    while (done != all) {
        const uint32_t cur = phase_next_slice(done, ph);
        done |= 1u << cur;
        asm volatile("" ::: "memory");  // the packed offsets are read again in every window, never carried in registers
        const bool any_len = (slice_shift >> 31) != 0 && slice_shift != 0x80000000u;
        const uint32_t span0 = any_len ? (slice_shift & 0x7FFFFFFFu) : 1u << min(slice_shift, 31u);
        const uint32_t start = any_len ? cur * span0 : cur << min(slice_shift, 31u);
        // a slice ends with the table: "no lookup" lies beyond it (made scalar by hand: left to itself the compiler takes the clamp to
        // the vector unit and then wraps every gather in a readfirstlane loop over a descriptor it no longer knows to be uniform)
        const uint32_t span = (uint32_t)__builtin_amdgcn_readfirstlane((int)min(span0, table_bytes - min(start, table_bytes)));
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char *>(reinterpret_cast<const char *>(f.words)) + start, 0, (int)span, kBufRsrcWord3);
        const uint32_t nstart = 0u - start;
#pragma unroll
        for (int u = 0; u < S; ++u) {
#pragma unroll
            for (int r0 = 0; r0 < R; r0 += RB) {
                rb_u32x4 d[RB][3][G];
                uint32_t rel[RB][3];
#pragma unroll
                for (int rr = 0; rr < RB; ++rr) {
                    const uint64_t pk = s_off[r0 + rr][u][lane];
                    const uint32_t lo = (uint32_t)pk, hi = (uint32_t)(pk >> 32);
                    // (field extract + shift-and-add of the negated slice start: seven VALU operations per k-mer)
                    if constexpr (NW == 1) {  // 22-bit numbers, the third one's top two bits from the spill word
                        rel[rr][0] = ((lo & kPackMask1) << kBlockShift) + nstart;
                        rel[rr][1] = ((__builtin_amdgcn_alignbit(hi, lo, kPackBits1) & kPackMask1) << kBlockShift) + nstart;
                        rel[rr][2] = (((hi >> (2 * kPackBits1 - 32)) | (__builtin_amdgcn_ubfe(spill[r0 + rr], 2 * u, 2) << 20)) << kBlockShift) + nstart;
                    } else {
                        rel[rr][0] = ((lo & kPackMask) << kBlockShift) + nstart;
                        rel[rr][1] = ((__builtin_amdgcn_alignbit(hi, lo, kPackBits) & kPackMask) << kBlockShift) + nstart;
                        rel[rr][2] = (__builtin_amdgcn_ubfe(hi, 2 * kPackBits - 32, kPackBits) << kBlockShift) + nstart;
                    }
#pragma unroll
                    for (int h = 0; h < 3; ++h) {
#pragma unroll
                        for (int g = 0; g < G; ++g) {  // ("no lookup" stays out of range with bit 4 set: its offset ends in zeros)
                            if constexpr (NW == 1) {
                                const rb_u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rs, rel[rr][h], 0, 0);
                                d[rr][h][g] = rb_u32x4{w.x, w.y, 0u, 0u};
                            } else {
                                d[rr][h][g] = __builtin_amdgcn_raw_buffer_load_b128(rs, rel[rr][h] | (uint32_t)(16 * g), 0, 0);
                            }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int rr = 0; rr < RB; ++rr) {
#pragma unroll
                    for (int h = 0; h < 3; ++h) {
                        const uint32_t o = (!INV && rel[rr][h] >= span) ? 0xFFFFFFFFu : 0u;  // AND form: lanes that loaded nothing
#pragma unroll
                        for (int g = 0; g < G; ++g) {
                            const uint64_t w0 = (((uint64_t)(d[rr][h][g].y | o)) << 32) | (d[rr][h][g].x | o);
                            if constexpr (INV) x[r0 + rr][NW == 1 ? 0 : 2 * g][u] |= w0;
                            else x[r0 + rr][NW == 1 ? 0 : 2 * g][u] &= w0;
                            if constexpr (NW > 1) {
                                const uint64_t w1 = (((uint64_t)(d[rr][h][g].w | o)) << 32) | (d[rr][h][g].z | o);
                                if constexpr (INV) x[r0 + rr][2 * g + 1][u] |= w1;
                                else x[r0 + rr][2 * g + 1][u] &= w1;
                            }
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// per-bin sums of one read across the wave, maxima per member
template <bool INV, int T, int NW>
__device__ __forceinline__ void multi_finish(uint64_t (&x)[NW][2 * T], const NarrowMerge &nm, int lane, uint16_t *out_row)
{
    if constexpr (INV) {
#pragma unroll
        for (int c = 0; c < NW; ++c) {
#pragma unroll
            for (int j = 0; j < 2 * T; ++j) x[c][j] = ~x[c][j];
        }
    }
    uint32_t colmax[NW];  // lane b: the larger of the two strands' counts of bin 64 c + b (at most 384 each: two share a register)
    if constexpr (NW == 1) {
        const uint32_t cf = wave_bin_counts<T>(x[0], lane), cr = wave_bin_counts<T>(x[0] + T, lane);
        colmax[0] = max(cf, cr);
    }
#pragma unroll
    for (int c = 0; c + 1 < NW; c += 2) {
        const uint32_t cf = wave_bin_counts<T>(x[c], lane) | (wave_bin_counts<T>(x[c + 1], lane) << 16);
        const uint32_t cr = wave_bin_counts<T>(x[c] + T, lane) | (wave_bin_counts<T>(x[c + 1] + T, lane) << 16);
        colmax[c] = max(cf & 0xFFFFu, cr & 0xFFFFu);
        colmax[c + 1] = max(cf >> 16, cr >> 16);
    }
    write_member_maxima<NW>(colmax, nm, lane, out_row);
}

// waves per SIMD the builds are compiled for: two-word four tiles 8 (R = 1: 50 registers) / 5 (R = 2: 91), six tiles 6; four-word 4
constexpr int multi_min_waves(int r, int t, int nw) { return nw == 1 ? (t == 4 ? 8 : 7) : nw == 4 ? (t == 4 ? RB_MULTI_WAVES_W4 : 3) : t == 4 ? (r == 1 ? 8 : RB_MULTI_WAVES) : (r == 1 ? 6 : 3); }

template <int R, bool INV, int T = 4, int NW = 2>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(multi_min_waves(R, T, NW), 8))) void ibf_count_max_phased_multi_kernel(
    IbfDev f, ReadSrc src, uint32_t n_reads, PhaseCfg ph, uint16_t *__restrict__ out, uint32_t out_read_stride, NarrowMerge nm)
{
    constexpr int S = 2 * T;  // slots per lane: the 64-k-mer tiles of forward k-mers, then those of the reverse complement
    __shared__ uint64_t s_off[R][S][64];
    if (ph.xcd_skew) ph.skew = xcc_id();
    if (ph.tskew) ph.tskew *= xcc_id();
    const int lane = threadIdx.x;
    const uint32_t read0 = blockIdx.x * (uint32_t)R;
    uint32_t nk[R];  // k-mers of each read (0: no such read, or a read longer than promised -- it writes 0 like the one-read build)
    uint32_t spill[R];  // one-word blocks: the top two bits of every slot's third block number
    constexpr int PB = NW == 1 ? (int)kPackBits1 : (int)kPackBits;
    static_assert(NW != 1 || R == 1, "the 22-bit packing is built for one read per wave");

    // ---- phase A: hash every k-mer of every read once; the block numbers go to LDS
#pragma unroll
    for (int r = 0; r < R; ++r) nk[r] = 0, spill[r] = ~0u;
#pragma unroll 1
    for (int r = 0; r < R; ++r) {
        uint32_t sp = ~0u;
        const uint32_t n = multi_hash_read<T, PB>(&s_off[r][0][0], f, src, read0 + (uint32_t)r, n_reads, lane, sp);
#pragma unroll
        for (int q = 0; q < R; ++q)
            if (q == r) nk[q] = n, spill[q] = sp;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();

    // ---- phase B: the windows
    uint64_t x[R][NW][S];
#pragma unroll
    for (int r = 0; r < R; ++r) multi_init<INV, T, NW>(x[r], nk[r], nm, lane);
    multi_windows<R, INV, T, NW>(x, s_off, f, ph, lane, spill);

    // ---- per-bin sums across the wave, maxima per member
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const uint32_t rid = read0 + (uint32_t)r;
        if (rid >= n_reads) break;  // wave-uniform
        multi_finish<INV, T, NW>(x[r], nm, lane, out + (size_t)rid * out_read_stride);
    }
}

// the inverted twin of a merged copy (ibf_count_max_phased_multi_kernel<R, true>): dst = ~src, word by word
__global__ void invert_words_kernel(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst, uint64_t n_words)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) dst[i] = ~src[i];
}

// ---------------------------------------------------------------------------------------------
// The decision for read i from the raw maxima of all filters: the body of K2, also run by the latency kernel for the reads of a
// micro-batch of a one-filter engine (FoldJob).  The maxima are read past this thread's caches.
__device__ __forceinline__ uint16_t load_count(const uint16_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void decide_one(const DecideParams &P, const uint16_t *maxcount, const uint32_t *lens,
                                           const uint8_t *pre_status, uint32_t i, int mode, int32_t *out_best_target,
                                           uint8_t *out_decision, uint8_t *out_status)
{
    const uint32_t nf = P.nd + P.nt;
    // raw maximum of (this read, filter fi): the max over the partial tables of the ranks (bin-sharded), or the one table
    auto raw_max = [&](uint32_t fi) -> uint32_t {
        uint16_t m = load_count(maxcount + (size_t)i * nf + fi);
        for (uint32_t q = 1; q < P.n_parts; ++q) {
            const uint16_t v = load_count(maxcount + (size_t)q * P.part_stride + (size_t)i * nf + fi);
            m = v > m ? v : m;
        }
        return m;
    };
    if (P.maxcount_copy) {
        for (uint32_t fi = 0; fi < nf; ++fi) P.maxcount_copy[(size_t)i * nf + fi] = (uint16_t)raw_max(fi);
    }
    if (pre_status && pre_status[i] != RB_OK) {  // e.g. a chunk beyond the end of the read: no decision
        if (out_best_target) out_best_target[i] = -1;
        if (out_decision) out_decision[i] = 0;
        if (out_status) out_status[i] = pre_status[i];
        return;
    }
    const uint32_t len = lens[i];
    if (len > P.max_len) {  // the caller understated max_len: counter width and threshold table were sized for less
        if (out_best_target) out_best_target[i] = -1;
        if (out_decision) out_decision[i] = 0;
        if (out_status) out_status[i] = RB_ERR_INVALID_ARG;
        return;
    }
    const uint32_t tl = len < P.thr_len ? len : P.thr_len - 1;  // len <= max_len < thr_len
    // group maxima at r (1) and at r - 0.02 (2), strictly-greater argmax at r (first wins ties)
    uint32_t D1 = 0, T1 = 0, D2 = 0, T2 = 0;
    int best_d = -1, best_t = -1;
    for (uint32_t fi = 0; fi < nf; ++fi) {
        const uint32_t M = raw_max(fi);
        uint32_t c1 = 0, c2 = 0;
        if (len >= P.k[fi]) {  // pair overload skips filters with k > len; others count 0 there anyway
            const uint16_t t1 = P.thr[((size_t)tl * nf + fi) * 2 + 0];
            const uint16_t t2 = P.thr[((size_t)tl * nf + fi) * 2 + 1];
            c1 = (M >= t1) ? M : 0;  // max_matches with the uint16_t threshold
            c2 = (M >= t2) ? M : 0;
        }
        if (fi < P.nd) {
            if (c1 > D1) { D1 = c1; best_d = (int)fi; }
            D2 = c2 > D2 ? c2 : D2;
        } else {
            if (c1 > T1) { T1 = c1; best_t = (int)(fi - P.nd); }
            T2 = c2 > T2 ? c2 : T2;
        }
    }
    uint8_t decision = 0, status = RB_OK;
    const bool short_d = P.nd && len < P.k[0];
    const bool short_t = P.nt && len < P.k[P.nd];
    if (mode == RB_MODE_CLASSIFY_ANY) {
        // Read::classify(std::vector<TIbf>&) -> find_matches -> select_matches (IBFClassify.cpp:181-226, 81-128, 16-38):
        // true iff some filter of the list (deplete entries first, then target) holds a bin with fwd >= t or rev >= t,
        // i.e. raw max >= t with the uint16_t threshold -- a threshold of 0 makes every read a hit, 0 matches included
        // (unlike max_matches, whose result is then 0 and reads as "no match").  No k > len skip here: find_matches
        // evaluates every filter; only filters[0].kmerSize is checked (ShortReadException).
        if (nf == 0) status = RB_ERR_NULL_FILTER;
        else if (len < P.k[0]) status = RB_ERR_SHORT_READ;
        else {
            bool found = false;
            for (uint32_t fi = 0; fi < nf; ++fi) found |= raw_max(fi) >= P.thr[((size_t)tl * nf + fi) * 2 + 0];
            decision = found ? 1 : 0;
        }
    } else if (mode == RB_MODE_CHECK_UNBLOCK) {
        if (P.nd && P.nt) {
            if (D1 > 0) {
                if (T1 > 0) decision = (D2 > 0 && T2 == 0) ? 1 : 0;
                else decision = 1;
            } else {
                decision = (T1 > 0) ? 2 : 0;
            }
        } else if (P.nd) {
            if (short_d) status = RB_ERR_SHORT_READ;
            else decision = (best_d > -1) ? 1 : 0;
        } else if (P.nt) {
            if (short_t) status = RB_ERR_SHORT_READ;
            else decision = (best_t < 0) ? 1 : 2;
        } else {
            status = RB_ERR_NULL_FILTER;
        }
    } else {  // RB_MODE_CLASSIFY_CHUNK
        if (P.nd && P.nt) {
            if (T1 > 0) {
                bool want_target = false;
                if (D1 > 0) {
                    if (T2 > 0 && D2 > 0) want_target = false;
                    else if (T2 > 0) want_target = true;
                } else {
                    want_target = true;
                }
                if (want_target) {  // r.classify(TargetFilters, Conf) at the restored error rate
                    if (short_t) status = RB_ERR_SHORT_READ;
                    else decision = (best_t != -1) ? 1 : 0;
                }
            }
        } else if (P.nd) {
            if (short_d) status = RB_ERR_SHORT_READ;
            else decision = (best_d > -1) ? 1 : 0;
        } else if (P.nt) {
            if (short_t) status = RB_ERR_SHORT_READ;
            else decision = (best_t != -1) ? 1 : 0;
        } else {
            status = RB_ERR_NULL_FILTER;
        }
    }
    if (out_best_target) out_best_target[i] = (P.nt && !short_t) ? best_t : -1;
    if (out_decision) out_decision[i] = decision;
    if (out_status) out_status[i] = status;
}

// every result of the call is in place: tell the host (DecideParams::done_flag).  Called by all threads of the decision kernel after their
// reads.  The results go to page-locked HOST memory and the host stops waiting for the stream once it sees the word, so every wave
// makes its own result stores visible at system scope before the barrier (thread 0's release orders only its own wave's
// stores behind the flag; the barrier alone is a workgroup-scope matter) -- four fences per 256 reads.
__device__ __forceinline__ void announce_done(const DecideParams &P)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (threadIdx.x != 0) return;
    if (gridDim.x > 1) {
        const uint32_t t = __hip_atomic_fetch_add(P.done_count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        if (t != gridDim.x - 1) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *P.done_count = 0;  // ready for the next call on this stream
    }
    __hip_atomic_store(P.done_flag, P.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// latency kernel of a one-filter engine, after its body: `fin` = the read whose raw maximum this workgroup has just written, or ~0u
__device__ __forceinline__ void fold_decide(const FoldJob &job, uint32_t fin)
{
    if (!job.on || fin == ~0u || threadIdx.x != 0) return;  // thread 0 wrote the maximum
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");  // own store before own load of the same element
    decide_one(job.P, job.maxcount, job.lens, job.pre_status, fin, job.mode, job.best_target, job.decision, job.status);
    // a call of ONE read (the engine asks for the word in a folded launch only then): this thread holds the call's last result
    if (job.P.done_flag) __hip_atomic_store(job.P.done_flag, job.P.done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// latency form for micro-batches: `parts` workgroups per (read, column slice).  Wave w of part p takes strand w&1 and
// slot = p * (waves/2) + (w>>1) of the per-strand work: share slot % sub of the eight-step blocks of every
// (slots/sub)-th macro tile starting at slot / sub.  Partial counters meet in LDS (bit-sliced adds); with parts > 1 the
// workgroups leave their sums in a workspace and the last one to finish (ticket counter) adds them and takes the max.
// A wide filter is latency bound per read -- 2088 dependent-free gathers, but only 12-24 of them in flight per wave --
// so the way to a short kernel is more waves per read than one workgroup holds.
// blockIdx.y selects one filter of the set; all filters of a micro-batch share ONE launch whatever their geometry (a
// micro-batch against deplete + several targets would otherwise queue one short kernel per filter, 10-25 us each, and
// the short kernel of a narrow target would wait for the long one of the wide deplete filter instead of hiding in it).
// The launch has grid_parts workgroups per (read, slice); a filter that wants fewer leaves the others idle.
template <int LG, int WPL, int NP, int H, bool NT>
__device__ __forceinline__ uint32_t split_body(const IbfDev &f, uint32_t col_begin, uint32_t col_end, uint32_t parts,
                                           uint32_t sub, const ReadSrc &src, uint32_t n_reads, uint32_t n_slices,
                                           uint16_t *__restrict__ out, uint32_t out_read_stride,
                                           uint32_t out_slice_stride, uint32_t grid_parts, uint64_t *__restrict__ ws,
                                           uint32_t *__restrict__ tickets, uint8_t *s_dyn)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int n_waves = blockDim.x >> 6;
    const int wps = n_waves >> 1;  // waves per strand in this workgroup
    const int strand = wave & 1, lslot = wave >> 1;
    const uint32_t item = blockIdx.x / grid_parts;
    const uint32_t part = blockIdx.x - item * grid_parts;
    if (part >= parts) return ~0u;  // workgroup-uniform: this filter uses fewer workgroups per read than the launch has
    const uint32_t read = item / n_slices;
    const uint32_t slice = item - read * n_slices;
    const uint32_t slots = (uint32_t)wps * parts;  // per strand, over all parts
    const uint32_t slot = part * (uint32_t)wps + (uint32_t)lslot;

    uint64_t *s_planes = reinterpret_cast<uint64_t *>(s_dyn);  // [wave][WPL][NP][64]
    uint32_t *s_max = reinterpret_cast<uint32_t *>(s_dyn + (size_t)n_waves * WPL * NP * 64 * 8);  // [0..1] maxima, [2] ticket
    uint8_t *stage = s_dyn + (size_t)n_waves * WPL * NP * 64 * 8 + 16 + (size_t)wave * kStageBytes;

    const LaneCols<WPL> lc = make_lane_cols<LG, WPL>(f, lane, col_begin, col_end, slice);
    uint32_t len;
    const BaseSrc seq = make_base_src(src, read, &len);
    const uint32_t n = len >= f.k ? len - f.k + 1 : 0;
    constexpr uint32_t ITEMS = TileShape<LG>::ITEMS;
    constexpr int BPT = TileShape<LG>::STEPS / 8;  // eight-step blocks per macro tile
    const int bps = BPT / (int)sub;                // ... per share (sub divides BPT and slots)
    const int sb = (int)(slot % sub);

    Planes<NP> pl[WPL];
#pragma unroll
    for (int w = 0; w < WPL; ++w) pl[w].clear();
    count_strand<LG, WPL, NP, H, NT>(pl, f, lc, seq, len, n, strand, (slot / sub) * ITEMS, (slots / sub) * ITEMS, sb * bps,
                                     (sb + 1) * bps, stage, lane);

    if (lslot != 0) {
#pragma unroll
        for (int w = 0; w < WPL; ++w)
#pragma unroll
            for (int i = 0; i < NP; ++i) s_planes[(((size_t)wave * WPL + w) * NP + i) * 64 + lane] = pl[w].p[i];
    }
    __syncthreads();
    const size_t gitem = (size_t)blockIdx.y * ((size_t)n_reads * n_slices) + item;
    if (lslot == 0) {
        for (int o = 1; o < wps; ++o) {
            const int ow = (o << 1) | strand;
#pragma unroll
            for (int w = 0; w < WPL; ++w) {
                uint64_t carry = 0;
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    const uint64_t other = s_planes[(((size_t)ow * WPL + w) * NP + i) * 64 + lane];
                    uint64_t h, l;
                    RB_CSA(h, l, pl[w].p[i], other, carry);
                    pl[w].p[i] = l;
                    carry = h;
                }
            }
        }
        if (parts == 1) {
            const uint32_t m = planes_max<NP, WPL>(pl, lc.valid);
            if (lane == 0) s_max[strand] = m;
        } else {
            uint64_t *dst = ws + ((gitem * grid_parts + part) * 2 + (size_t)strand) * (2 * NP * 64);
#pragma unroll
            for (int w = 0; w < WPL; ++w)
#pragma unroll
                for (int i = 0; i < NP; ++i) dst[(w * NP + i) * 64 + lane] = pl[w].p[i];
        }
    }
    __syncthreads();
    if (parts > 1) {
        // The last workgroup of this (read, slice, filter) to arrive owns the result.  One agent-scope release per
        // workgroup (the barrier above orders the other waves' stores before it) and one acquire in the last one (the
        // barrier below orders it before the other waves' loads): on gfx950 a release writes the XCD's L2 back, which
        // costs about a microsecond and serialises per XCD -- fences per wave made 64-read batches 2.4x slower.
        if (threadIdx.x == 0) {
            const uint32_t t = __hip_atomic_fetch_add(&tickets[gitem], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (t == parts - 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            s_max[2] = t;
        }
        __syncthreads();
        if (s_max[2] != parts - 1) return ~0u;  // workgroup-uniform
        if (lslot == 0) {
#pragma unroll
            for (int w = 0; w < WPL; ++w) pl[w].clear();
            for (uint32_t q = 0; q < parts; ++q) {
                const uint64_t *srcp = ws + ((gitem * grid_parts + q) * 2 + (size_t)strand) * (2 * NP * 64);
#pragma unroll
                for (int w = 0; w < WPL; ++w) {
                    uint64_t carry = 0;
#pragma unroll
                    for (int i = 0; i < NP; ++i) {
                        const uint64_t other = srcp[(w * NP + i) * 64 + lane];
                        uint64_t h, l;
                        RB_CSA(h, l, pl[w].p[i], other, carry);
                        pl[w].p[i] = l;
                        carry = h;
                    }
                }
            }
            const uint32_t m = planes_max<NP, WPL>(pl, lc.valid);
            if (lane == 0) s_max[strand] = m;
        }
        __syncthreads();
        if (threadIdx.x == 0) tickets[gitem] = 0;  // ready for the next launch on this stream
    }
    if (threadIdx.x == 0) {
        const uint32_t a = s_max[0], b = s_max[1];
        out[(size_t)read * out_read_stride + (size_t)slice * out_slice_stride] = (uint16_t)(a > b ? a : b);
    }
    return item;  // the (read, slice) this workgroup has written (a FoldJob implies one slice: item = read)
}


// filters of ONE kernel geometry (single-filter engines, several targets of equal width): the body alone, with the
// workgroup size its register budget allows
template <int LG, int WPL, int NP, int H, bool NT>
__global__ __launch_bounds__(WPL == 2 ? (NP > 10 ? 512 : 768) : (LG == 0 ? 512 : 1024)) void ibf_count_max_split_kernel(
    FilterSet set, ReadSrc src, uint32_t n_reads, uint32_t n_slices, uint16_t *__restrict__ out_base,
    uint32_t out_read_stride, uint32_t out_slice_stride, uint32_t grid_parts, uint64_t *__restrict__ ws,
    uint32_t *__restrict__ tickets, FoldJob job)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    const uint32_t fin = split_body<LG, WPL, NP, H, NT>(set.f[blockIdx.y], set.col_begin[blockIdx.y], set.col_end[blockIdx.y],
                                                        set.parts[blockIdx.y], set.sub[blockIdx.y], src, n_reads, n_slices,
                                                        out_base + set.out_offset[blockIdx.y], out_read_stride, out_slice_stride,
                                                        grid_parts, ws, tickets, s_dyn);
    fold_decide(job, fin);
}

// filters of DIFFERENT geometries in one launch (deplete = human genome, targets = a few small genomes): the geometry
// of the filter picked by blockIdx.y selects the body (workgroup-uniform branch).  Built for 512 threads: with all
// bodies in one function the scalar state of sixteen specialisations spills into vector registers, and at 768+
// threads that pushes the widest bodies over the budget.
template <int NP, bool WIDE>
__global__ __launch_bounds__(512) void ibf_count_max_split_any_kernel(
    FilterSet set, ReadSrc src, uint32_t n_reads, uint32_t n_slices, uint16_t *__restrict__ out_base,
    uint32_t out_read_stride, uint32_t out_slice_stride, uint32_t grid_parts, uint64_t *__restrict__ ws,
    uint32_t *__restrict__ tickets, FoldJob job)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    uint32_t fin = ~0u;
    const IbfDev &f = set.f[blockIdx.y];
    const uint32_t col_begin = set.col_begin[blockIdx.y], col_end = set.col_end[blockIdx.y];
    uint16_t *__restrict__ out = out_base + set.out_offset[blockIdx.y];
    const uint32_t parts = set.parts[blockIdx.y], sub = set.sub[blockIdx.y];
#define RB_SPLIT_CASE(LG_, WPL_)                                                                                          \
    case (LG_) | ((WPL_) == 2 ? 8 : 0):                                                                                    \
        fin = split_body<LG_, WPL_, NP, 3, false>(f, col_begin, col_end, parts, sub, src, n_reads, n_slices, out,          \
                                                  out_read_stride, out_slice_stride, grid_parts, ws, tickets, s_dyn);      \
        break;                                                                                                             \
    case (LG_) | ((WPL_) == 2 ? 8 : 0) | 16:                                                                               \
        fin = split_body<LG_, WPL_, NP, 3, true>(f, col_begin, col_end, parts, sub, src, n_reads, n_slices, out,           \
                                                 out_read_stride, out_slice_stride, grid_parts, ws, tickets, s_dyn);       \
        break;
    switch (set.geom[blockIdx.y]) {
        RB_SPLIT_CASE(0, 1)
        RB_SPLIT_CASE(1, 1)
        RB_SPLIT_CASE(2, 1)
        RB_SPLIT_CASE(3, 1)
        RB_SPLIT_CASE(4, 1)
        RB_SPLIT_CASE(5, 1)
        RB_SPLIT_CASE(6, 1)
    default:
        if constexpr (WIDE) {
            switch (set.geom[blockIdx.y]) {
                RB_SPLIT_CASE(6, 2)
            default: break;
            }
        }
        break;
    }
#undef RB_SPLIT_CASE
    fold_decide(job, fin);
}

// combine the per-slice partial maxima of one filter: part[slice][read] -> maxcount[read*nf + f]
__global__ void reduce_slices_kernel(const uint16_t *__restrict__ part, uint32_t n_slices, uint32_t n_reads,
                                     uint16_t *__restrict__ maxcount, uint32_t nf, uint32_t fidx)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_reads) return;
    uint16_t m = 0;
    for (uint32_t s = 0; s < n_slices; ++s) {
        const uint16_t v = part[(size_t)s * n_reads + i];
        m = v > m ? v : m;
    }
    maxcount[(size_t)i * nf + fidx] = m;
}

// ---------------------------------------------------------------------------------------------
// on-GPU chunking (classify.hpp:264-271): work item i looks at bases [chunk_start, min(chunk_start+chunk_len, len)) of
// its read; a chunk that starts beyond the read's end is the reference's undefined infix -> RB_ERR_BAD_CHUNK
__global__ void chunk_prep_kernel(const uint32_t *__restrict__ lens, const uint32_t *__restrict__ ids, uint32_t n_items,
                                  uint32_t chunk_start, uint32_t chunk_len, uint32_t *__restrict__ eff_lens,
                                  uint8_t *__restrict__ pre_status)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    const uint32_t len = lens[ids ? ids[i] : i];
    const bool bad = chunk_start > len;
    const uint32_t rest = bad ? 0u : len - chunk_start;
    eff_lens[i] = (chunk_len && chunk_len < rest) ? chunk_len : rest;
    pre_status[i] = bad ? (uint8_t)RB_ERR_BAD_CHUNK : (uint8_t)RB_OK;
}

// K2: one thread per read.
__global__ void decide_kernel(DecideParams P, const uint16_t *__restrict__ maxcount, const uint32_t *__restrict__ lens,
                              const uint8_t *__restrict__ pre_status, uint32_t n_reads, int mode,
                              int32_t *__restrict__ out_best_target, uint8_t *__restrict__ out_decision,
                              uint8_t *__restrict__ out_status)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_reads) decide_one(P, maxcount, lens, pre_status, i, mode, out_best_target, out_decision, out_status);
    if (P.done_flag) announce_done(P);  // (kernel argument: uniform)
}

// ---------------------------------------------------------------------------------------------
// K4: one thread per (fragment, k-mer position); h atomicOr per k-mer.
__global__ void ibf_insert_kernel(IbfDev f, uint64_t *__restrict__ words, const uint8_t *__restrict__ seq,
                                  const uint64_t *__restrict__ starts, const uint64_t *__restrict__ ends,
                                  const uint64_t *__restrict__ bins, const uint64_t *__restrict__ kmer_prefix,
                                  uint32_t n_fragments, uint64_t total_kmers)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total_kmers) return;
    // fragment lookup: kmer_prefix[i] = number of k-mers in fragments < i (binary search)
    uint32_t lo = 0, hi = n_fragments;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (kmer_prefix[mid] <= t) lo = mid; else hi = mid;
    }
    const uint64_t pos = starts[lo] + (t - kmer_prefix[lo]);
    if (pos + f.k > ends[lo]) return;
    uint64_t v = 0;
    for (uint32_t i = 0; i < f.k; ++i) v = v * 5u + rbspec::dna5_ord(seq[pos + i]);
    const uint64_t bin = bins[lo];
    for (uint32_t h = 0; h < f.n_hash; ++h) {
        const uint64_t blk = rbspec::block_index(v, f.precalc[h], f.n_blocks, f.magic, f.pow2_mask);
        const uint64_t bit = blk * ((uint64_t)f.stride * 64u) + bin;  // padded HBM layout
        atomicOr(reinterpret_cast<unsigned long long *>(words + (bit >> 6)), 1ULL << (bit & 63));
    }
}

// Re-stride a block matrix: block b = src[b*s_src .. +w_copy) -> dst[b*s_dst .. +w_copy), the rest of each destination
// block (s_dst - w_copy words) is zero.  Serves resizeBins (wider blocks, new bins empty), the padded HBM layout
// (file layout -> aligned blocks) and the way back (download).
__global__ void restride_blocks_kernel(const uint64_t *__restrict__ src, uint32_t s_src, uint64_t *__restrict__ dst,
                                       uint32_t s_dst, uint32_t w_copy, uint64_t n_blocks)
{
    const uint64_t total = n_blocks * s_dst;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const uint64_t b = i / s_dst;
        const uint32_t c = (uint32_t)(i - b * s_dst);
        dst[i] = c < w_copy ? src[b * s_src + c] : 0ULL;
    }
}

// first-contact check (rb_dibf_compare): bit statistics of two filters of one geometry -- out[0] = bits set in a,
// out[1] = bits set in b, out[2] = bits set in b but not in a.  One 64-bit atomic per wave.
__global__ void compare_bits_kernel(const uint64_t *__restrict__ a, const uint64_t *__restrict__ b, uint64_t n_words,
                                    unsigned long long *__restrict__ out)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t sa = 0, sb = 0, snew = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride) {
        const uint64_t x = a[i], y = b[i];
        sa += (uint64_t)__popcll(x);
        sb += (uint64_t)__popcll(y);
        snew += (uint64_t)__popcll(y & ~x);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        sa += shfl64(sa, (threadIdx.x & 63) ^ m);
        sb += shfl64(sb, (threadIdx.x & 63) ^ m);
        snew += shfl64(snew, (threadIdx.x & 63) ^ m);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(out + 0, (unsigned long long)sa);
        atomicAdd(out + 1, (unsigned long long)sb);
        atomicAdd(out + 2, (unsigned long long)snew);
    }
}

// word w of the FILE layout (block w / bin_width, column w % bin_width) gets synth_word(seed, w); it is stored at the
// padded position of that block
__global__ void fill_synth_kernel(uint64_t *__restrict__ words, uint64_t used_words, uint32_t bin_width, uint32_t stride_words,
                                  uint64_t last_mask, uint64_t seed)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < used_words; w += stride) {
        const uint64_t b = w / bin_width;
        const uint32_t c = (uint32_t)(w - b * bin_width);
        uint64_t x = rbspec::synth_word(seed, w);
        if (c == bin_width - 1) x &= last_mask;
        words[b * stride_words + c] = x;
    }
}

// synthetic batch for rb_engine_calibrate: n_reads reads of read_len uniform ACGT bases, back to back
__global__ void fill_reads_kernel(uint8_t *__restrict__ seqs, uint64_t *__restrict__ offsets, uint32_t *__restrict__ lens, uint64_t n_reads,
                                  uint32_t read_len, uint64_t seed)
{
    const uint64_t total = n_reads * read_len;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        // 32 bases per 64-bit word of a UNIFORM mixer (not synth_word: its bits are set with the filters' design load, 0.215 -- reads
        // of 62 % A would repeat their k-mers, hit the caches and mislead the calibration: round 4 saw it pick windows that were 13 %
        // slower on real reads)
        const uint64_t w = rbspec::mix64(seed + ((i >> 5) + 1) * 0x9E3779B97F4A7C15ULL);
        seqs[i] = (uint8_t)"ACGT"[(w >> ((i & 31u) * 2u)) & 3u];
        if (i < n_reads) {
            offsets[i] = i * read_len;
            lens[i] = read_len;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
template <int LG, int WPL, int NP, int H, bool NT>
static hipError_t launch_count_nt(const CountLaunch &a, hipStream_t st)
{
    FilterSet set;
    set.n = a.n_fused > 0 ? (uint32_t)a.n_fused : 1u;
    if (a.n_fused > 0) {
        for (uint32_t i = 0; i < set.n; ++i) {
            set.f[i] = a.fused_f[i];
            set.col_begin[i] = a.fused_col_begin[i];
            set.col_end[i] = a.fused_col_end[i];
            set.out_offset[i] = a.fused_out_offset[i];
        }
    } else {
        set.f[0] = a.f;
        set.col_begin[0] = a.col_begin;
        set.col_end[0] = a.col_end;
        set.out_offset[0] = 0;
    }
    const uint64_t items = (uint64_t)a.n_reads * a.n_slices;
    dim3 grid((uint32_t)((items + kWavesPerBlock - 1) / kWavesPerBlock), set.n);
    EarlyCfg early{};
    if constexpr (H == 3 && NP == 10) {  // (the early-decision builds exist for three hash functions and ten counter planes: what config 3 / 4 take)
        if (a.early_thr && a.n_fused == 0) {
            early.thr = a.early_thr;
            early.thr_len = a.early_thr_len;
            early.nf = a.early_nf;
            early.fi[0] = a.early_fi;
            hipLaunchKernelGGL((ibf_count_max_kernel<LG, WPL, NP, H, NT, true>), grid, dim3(64 * kWavesPerBlock), 0, st, set, a.src,
                               a.n_reads, a.n_slices, a.out, a.out_read_stride, a.out_slice_stride, early);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((ibf_count_max_kernel<LG, WPL, NP, H, NT>), grid, dim3(64 * kWavesPerBlock), 0, st, set, a.src,
                       a.n_reads, a.n_slices, a.out, a.out_read_stride, a.out_slice_stride, early);
    return hipGetLastError();
}

// latency form.  Dynamic LDS = plane exchange + maxima + per-wave staging; the opt-in for > 64 KiB of it is per function
// AND per device, remembered per device id.
template <typename K>
static hipError_t launch_split_kernel(K kern, std::atomic<uint64_t> &done, const FilterSet &set, const CountLaunch &a, int max_wpl,
                                      int np, uint32_t grid_parts, hipStream_t st)
{
    const int nw = a.split_waves;
    const size_t lds = (size_t)nw * max_wpl * np * 64 * 8 + 16 + (size_t)nw * kStageBytes;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = (dev >= 0 && dev < 64) ? (1ULL << dev) : 0;
    if (!bit || !(done.load(std::memory_order_acquire) & bit)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           160 * 1024);
        if (e != hipSuccess) return e;
        done.fetch_or(bit, std::memory_order_release);
    }
    dim3 grid(a.n_reads * a.n_slices * grid_parts, set.n);
    FoldJob job{};
    if (a.fold) {
        if (a.n_slices != 1 || set.n != 1 || !a.fold->maxcount || !a.fold->lens) return hipErrorInvalidValue;
        job = *a.fold;
    }
    hipLaunchKernelGGL(kern, grid, dim3(64 * nw), lds, st, set, a.src, a.n_reads, a.n_slices, a.out, a.out_read_stride,
                       a.out_slice_stride, grid_parts, a.split_ws, a.split_tickets, job);
    return hipGetLastError();
}

template <int LG, int WPL, int NP, bool NT>
static hipError_t launch_split_one(const FilterSet &set, const CountLaunch &a, uint32_t grid_parts, hipStream_t st)
{
    static std::atomic<uint64_t> done{0};
    return launch_split_kernel(ibf_count_max_split_kernel<LG, WPL, NP, 3, NT>, done, set, a, WPL, NP, grid_parts, st);
}

template <int NP, bool NT>
static hipError_t launch_split_same(int lg, int wpl, const FilterSet &set, const CountLaunch &a, uint32_t grid_parts,
                                    hipStream_t st)
{
    if (wpl == 2) return launch_split_one<6, 2, NP, NT>(set, a, grid_parts, st);
    switch (lg) {
    case 0: return launch_split_one<0, 1, NP, NT>(set, a, grid_parts, st);
    case 1: return launch_split_one<1, 1, NP, NT>(set, a, grid_parts, st);
    case 2: return launch_split_one<2, 1, NP, NT>(set, a, grid_parts, st);
    case 3: return launch_split_one<3, 1, NP, NT>(set, a, grid_parts, st);
    case 4: return launch_split_one<4, 1, NP, NT>(set, a, grid_parts, st);
    case 5: return launch_split_one<5, 1, NP, NT>(set, a, grid_parts, st);
    default: return launch_split_one<6, 1, NP, NT>(set, a, grid_parts, st);
    }
}

template <int NP>
static hipError_t launch_split(const CountLaunch &a, hipStream_t st)
{
    FilterSet set;
    set.n = a.n_fused > 0 ? (uint32_t)a.n_fused : 1u;
    if (a.n_fused > 0) {
        for (uint32_t i = 0; i < set.n; ++i) {
            set.f[i] = a.fused_f[i];
            set.col_begin[i] = a.fused_col_begin[i];
            set.col_end[i] = a.fused_col_end[i];
            set.out_offset[i] = a.fused_out_offset[i];
            set.geom[i] = a.fused_geom[i];
            set.parts[i] = a.fused_parts[i];
            set.sub[i] = a.fused_sub[i];
        }
    } else {
        set.f[0] = a.f;
        set.col_begin[0] = a.col_begin;
        set.col_end[0] = a.col_end;
        set.out_offset[0] = 0;
        set.geom[0] = geom_code(a.lg, a.wpl, a.nt);
        set.parts[0] = (uint32_t)(a.split_parts > 1 ? a.split_parts : 1);
        set.sub[0] = (uint32_t)(a.split_sub > 1 ? a.split_sub : 1);
    }
    const uint32_t grid_parts = a.grid_parts > 1 ? (uint32_t)a.grid_parts : 1u;
    bool same = true, wide = false;
    for (uint32_t i = 0; i < set.n; ++i) {
        if (set.parts[i] < 1 || set.parts[i] > grid_parts || set.sub[i] < 1) return hipErrorInvalidValue;
        same &= set.geom[i] == set.geom[0];
        wide |= (set.geom[i] & 8) != 0;
    }
    if (grid_parts > 1 && (!a.split_ws || !a.split_tickets)) return hipErrorInvalidValue;
    const int nw = a.split_waves;
    const int lg = (int)(set.geom[0] & 7), wpl = (set.geom[0] & 8) ? 2 : 1;
    const int cap = same ? split_waves_cap(wpl, NP, lg) : kSplitAnyWaves;
    if (nw < 2 || (nw & 1) || nw > cap) return hipErrorInvalidValue;
    if (same)
        return (set.geom[0] & 16) ? launch_split_same<NP, true>(lg, wpl, set, a, grid_parts, st)
                                  : launch_split_same<NP, false>(lg, wpl, set, a, grid_parts, st);
    static std::atomic<uint64_t> done[2];
    if (wide) return launch_split_kernel(ibf_count_max_split_any_kernel<NP, true>, done[1], set, a, 2, NP, grid_parts, st);
    return launch_split_kernel(ibf_count_max_split_any_kernel<NP, false>, done[0], set, a, 1, NP, grid_parts, st);
}

template <int LG, int WPL, int NP, int H>
static hipError_t launch_count(const CountLaunch &a, hipStream_t st)
{
    return a.nt ? launch_count_nt<LG, WPL, NP, H, true>(a, st) : launch_count_nt<LG, WPL, NP, H, false>(a, st);
}

// most waves a workgroup of the one-geometry latency kernel may have: the 16-byte-lane instantiations are built for
// 768 (10 planes) / 512 (16 planes) threads, the one-lane-per-block ones for 512 (eight tiles of block numbers in
// registers: at 1024 threads they spilled 36-49 registers; a 512-k-mer macro tile needs two waves anyway), the others
// for 1024, and the plane exchange of every wave has to fit the 160 KiB of LDS.  The mixed-geometry kernel takes
// kSplitAnyWaves.
int split_waves_cap(int wpl, int planes, int lg)
{
    const int np = planes <= 10 ? 10 : 16;
    const size_t per_wave = (size_t)wpl * np * 64 * 8 + kStageBytes;
    int cap = (int)((160 * 1024 - 16) / per_wave);
    const int by_bounds = wpl == 2 ? (np > 10 ? 8 : 12) : (lg == 0 ? 8 : 16);
    if (cap > by_bounds) cap = by_bounds;
    return cap & ~1;
}

// number of waves the latency form wants per read for this geometry (0 = latency form not applicable)
int split_waves_limit(int wpl, int planes, uint32_t max_kmers, int lg)
{
    int items = 64 * (lg >= 3 ? 1 : (8 >> lg));
    int tiles = (int)((max_kmers + items - 1) / items);
    int nw = 2 * (tiles < 1 ? 1 : tiles);
    const int cap = split_waves_cap(wpl, planes, lg);
    if (nw > cap) nw = cap;
    nw &= ~1;
    return nw >= 2 ? nw : 0;  // two waves = one per strand (reads that fit one macro tile, e.g. one-word filters)
}

// Several workgroups per read for the latency form (see ibf_count_max_split_kernel): worth it when a wave would
// otherwise walk many eight-step blocks in sequence, i.e. for blocks of 32+ lanes (filters of 17+ word columns).
// Returns parts (1 = keep the one-workgroup form and *nw untouched); with parts > 1 sets the waves per workgroup and
// the shares per macro tile.  parts * (*nw / 2) is a multiple of *sub, and *sub divides the blocks per macro tile.
int split_parts_plan(int wpl, int planes, uint32_t max_kmers, int lg, uint32_t n_items, uint32_t max_parts, uint32_t max_sub,
                     int *nw, int *sub)
{
    *sub = 1;
    if (max_parts <= 1 || lg < 5 || *nw < 2) return 1;
    const int bpt = lg == 6 ? 8 : 4;           // TileShape<LG>::STEPS / 8
    const uint32_t tiles = (max_kmers + 63) / 64;  // J == 1 for these shapes
    if (tiles == 0) return 1;
    const int wps = kSplitAnyWaves / 2;        // waves per strand and workgroup: fits every kernel of the latency form
    uint32_t cap = max_parts;
    // measured on the 8 GiB filter: beyond ~200 workgroups per launch the extra parts only queue behind each other
    // (64 reads: 3 parts 66 us, 4 parts 72 us, 6 parts 87 us, 1 part 82 us; 256 reads: 1 part is best)
    const uint32_t by_grid = n_items ? 200u / n_items : 1u;
    if (cap > by_grid) cap = by_grid;
    if (cap <= 1) return 1;
    int s0 = 1;
    while (s0 * 2 <= bpt && (uint32_t)(s0 * 2) <= max_sub) s0 *= 2;
    for (int s = s0; s >= 1; s >>= 1) {
        uint32_t p = (tiles * (uint32_t)s + (uint32_t)wps - 1) / (uint32_t)wps;
        if (p > cap) p = cap;
        const uint32_t m = s > wps ? (uint32_t)(s / wps) : 1u;
        p = p / m * m;
        if (p >= 2) {
            *nw = 2 * wps;
            *sub = s;
            return (int)p;
        }
    }
    return 1;
}

template <int NP, int H>
static hipError_t dispatch_geometry(const CountLaunch &a, hipStream_t st)
{
    if (a.wpl == 2) return launch_count<6, 2, NP, H>(a, st);
    switch (a.lg) {
    case 0: return launch_count<0, 1, NP, H>(a, st);
    case 1: return launch_count<1, 1, NP, H>(a, st);
    case 2: return launch_count<2, 1, NP, H>(a, st);
    case 3: return launch_count<3, 1, NP, H>(a, st);
    case 4: return launch_count<4, 1, NP, H>(a, st);
    case 5: return launch_count<5, 1, NP, H>(a, st);
    default: return launch_count<6, 1, NP, H>(a, st);
    }
}

template <int LG, int NP>
static hipError_t launch_phased(const CountLaunch &a, hipStream_t st)
{
    dim3 grid((a.n_reads + kWavesPerBlock - 1) / kWavesPerBlock);
    // two-word blocks, reads of up to 256 / 384 k-mers, tables whose block numbers fit 21 bits: offsets in LDS, one or two reads per wave
    if constexpr (LG == 1 && NP == 10) {
        if (a.multi_reads && a.multi_tiles && a.col_begin == 0 && a.col_end == 2 && a.f.stride == 2) {
            if (a.f.n_blocks > kPackMask) return hipErrorInvalidValue;
            const uint32_t R = a.multi_tiles == 6 ? 1u : (uint32_t)a.multi_reads;  // (six tiles: one read per wave)
            dim3 g2((a.n_reads + R - 1) / R);
#define RB_LAUNCH_MULTI(RR, INV, TT)                                                                                                    \
    hipLaunchKernelGGL((ibf_count_max_phased_multi_kernel<RR, INV, TT, 2>), g2, dim3(64), 0, st, a.f, a.src, a.n_reads, a.phase, a.out, \
                       a.out_read_stride, a.narrow)
            if (a.multi_tiles == 6 && a.multi_inv) RB_LAUNCH_MULTI(1, true, 6);
            else if (a.multi_tiles == 6) RB_LAUNCH_MULTI(1, false, 6);
            else if (R == 1 && a.multi_inv) RB_LAUNCH_MULTI(1, true, 4);
            else if (R == 1) RB_LAUNCH_MULTI(1, false, 4);
            else if (R == 2 && a.multi_inv) RB_LAUNCH_MULTI(2, true, 4);
            else if (R == 2) RB_LAUNCH_MULTI(2, false, 4);
            else return hipErrorInvalidValue;
#undef RB_LAUNCH_MULTI
            return hipGetLastError();
        }
    }
    // one-word blocks the same way (a filter of up to 64 bins on its own: the AND form)
    if constexpr (LG == 0 && NP == 10) {
        if (a.multi_reads && a.multi_tiles && a.col_begin == 0 && a.col_end == 1 && a.f.stride == 1) {
            if (a.f.n_blocks >= kPackMask1) return hipErrorInvalidValue;  // (0x3FFFFF itself is "no lookup")
            dim3 g1(a.n_reads);
            if (a.multi_tiles == 6)
                hipLaunchKernelGGL((ibf_count_max_phased_multi_kernel<1, false, 6, 1>), g1, dim3(64), 0, st, a.f, a.src, a.n_reads, a.phase, a.out, a.out_read_stride, a.narrow);
            else
                hipLaunchKernelGGL((ibf_count_max_phased_multi_kernel<1, false, 4, 1>), g1, dim3(64), 0, st, a.f, a.src, a.n_reads, a.phase, a.out, a.out_read_stride, a.narrow);
            return hipGetLastError();
        }
    }
    // both-strands-only build when every read of the batch fits it (LG 0/1, whole blocks owned by this rank)
    if constexpr (LG <= 1 && NP == 10) {
        if (a.short_only == 1 && (LG == 0 || (a.col_begin == 0 && a.col_end == 2 && a.f.stride == 2))) {
            hipLaunchKernelGGL((ibf_count_max_phased_kernel<LG, NP, 1>), grid, dim3(64 * kWavesPerBlock), 0, st, a.f, a.col_begin,
                               a.col_end, a.src, a.n_reads, a.phase, a.out, a.out_read_stride, a.narrow);
            return hipGetLastError();
        }
    }
    if constexpr (LG <= 1 && NP == 10) {
        if (a.short_only == 2 && (LG == 0 || (a.col_begin == 0 && a.col_end == 2 && a.f.stride == 2))) {
            hipLaunchKernelGGL((ibf_count_max_phased_kernel<LG, NP, 2>), grid, dim3(64 * kWavesPerBlock), 0, st, a.f, a.col_begin,
                               a.col_end, a.src, a.n_reads, a.phase, a.out, a.out_read_stride, a.narrow);
            return hipGetLastError();
        }
    }
    if constexpr (LG <= 1 && NP == 10) {
        if (a.short_only == 3 && (LG == 0 || (a.col_begin == 0 && a.col_end == 2 && a.f.stride == 2))) {
            hipLaunchKernelGGL((ibf_count_max_phased_kernel<LG, NP, 3>), grid, dim3(64 * kWavesPerBlock), 0, st, a.f, a.col_begin,
                               a.col_end, a.src, a.n_reads, a.phase, a.out, a.out_read_stride, a.narrow);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((ibf_count_max_phased_kernel<LG, NP, 0>), grid, dim3(64 * kWavesPerBlock), 0, st, a.f, a.col_begin,
                       a.col_end, a.src, a.n_reads, a.phase, a.out, a.out_read_stride, a.narrow);
    return hipGetLastError();
}

template <int NP>
static hipError_t dispatch_phased(const CountLaunch &a, hipStream_t st)
{
    // blocks of one and two words; three- and four-word blocks (stride 4) only in the both-strands build for reads of up to
    // 512 k-mers (short_only 4); wider blocks gain nothing from phases (rb_engine.hip, phase_slice_log2)
    if (a.lg == 2) {
        if constexpr (NP == 10) {
            // blocks of three and four words (stride 4), reads of up to 256 / 384 k-mers, block numbers of 21 bits: the builds with the offsets in LDS
            if (a.multi_reads && a.multi_tiles && a.col_begin == 0 && (a.col_end == 3 || a.col_end == 4) && a.f.stride == 4) {
                if (a.f.n_blocks > kPackMask) return hipErrorInvalidValue;
                dim3 g1(a.n_reads);
#define RB_LAUNCH_MULTI4(INV, TT)                                                                                                       \
    hipLaunchKernelGGL((ibf_count_max_phased_multi_kernel<1, INV, TT, 4>), g1, dim3(64), 0, st, a.f, a.src, a.n_reads, a.phase, a.out, \
                       a.out_read_stride, a.narrow)
                if (a.multi_tiles == 6 && a.multi_inv) RB_LAUNCH_MULTI4(true, 6);
                else if (a.multi_tiles == 6) RB_LAUNCH_MULTI4(false, 6);
                else if (a.multi_inv) RB_LAUNCH_MULTI4(true, 4);
                else RB_LAUNCH_MULTI4(false, 4);
#undef RB_LAUNCH_MULTI4
                return hipGetLastError();
            }
            if ((a.short_only == 4 || a.short_only == 5) && a.col_begin == 0 && (a.col_end == 3 || a.col_end == 4) && a.f.stride == 4) {
                dim3 grid((a.n_reads + kWavesPerBlock - 1) / kWavesPerBlock);
                // short_only 5: every read of the batch has at most 256 k-mers: one round of four tiles per strand; three-word blocks
                // have a build without the fourth column (one 8-byte gather instead of the second 16-byte one, 16 registers less)
                if (a.short_only == 5 && a.col_end == 4)
                    hipLaunchKernelGGL((ibf_count_max_phased_kernel<2, 10, 1, 4>), grid, dim3(64 * kWavesPerBlock), 0, st, a.f, a.col_begin,
                                       a.col_end, a.src, a.n_reads, a.phase, a.out, a.out_read_stride, a.narrow);
                else if (a.short_only == 5)
                    hipLaunchKernelGGL((ibf_count_max_phased_kernel<2, 10, 1, 3>), grid, dim3(64 * kWavesPerBlock), 0, st, a.f, a.col_begin,
                                       a.col_end, a.src, a.n_reads, a.phase, a.out, a.out_read_stride, a.narrow);
#if RB_WIDE3_ROUNDS
                else if (a.col_end == 3)
                    hipLaunchKernelGGL((ibf_count_max_phased_kernel<2, 10, 2, 3>), grid, dim3(64 * kWavesPerBlock), 0, st, a.f, a.col_begin,
                                       a.col_end, a.src, a.n_reads, a.phase, a.out, a.out_read_stride, a.narrow);
#endif
                else  // (four-word blocks; at four waves per SIMD a three-word build had measured 7 % slower, at five it wins)
                    hipLaunchKernelGGL((ibf_count_max_phased_kernel<2, 10, 2, 4>), grid, dim3(64 * kWavesPerBlock), 0, st, a.f, a.col_begin,
                                       a.col_end, a.src, a.n_reads, a.phase, a.out, a.out_read_stride, a.narrow);
                return hipGetLastError();
            }
        }
        return hipErrorInvalidValue;
    }
    return a.lg == 0 ? launch_phased<0, NP>(a, st) : launch_phased<1, NP>(a, st);
}

hipError_t launch_ibf_count_max(const CountLaunch &a, hipStream_t st)
{
    if (a.n_reads == 0) return hipSuccess;
    if (a.phase.n_slices && a.split_waves < 2) {  // planned by the engine for: 3 hash functions, one slice, lg <= 1, wpl 1, n_fused 0
        if (a.f.n_hash != 3 || a.wpl != 1 || a.lg > 2 || a.n_slices != 1 || a.n_fused > 0 || a.phase.n_slices > 32) return hipErrorInvalidValue;
        // slice of a lookup = byte offset >> (shift + 3 + log2 stride): more than one slice needs a power-of-two block stride
        if (a.phase.n_slices > 1 && (a.f.stride & (a.f.stride - 1)) != 0) return hipErrorInvalidValue;
        return a.planes <= 10 ? dispatch_phased<10>(a, st) : dispatch_phased<16>(a, st);
    }
    if (a.split_waves >= 2) {  // latency form (three hash functions only; the engine plans it for those)
        if (a.f.n_hash != 3) return hipErrorInvalidValue;
        return a.planes <= 10 ? launch_split<10>(a, st) : launch_split<16>(a, st);
    }
    if (a.f.n_hash == 3) {
        if (a.planes <= 10) return dispatch_geometry<10, 3>(a, st);
        return dispatch_geometry<16, 3>(a, st);
    }
    return dispatch_geometry<16, 0>(a, st);
}

template <int LG, int NP>
static hipError_t launch_merged_nt(const CountLaunch &a, const MergeMap &map, hipStream_t st)
{
    dim3 grid((a.n_reads + kWavesPerBlock - 1) / kWavesPerBlock);
    if (a.nt)
        hipLaunchKernelGGL((ibf_count_max_merged_kernel<LG, NP, true>), grid, dim3(64 * kWavesPerBlock), 0, st, a.f, map, a.src, a.n_reads,
                           a.out, a.out_read_stride);
    else
        hipLaunchKernelGGL((ibf_count_max_merged_kernel<LG, NP, false>), grid, dim3(64 * kWavesPerBlock), 0, st, a.f, map, a.src, a.n_reads,
                           a.out, a.out_read_stride);
    return hipGetLastError();
}

template <int NP>
static hipError_t launch_merged_lg(const CountLaunch &a, const MergeMap &map, hipStream_t st)
{
    switch (a.lg) {
    case 1: return launch_merged_nt<1, NP>(a, map, st);
    case 2: return launch_merged_nt<2, NP>(a, map, st);
    case 3: return launch_merged_nt<3, NP>(a, map, st);
    case 4: return launch_merged_nt<4, NP>(a, map, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_ibf_count_max_merged(const CountLaunch &a, const MergeMap &map, hipStream_t st)
{
    if (a.n_reads == 0) return hipSuccess;
    if (a.f.n_hash != 3 || a.wpl != 1 || map.n == 0 || map.n > kMaxMerged || map.width == 0 || map.width > (1u << a.lg)) return hipErrorInvalidValue;
    return a.planes <= 10 ? launch_merged_lg<10>(a, map, st) : launch_merged_lg<16>(a, map, st);
}

hipError_t launch_merge_bits(const uint64_t *src, uint32_t s_src, uint32_t width, uint32_t n_bins, uint64_t *dst, uint32_t s_dst, uint32_t dst_bit,
                             uint64_t n_blocks, hipStream_t st)
{
    if (n_blocks == 0 || width == 0) return hipSuccess;
    uint64_t blocks = (n_blocks * width + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(merge_bits_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, src, s_src, width, n_bins, dst, s_dst, dst_bit, n_blocks);
    return hipGetLastError();
}

hipError_t launch_invert_words(const uint64_t *src, uint64_t *dst, uint64_t n_words, hipStream_t st)
{
    if (n_words == 0) return hipSuccess;
    uint64_t blocks = (n_words + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(invert_words_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, src, dst, n_words);
    return hipGetLastError();
}

hipError_t launch_reduce_slices(const uint16_t *part, uint32_t n_slices, uint32_t n_reads, uint16_t *maxcount,
                                uint32_t nf, uint32_t fidx, hipStream_t st)
{
    if (n_reads == 0) return hipSuccess;
    hipLaunchKernelGGL(reduce_slices_kernel, dim3((n_reads + 255) / 256), dim3(256), 0, st, part, n_slices, n_reads,
                       maxcount, nf, fidx);
    return hipGetLastError();
}

hipError_t launch_decide(const DecideParams &P, const uint16_t *maxcount, const uint32_t *lens, const uint8_t *pre_status,
                         uint32_t n_reads, int mode, int32_t *best_target, uint8_t *decision, uint8_t *status,
                         hipStream_t st)
{
    if (n_reads == 0) return hipSuccess;
    if (P.done_flag && n_reads > 256 && !P.done_count) return hipErrorInvalidValue;
    hipLaunchKernelGGL(decide_kernel, dim3((n_reads + 255) / 256), dim3(256), 0, st, P, maxcount, lens, pre_status, n_reads,
                       mode, best_target, decision, status);
    return hipGetLastError();
}

// A micro-batch's way into HBM: the runtime's copy of a few KB is a blit kernel of its own that takes 11 us per call (the driver's
// command under rocprofv3, profiles/r05/default_kernel_stats.csv: __amd_rocclr_copyBuffer, 75 k calls) -- a quarter of a micro-batch's
// service time.  This one reads the pinned (coherent) host block with 16 bytes per lane -- a GPU read of host memory crosses PCIe one
// request per lane -- and is queued like any other kernel of the call.  `units` 16-byte units.
__global__ __launch_bounds__(256) void copy_from_host_kernel(const rb_u32x4 *__restrict__ src, rb_u32x4 *__restrict__ dst, uint32_t units)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < units) dst[i] = __builtin_nontemporal_load(src + i);
}

hipError_t launch_copy_from_host(const void *h_src, void *d_dst, size_t bytes, hipStream_t st)
{
    const uint32_t units = (uint32_t)((bytes + 15) >> 4);
    if (units == 0) return hipSuccess;
    hipLaunchKernelGGL(copy_from_host_kernel, dim3((units + 255) / 256), dim3(256), 0, st, (const rb_u32x4 *)h_src, (rb_u32x4 *)d_dst, units);
    return hipGetLastError();
}

hipError_t launch_chunk_prep(const uint32_t *lens, const uint32_t *ids, uint32_t n_items, uint32_t chunk_start,
                             uint32_t chunk_len, uint32_t *eff_lens, uint8_t *pre_status, hipStream_t st)
{
    if (n_items == 0) return hipSuccess;
    hipLaunchKernelGGL(chunk_prep_kernel, dim3((n_items + 255) / 256), dim3(256), 0, st, lens, ids, n_items, chunk_start,
                       chunk_len, eff_lens, pre_status);
    return hipGetLastError();
}

hipError_t launch_insert(const IbfDev &f, uint64_t *words, const uint8_t *seq, const uint64_t *starts,
                         const uint64_t *ends, const uint64_t *bins, const uint64_t *kmer_prefix,
                         uint32_t n_fragments, uint64_t total_kmers, hipStream_t st)
{
    if (total_kmers == 0) return hipSuccess;
    const uint64_t blocks = (total_kmers + 255) / 256;
    hipLaunchKernelGGL(ibf_insert_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, f, words, seq, starts, ends, bins,
                       kmer_prefix, n_fragments, total_kmers);
    return hipGetLastError();
}

hipError_t launch_restride_blocks(const uint64_t *src, uint32_t s_src, uint64_t *dst, uint32_t s_dst, uint32_t w_copy,
                                  uint64_t n_blocks, hipStream_t st)
{
    if (n_blocks == 0) return hipSuccess;
    uint64_t blocks = (n_blocks * s_dst + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(restride_blocks_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, src, s_src, dst, s_dst, w_copy,
                       n_blocks);
    return hipGetLastError();
}

hipError_t launch_compare_bits(const uint64_t *a, const uint64_t *b, uint64_t n_words, uint64_t *out3, hipStream_t st)
{
    if (n_words == 0) return hipSuccess;
    uint64_t blocks = (n_words + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(compare_bits_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, a, b, n_words,
                       reinterpret_cast<unsigned long long *>(out3));
    return hipGetLastError();
}

hipError_t launch_fill_reads(uint8_t *seqs, uint64_t *offsets, uint32_t *lens, size_t n_reads, uint32_t read_len, uint64_t seed, hipStream_t st)
{
    if (n_reads == 0 || read_len == 0) return hipSuccess;
    uint64_t blocks = ((uint64_t)n_reads * read_len + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(fill_reads_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, seqs, offsets, lens, (uint64_t)n_reads, read_len, seed);
    return hipGetLastError();
}

hipError_t launch_fill_synth(uint64_t *words, uint64_t used_words, uint32_t bin_width, uint32_t stride_words,
                             uint64_t last_mask, uint64_t seed, hipStream_t st)
{
    if (used_words == 0) return hipSuccess;
    uint64_t blocks = (used_words + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(fill_synth_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, words, used_words, bin_width,
                       stride_words, last_mask, seed);
    return hipGetLastError();
}

}  // namespace rb

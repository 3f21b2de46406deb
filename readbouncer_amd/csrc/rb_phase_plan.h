// rb_phase_plan.h -- the planner of the clock-phased gathers (rb_kernels.hip, "phased form"): for a kernel shape, a block width
// and a table size, whether the phased form is used at all, how the table is cut into slices and how long a window lasts.
// Plain C++ (no HIP): rb_engine.hip plans with it, tests/cpp/dump_phase_plan.cpp pins it on a CPU (tests/golden/phase_plan.txt),
// profiles/phase_rule_check.py holds it against measurements at points BETWEEN the fitted ones and fails when it is off.
//
// Everything the planner knows is the ONE table kPhaseRules below: a row per (kernel shape, block width) with named fields.  The
// functions under it only read that table.  Where the numbers come from (K1 ms per 1 M reads on one MI355X,
// profiles/r03/slice_size.txt and window_sweep.txt, sessions 25-61):
//  - the best window length falls with the number of slices n as CYCLE / n: what is constant is the length of a whole cycle
//    over the table, 33-60 us -- the time the resident waves need for one round of their lookups -- so a lookup of any slice
//    waits at most one cycle whatever n is (one-word blocks, 250 bp, 4 MiB slices: n = 3: 1500 ticks, 4: 1000-1200, 6: 850,
//    8: 700, 10-12: 500, 16: 400, 24: 250); after the waves had learnt to serve the slices in the order of the clock
//    (phase_next_slice) the optima are flat: +-100 ticks cost 1-3 %;
//  - a slice of 4 MiB (one XCD's whole L2) is the better cut for larger tables, 2 MiB for smaller ones (`four_mib_from_mib`), and
//    the short-read shapes walk even tables that fit an L2 in pieces of 512 KiB / 1 MiB (`small_full`, session 41: 2 MiB one-word
//    5.95 -> 5.30 ms); 1 MiB slices lose on everything larger;
//  - the windows were fitted with reads that FILL their kernel shape (`fit_kmers`: 238 k-mers in four tiles, 348 in six, 488 in
//    two rounds of four); a wave with fewer k-mers is through its round sooner and the best cycle shrinks with it (session 38:
//    150 bp reads, one-word 20 MiB: best window 700 ticks against 850-1000 at 250 bp) -- phase_fill(), 0.5 ... 1, scales the
//    curves, moves the 2 / 4 MiB switch and narrows the range of table sizes;
//  - beyond `max_bytes` the plain kernel (at the fabric-request wall from about 64 MiB on) is as fast or faster, below
//    `min_bytes` the phased form has nothing to win; blocks of five and more words gain nothing from phases at any size.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>

namespace rbplan {

// Kernel shapes of ibf_count_max_phased_kernel (rb_kernels.hip): how many 64-k-mer tiles per strand a wave holds per round of windows
enum class PhaseShape : int {
    General = 0,         // per-strand tiles of the general build (reads of more than 512 k-mers; 16 planes)
    FourTiles = 1,       // both strands in ONE round of four tiles: reads of up to 256 k-mers (the reference's default 250 bp chunk)
    Rounds = 2,          // rounds of three (one-word) or four (two-word) tiles: up to 512 k-mers
    SixTiles = 3,        // ONE round of six tiles: 257-384 k-mers (360 bp reads)
    WideRounds = 4,      // three- and four-word blocks held by one lane, rounds of three tiles (up to 512 k-mers), four-word build
    WideFourTiles = 5,   // ... one round of four tiles (up to 256 k-mers), four-word build
    Wide3FourTiles = 6,  // ... the three-word build of that (five waves per SIMD)
    Wide3Rounds = 7,     // ... the three-word build of the rounds of three tiles
};
constexpr int kPhaseShapes = 8;

constexpr uint64_t operator""_KiB(unsigned long long v) { return v << 10; }
constexpr uint64_t operator""_MiB(unsigned long long v) { return v << 20; }
constexpr uint64_t kNever = ~0ull;

// "below this table size (MiB): slices of 2^slice_log2 bytes"; bound 0 ends a list
struct SliceStep {
    double below_mib;
    uint32_t slice_log2;
};
// "reads that fill the shape at least this much: phased from this table size on"; evaluated in order, the last row has fill 0
struct MinStep {
    double fill_at_least;
    uint64_t bytes;
};
// Window length in 10 ns ticks for one slice size.  Either up to three constants by the number of slices (n <= n0: t0,
// n <= n1: t1, else t2; kind Steps) or base + cycle / n (kind Curve: scaled by phase_fill; kind IntCurve: integer division, not
// scaled -- the wide rounds were fitted that way), never below `floor`.
struct Window {
    enum Kind { Steps, Curve, IntCurve } kind;
    uint32_t n0, n1;
    double t0, t1, t2;
    double base, cycle, floor;
    // Curve only, reads that fill the shape (fill >= 0.9): unscaled, and never below full_floor_small up to full_floor_n slices,
    // full_floor_large beyond (two-word 250 bp: below its optimum the times jump -- 48 MiB: 12.6 ms at 600 ticks, 17.8 at 500)
    uint32_t full_floor_n;
    double full_floor_small, full_floor_large;
};
constexpr Window constant(double t) { return Window{Window::Steps, ~0u, ~0u, t, t, t, 0, 0, 0, 0, 0, 0}; }
constexpr Window steps(uint32_t n0, double t0, uint32_t n1, double t1, double t2) { return Window{Window::Steps, n0, n1, t0, t1, t2, 0, 0, 0, 0, 0, 0}; }
constexpr Window curve(double base, double cycle, double floor = 0.0) { return Window{Window::Curve, 0, 0, 0, 0, 0, base, cycle, floor, 0, 0, 0}; }
constexpr Window curve_full_floor(double base, double cycle, double floor, uint32_t n, double small, double large)
{
    return Window{Window::Curve, 0, 0, 0, 0, 0, base, cycle, floor, n, small, large};
}
constexpr Window int_curve(double cycle, double floor) { return Window{Window::IntCurve, 0, 0, 0, 0, 0, 0, cycle, floor, 0, 0, 0}; }

struct PhaseRule {
    const char *name;
    PhaseShape shape;
    int lg;               // log2 lanes per block: 0 = one-word blocks, 1 = two-word blocks, 2 = stride-4 blocks (three / four words) in one lane
    double fit_kmers;     // k-mers per read the windows were fitted with; 0: the work of a round does not depend on the read length
    // ---- which tables
    MinStep min_bytes[3];
    uint64_t max_bytes;   // (x fill when the reads fill less than 80 % of the shape: the plain kernel catches up sooner)
    // ---- slice size
    SliceStep small_full[2];     // reads that fill the shape (>= 0.9): tables below these sizes get 512 KiB / 1 MiB slices
    SliceStep small_partial[2];  // reads that leave it partly empty: larger pieces (session 57)
    double four_mib_from_mib;    // 2 MiB slices below this table size, 4 MiB from it on
    bool four_mib_scaled;        // ... x fill, and "below" is strict; false: unscaled, and the bound itself still takes 2 MiB slices (wide rounds)
    // ---- window length by slice size: 512 KiB, 1 MiB, 2 MiB, 4 MiB and more
    Window window[4];
    double partial_2mib_ticks;   // 2 MiB slices, at most four of them, reads that leave the shape partly empty (0: no such rule)
};

// clang-format off
constexpr PhaseRule kPhaseRules[] = {
    // ------------------------------------------------------------------------------------------------ one-word blocks (<= 64 bins)
    {"general, one-word", PhaseShape::General, 0, 0.0,
     {{0.0, 6_MiB}}, 64_MiB,                       // 500 bp reads: 6 MiB 17.7 -> 16.1 ms; 64 MiB 34.4 against 49.3 plain
     {}, {}, 10.0, true,
     {curve(0, 2400), curve(0, 2400), curve(0, 2400), curve(200, 2500)}, 0},
    {"four tiles, one-word", PhaseShape::FourTiles, 0, 238.0,
     {{0.9, 1280_KiB}, {0.0, 2_MiB}}, 128_MiB,     // 127 MiB: 22.1 against 25.2 ms plain
     {{3.5, 19}, {7.0, 20}}, {{3.5, 20}, {7.5, 21}}, 10.0, true,
     {constant(250), constant(325), curve(0, 4600), curve(150, 5500)}, 450},   // (eight waves per SIMD since session 55: 150 + 5500 / n)
    {"rounds of three tiles, one-word", PhaseShape::Rounds, 0, 488.0,
     {{0.0, 5_MiB}}, 64_MiB,
     {}, {}, 10.0, true,
     {curve(0, 2400), curve(0, 2400), curve(0, 2400), curve(300, 2400)}, 0},   // (session 61)
    {"six tiles, one-word", PhaseShape::SixTiles, 0, 348.0,
     {{0.9, 1280_KiB}, {0.0, 2_MiB}}, 128_MiB,     // 127 MiB: 32.5 against 36.7 ms plain
     {{3.5, 19}, {7.0, 20}}, {{3.5, 20}, {7.5, 21}}, 17.0, true,
     {constant(250), constant(400), steps(7, 500, ~0u, 400, 400), curve(150, 6800)}, 600},
    // ------------------------------------------------------------------------------------------------ two-word blocks (65-128 bins)
    {"general, two-word", PhaseShape::General, 1, 0.0,
     {{0.0, 6_MiB}}, 48_MiB,                       // 1000 bp: 48 MiB 85.9 against 97.4 plain
     {}, {}, 10.0, true,
     {curve(450, 0), curve(450, 0), curve(450, 0), curve(100, 2000)}, 0},
    {"four tiles, two-word", PhaseShape::FourTiles, 1, 238.0,
     {{0.9, 1280_KiB}, {0.75, 3_MiB}, {0.0, 4608_KiB}}, 96_MiB,   // 96 MiB: 20.2 against 24.8 ms plain
     {{2.5, 19}, {7.0, 20}}, {{7.5, 21}}, 17.0, true,           // (with 2 MiB slices the optimum of 18-24 MiB tables is a narrow dip, with 4 MiB a flat region)
     {constant(400), constant(400), steps(9, 500, ~0u, 400, 400), curve_full_floor(100, 5000, 325, 12, 600, 500)}, 450},
    {"rounds of four tiles, two-word", PhaseShape::Rounds, 1, 488.0,
     {{0.0, 6_MiB}}, 48_MiB,
     {}, {}, 10.0, true,
     {curve(450, 0), curve(450, 0), curve(450, 0), curve(200, 2500)}, 0},
    {"six tiles, two-word", PhaseShape::SixTiles, 1, 348.0,
     {{0.9, 1280_KiB}, {0.75, 3_MiB}, {0.0, 4608_KiB}}, 80_MiB,   // 64 MiB: 22.2 against 35.1 ms plain; 80 MiB (300 bp): 25.3-27.5 against 29.7
                                                                  // (r04 guard run); at 96 MiB the optimum is narrow
     {{2.5, 19}, {7.0, 20}}, {{7.5, 21}}, 18.5, true,
     {constant(400), constant(400), curve(400, 0, 400), curve(150, 4400, 400)}, 600},   // (below 400 ticks the times get erratic)
    // ------------------------------------------------------------------------------- three- and four-word blocks (129-256 bins), one lane per block
    {"wide, rounds of three tiles (four-word build)", PhaseShape::WideRounds, 2, 348.0,
     {{0.0, 6_MiB}}, 48_MiB,                       // 40 MiB: 25.4 against 33.6 ms plain (360 bp); 64 MiB: even
     {}, {}, 13.0, false,
     {constant(325), constant(325), constant(325), int_curve(2400, 400)}, 0},   // (16 MiB: 600, 24 MiB: 400 -- session 50)
    {"wide, four tiles (four-word build)", PhaseShape::WideFourTiles, 2, 238.0,
     {{0.75, 4608_KiB}, {0.0, kNever}}, 48_MiB,    // 40 MiB: 17.2 against 22.8 ms plain (250 bp); 200 bp reads, 36 MiB: 15.1 against 17.6 (r04 guard run)
     {}, {}, 12.0, true,
     {constant(400), constant(400), constant(400), constant(500)}, 0},
    {"wide, four tiles (three-word build)", PhaseShape::Wide3FourTiles, 2, 238.0,
     {{0.75, 3_MiB}, {0.55, 12_MiB}, {0.0, kNever}}, 48_MiB,   // 4 MiB table: 7.4 ms without a clock, 6.3 in two slices of 2 MiB; 150 bp reads pay on
                                                   // larger tables only (16 MiB: 9.4 against 10.3 ms plain, r04 wide-grid guard run); 200 bp reads, 13 MiB: 10.2 in
                                                   // four slices of 4 MiB against 13.0 plain and 12.0 in slices of 2 MiB (r04 guard run: the switch scales)
     {}, {}, 14.0, true,
     {constant(500), constant(500), constant(500), steps(4, 850, 8, 600, 500)}, 0},   // (five waves per SIMD: longer windows, session 53)
    {"wide, rounds of three tiles (three-word build)", PhaseShape::Wide3Rounds, 2, 348.0,
     {{0.0, 6_MiB}}, 48_MiB,
     {}, {}, 13.0, false,
     {constant(400), constant(400), constant(400), int_curve(3400, 400)}, 0},   // (session 59)
};
// clang-format on

// the row of a shape and block width (shapes 0-3 exist for one- and two-word blocks, the wide shapes for stride-4 blocks)
constexpr bool phase_shape_is_wide(PhaseShape shape) { return (int)shape >= (int)PhaseShape::WideRounds; }
constexpr const PhaseRule &phase_rule(PhaseShape shape, int lg)
{
    for (const PhaseRule &r : kPhaseRules)
        if (r.shape == shape && (r.lg == lg || phase_shape_is_wide(shape))) return r;
    for (const PhaseRule &r : kPhaseRules)
        if (r.shape == shape) return r;  // (a block width the shape has no row for: its first row)
    return kPhaseRules[0];
}

// how full the kernel shape is with reads of `kmers` k-mers: factor on the cycle, 0.5 ... 1
inline double phase_fill(PhaseShape shape, uint32_t kmers)
{
    const double fit = phase_rule(shape, phase_shape_is_wide(shape) ? 2 : 0).fit_kmers;
    if (fit == 0.0) return 1.0;
    return std::min(1.0, std::max(0.5, (double)kmers / fit));
}

inline uint32_t phase_slice_log2(PhaseShape shape, int lg, uint64_t table_bytes, uint32_t kmers)
{
    const PhaseRule &r = phase_rule(shape, lg);
    const double mib = (double)table_bytes / 1048576.0;
    const double fill = phase_fill(shape, kmers);
    if (!r.four_mib_scaled) return mib <= r.four_mib_from_mib ? 21 : 22;
    if (fill >= 0.9) {
        for (const SliceStep &s : r.small_full)
            if (s.below_mib > 0.0 && mib < s.below_mib) return s.slice_log2;
    } else {
        for (const SliceStep &s : r.small_partial)
            if (s.below_mib > 0.0 && mib < s.below_mib) return s.slice_log2;
    }
    return mib < r.four_mib_from_mib * fill ? 21 : 22;
}

// Slices of EQUAL length, not of 2^n bytes, for the three- and four-word one-lane builds when their rule asks for 4 MiB slices AND cutting the
// table at 4 MiB is wasteful (at least two slices more than slices of up to 4.75 MiB need): the TA spends 16 cycles on every predicated
// wave-level load whatever share of its lanes lies in the slice of the moment, and every slice means one more pass over all of a
// wave's loads -- 74 % of that kernel's cycles at ten slices (profiles/r04/pmc_units_readme.txt).  README shape (37.7 MiB), M reads/s at
// the best window: 250 bp 10 slices of 4 MiB (the last one 1.7 MiB) 59.5, 9 slices 63.1, 8 slices of 4.7 MiB 66.5, 7 slices 66.0,
// 6 slices 63.7; 360 bp 37.6 / 40.6 / 43.4 / 43.8 / 43.2 (profiles/r04/slice_count_sweep.txt).  Where 4 MiB slices fill the table
// exactly (28, 32 MiB) one slice less buys nothing: the window has to grow with the lookups per slice (its cliff sits near 4 100 / n
// ticks at 250 bp), profiles/r04/equal_slices_check*.txt.  Returns the number of slices (0: keep the slices of 2^slice_log2 bytes).
constexpr double kWideEqualSliceMiB = 4.75;
inline uint32_t phase_equal_slices(PhaseShape shape, uint32_t slice_log2, uint64_t table_bytes)
{
    if (slice_log2 != 22 || !phase_shape_is_wide(shape)) return 0;  // (the four- and the three-word one-lane builds)
    const uint32_t pow2 = (uint32_t)((table_bytes + (4_MiB - 1)) / 4_MiB);
    const uint32_t n = (uint32_t)std::ceil((double)table_bytes / (kWideEqualSliceMiB * 1048576.0));
    return n >= 2 && n + 2 <= pow2 ? n : 0;
}

inline uint64_t phase_window_ticks(PhaseShape shape, int lg, uint32_t slice_log2, uint32_t n_slices, uint32_t kmers)
{
    const PhaseRule &r = phase_rule(shape, lg);
    const double fill = phase_fill(shape, kmers);
    const int cls = slice_log2 <= 19 ? 0 : slice_log2 == 20 ? 1 : slice_log2 == 21 ? 2 : 3;
    if (cls == 2 && r.partial_2mib_ticks > 0.0 && n_slices <= 4 && fill < 0.9) return (uint64_t)r.partial_2mib_ticks;
    const Window &w = r.window[cls];
    const uint32_t n = std::max(n_slices, 1u);
    switch (w.kind) {
    case Window::Steps: return (uint64_t)(n_slices <= w.n0 ? w.t0 : n_slices <= w.n1 ? w.t1 : w.t2);
    case Window::IntCurve: return std::max<uint64_t>((uint64_t)w.floor, (uint32_t)w.cycle / n);
    default: break;
    }
    if (w.full_floor_n && fill >= 0.9) return (uint64_t)std::max(w.base + w.cycle / n, n_slices <= w.full_floor_n ? w.full_floor_small : w.full_floor_large);
    const double t = (w.base + w.cycle / n) * fill;
    // (reads that leave the shape partly empty: the floor of the few-slices case shrinks with the cycle -- round 4's guard run found the
    // scaled curve alone in a bad spot: two-word 200 bp 45 MiB 14.7 ms at 408 ticks against 11.0 at 489)
    if (w.full_floor_n && n_slices <= w.full_floor_n) return (uint64_t)std::max(t, std::max(w.floor, w.full_floor_small * fill));
    return (uint64_t)std::max(t, w.floor);
}

// the window for such slices: the shape's rule for that many slices, and for the rounds of three tiles not below 3 360 / n ticks (eight
// slices of 4.7 MiB, 360 bp: 360 ticks 31.9 ms, 400 ticks 23.0, 440 ticks 24.5 -- the rule's 400 would sit right on the cliff)
inline uint64_t phase_equal_slices_ticks(PhaseShape shape, int lg, uint32_t n_slices, uint32_t kmers)
{
    const uint64_t rule = phase_window_ticks(shape, lg, 22, n_slices, kmers);
    const uint32_t n = std::max(n_slices, 1u);
    // the three-word builds (five waves per SIMD: longer passes): 250 bp, 41.5 MiB in nine slices: 500 ticks 19.8 ms, 550 ticks 15.2;
    // 360 bp, 37.7 MiB in eight slices: 382 ticks 27.3 ms, 425 ticks 23.1, 467 ticks 22.4 (profiles/r04/equal_slices_three_word_builds.txt)
    const uint32_t floor_ticks = shape == PhaseShape::WideRounds ? 3360u / n : shape == PhaseShape::Wide3FourTiles ? 5200u / n
                               : shape == PhaseShape::Wide3Rounds ? 3600u / n : 0u;
    return std::max<uint64_t>(rule, floor_ticks);
}


// One-word tables of 50 MiB and more (a 64-bin filter of a 35-80 Mbp genome): equal-length slices LONGER than an L2, fewer of them, a
// longer cycle.  Every slice is one more pass of a wave over all its predicated loads; from ~50 MiB on that costs more than the L2 hits a
// 4 MiB slice buys (profiles/r05/one_word_equal_slices*.txt, 250 bp, K1 ms per 1 M reads, rule of 4 MiB slices -> best equal slices):
// 56 MiB 13.6 -> 12.2 (11 slices), 64: 14.0 -> 13.1 (13), 80: 16.0 -> 14.1 (13), 96: 17.7 -> 15.1 (15), 112: 19.5 -> 15.9 (14),
// 127: 21.4 -> 16.8 (16); 360 bp: 64 MiB 21.4 -> 19.6, 96: 26.1 -> 23.1, 127: 32.1 -> 25.6; up to 48 MiB within 2 %.  The best slice
// length grows with the table (4.9 MiB at 56, 6.2 at 80, 8 at 112-127): n = round(MiB / (2.5 + 0.044 MiB)); the best cycle (window x
// slices) is 8 000-9 000 ticks at 56-64 MiB and 12 000 at 112-127: 5 600 + 54 per MiB, x 1.24 for the six-tile build, and -- like every
// curve of this file -- shorter for reads that leave the shape partly empty.  The optima are narrow (a window 10 % short falls off the
// cliff: 80 MiB, 13 slices: 692 ticks 15.8 ms, 769 ticks 14.1); rb_engine_calibrate refines them per device.
constexpr uint64_t kOneWordEqualFrom = 50_MiB;  // (48 MiB: within 2 % either way; 52 MiB: 13.4 -> 11.6-12.1 ms)
inline uint32_t phase_equal_slices_one_word(PhaseShape shape, int lg, uint32_t slice_log2, uint64_t table_bytes)
{
    if (lg != 0 || slice_log2 != 22 || table_bytes < kOneWordEqualFrom) return 0;
    if (shape != PhaseShape::FourTiles && shape != PhaseShape::SixTiles) return 0;
    const double mib = (double)table_bytes / 1048576.0;
    const uint32_t n = (uint32_t)std::lround(mib / (2.5 + 0.044 * mib));
    const uint32_t pow2 = (uint32_t)((table_bytes + (4_MiB - 1)) / 4_MiB);
    return n >= 2 && n < pow2 ? n : 0;
}

inline uint64_t phase_equal_slices_one_word_ticks(PhaseShape shape, uint32_t n_slices, uint64_t table_bytes, uint32_t kmers)
{
    const double mib = (double)table_bytes / 1048576.0;
    const double fill = phase_fill(shape, kmers);
    const double cycle = (5600.0 + 54.0 * mib) * (shape == PhaseShape::SixTiles ? 1.24 : 1.0) * (0.5 + 0.5 * (fill < 1.0 ? fill : 1.0));
    return (uint64_t)(cycle / (double)std::max(n_slices, 1u));
}

// The builds of round 6 that keep a read's block numbers in LDS (rb_kernels.hip, ibf_count_max_phased_multi_kernel: two-word tables of
// up to 32 MiB): more waves per SIMD than the register builds, so a CU's waves hold more lookups and a window has more to serve.  Where
// that moves the optimum, the window is given here; elsewhere the shape's rule stands (`rule_ticks`).  K1 ms per 1 M reads,
// profiles/r06/multi/ (s13_six_tiles_sweep.txt, s14_guard_first_run.txt, guard_two_word.txt):
//  - four tiles (eight waves per SIMD; the register build had seven): the rule's windows hold -- 18.9 MiB in five slices of 4 MiB, 250 bp:
//    990 ticks 8.72, 1 100: 8.21, 1 155: 8.52; 28 MiB, seven slices: 814 ticks 10.08, 976: 9.70 -- except for up to five slices of 2 MiB:
//    8 MiB, 250 bp: 500 ticks 7.60, 550: 6.92, 600: 6.83, 675: 6.78, 750: 6.97;
//  - six tiles (seven waves where the register build had four): a cliff on the short side -- 19 MiB, five slices of 4 MiB, 360 bp: 988 ticks
//    13.85, 1 236: 12.81, 1 359: 12.94; 28 MiB, seven slices: 840 ticks 18.18, 934: 13.46, 1 027: 14.06; 30 MiB, eight slices, 300 bp: 695
//    ticks 17.45, 764: 12.21, 834: 12.97 -- a whole cycle of 6 700 ticks, NOT scaled with the reads' fill (a wave walks all twelve slots of
//    a lane whatever the read's length); slices of 2 MiB: 13 MiB, seven slices, 300 bp: 480 ticks 11.92, 576: 9.58, 648: 9.90; 8 MiB, four
//    slices, 360 bp: 576 ticks 10.52, 648: 9.91, 720: 9.27.
inline uint64_t phase_multi_window_ticks(PhaseShape shape, uint32_t slice_log2, uint32_t n_slices, uint32_t kmers, uint64_t rule_ticks)
{
    const uint32_t n = std::max(n_slices, 1u);
    if (shape == PhaseShape::SixTiles) {
        if (slice_log2 >= 22) return std::max<uint64_t>(rule_ticks, 6700u / n);
        if (slice_log2 == 21) return std::max<uint64_t>(rule_ticks, 350u + 1700u / n);
        return rule_ticks + rule_ticks / 2;
    }
    if (shape == PhaseShape::FourTiles && slice_log2 == 21 && n <= 5) {
        const double fill = phase_fill(shape, kmers);
        return std::max<uint64_t>(rule_ticks, (uint64_t)(620.0 * (0.5 + 0.5 * fill)));
    }
    // blocks of three and four words (five waves per SIMD where the register build had four; reads of 257-384 k-mers in ONE round of six
    // tiles where it takes two rounds of three).  Four tiles, slices of 4 MiB or equal-length ones, best window by slices: 4 slices
    // (13 MiB, 200 bp) 744 ticks 10.13 ms, 868: 11.20; 6 (24 MiB, 250 bp) 578: 12.27, 660: 11.83, 743: 12.68; 8 (README shape, 37.7 MiB,
    // 250 bp) 550: 15.54, 600: 13.07, 675: 13.93 (register build: 14.82 at 500); 9 (36 MiB, 200 bp) 495: 12.70, 551: 13.03 -- 400 +
    // 1 700 / n.  Six tiles: a cliff near 650-700 ticks whatever the number of slices (a wave's pass over twelve slots of three
    // lookups of two gathers) -- 24 MiB, 6 slices, 360 bp: 607 ticks 22.06, 709: 15.89, 810: 17.92; 37.7 MiB, 8 slices: 630: 27.52, 735:
    // 19.84, 840: 20.78 (register build: 22.18 at 378); 12 MiB in 6 slices of 2 MiB: 585: 15.56, 682: 15.10 -- never below 750.
    if (phase_shape_is_wide(shape)) {
        const bool six = shape == PhaseShape::WideRounds || shape == PhaseShape::Wide3Rounds;
        if (six) return slice_log2 >= 21 ? std::max<uint64_t>(rule_ticks, 750u) : rule_ticks + rule_ticks / 5;
        return slice_log2 >= 22 ? (uint64_t)(400u + 1700u / n) : rule_ticks + rule_ticks / 5;
    }
    return rule_ticks;
}

// Round 6, after the LDS-offset builds: where the two-word rule asks for 4 MiB slices, slices of EQUAL length SHORTER than an L2 (2.4-3.2 MiB)
// serve the LDS-offset builds better.  A 4 MiB slice is a whole L2: the lines of the window before are evicted as the next slice arrives,
// waves that are a little late miss, and the table's last, shorter slice runs under the same window as the others.  With ~2.5 MiB slices
// the window shrinks to what a slice's lookups need (about 500 ticks whatever the table) and the cycle with it.  K1 ms per 1 M reads, single
// 128-bin filters (AND form), rule of 4 MiB slices -> equal slices at the best window (profiles/r06/multi/equal_slices_fit_two_word.txt):
//   four tiles, 250 bp: 18.9 MiB 8.41 -> 7.66 (8 slices, 500 ticks); 22: 9.06 -> 7.97 (9, 511); 26: 9.86 -> 8.47 (9, 511); 31: 10.21 -> 8.90 (10, 540);
//               200 bp: 16 MiB 6.85 -> 6.25 (6-7); 18.9: 7.54 -> 6.64 (7, 485); 22: 8.03 -> 6.92 (7-8); 26: 8.45 -> 7.38 (9, 477); 31: 9.83 -> 8.10 (10, 460);
//   six tiles,  360 bp: 18.9 MiB 12.90 -> 10.95 (7 slices, 687 ticks); 22: 13.33 -> 11.53 (8, 650); 26: 14.14 -> 12.24 (9, 621); 31: 13.53 -> 13.11 (10, 598);
//               300 bp: 16 MiB 11.10 -> 9.16 (5, 806); 18.9: 11.96 -> 9.71 (6, 736) / 9.93 (7, 631); 22: 12.51 -> 10.40 (7, 687); 26: 12.70 -> 11.11 (8, 650);
//   the merged OR form (bench legs deplete_target / targets3, 18.9 MiB): 250 bp 8.22 -> 7.58 (8 slices, 507 ticks); 360 bp 12.7 -> 10.93 (7, 727).
// Every curve has a cliff on the short side (a window shorter than about 1.4 x the slice's refill, or -- six tiles -- than a wave's pass
// over its twelve slots): four tiles near 165 ticks per MiB of slice, 5-15 % up; six tiles 30-40 % up within 8 % of the optimum
// (18.9 MiB, 360 bp, 7 slices: 631 ticks 12.57, 687: 10.95; the merged form: 669: 12.15, 727: 10.93) -- the rule stays 5-8 % above it.
// Tables the rule cuts into 2 MiB slices (up to 13-16 MiB) are within 2-4 % of the best equal cut and keep it.
// The one-word LDS-offset builds (tables of up to 32 MiB: 22-bit block numbers) follow the same curves where their rule asks for 4 MiB slices
// (equal_slices_fit_one_word.txt: 13 MiB, 250 bp: 8.02 -> 7.3 in six slices; 16 MiB: 8.27 -> 7.5 in seven, 300 bp: 10.59 -> 9.3-9.7 in six; the
// 19.8 MB filter of the reference's test data: 8.42 -> 7.78 in eight, 360 bp 12.40 -> 10.69 in seven; guard_one_word_22bit.txt: seven points of 19.8-31 MiB
// within tolerance) and take the same cut; the one-word REGISTER builds (larger tables) gain 0-6 % at their best cut and keep the 4 MiB slices.
inline uint32_t phase_multi_equal_slices(PhaseShape shape, int lg, uint32_t slice_log2, uint64_t table_bytes, uint32_t kmers)
{
    if (lg > 1 || slice_log2 != 22) return 0;
    const double mib = (double)table_bytes / 1048576.0, fill = phase_fill(shape, kmers);
    double slice_mib;
    if (shape == PhaseShape::FourTiles) slice_mib = (1.45 + 0.048 * mib) * (1.5 - 0.5 * fill);  // (shorter reads: fewer, longer slices)
    else if (shape == PhaseShape::SixTiles) slice_mib = 2.5 + 0.02 * mib;
    else return 0;
    const uint32_t n = (uint32_t)(mib / slice_mib + 0.5);
    return n >= 2 && n <= 32 ? n : 0;
}
inline uint64_t phase_multi_equal_ticks(PhaseShape shape, uint32_t n_slices, uint64_t table_bytes, uint32_t kmers)
{
    const double mib = (double)table_bytes / 1048576.0, fill = phase_fill(shape, kmers);
    const uint32_t n = std::max(n_slices, 1u);
    if (shape == PhaseShape::SixTiles) return (uint64_t)((2350.0 + 130.0 * mib) * 1.08 * (0.75 + 0.25 * fill) / n);  // (a cycle, over the slices)
    return (uint64_t)(1.05 * std::max(480.0, 168.0 * mib / n) * (0.85 + 0.15 * fill));
}

// from which table size on the phased form pays ...
inline uint64_t phase_shape_min_bytes(PhaseShape shape, int lg, double fill)
{
    const PhaseRule &r = phase_rule(shape, lg);
    for (const MinStep &m : r.min_bytes)
        if (fill >= m.fill_at_least) return m.bytes;
    return kNever;
}

// ... and up to which (profiles/r03/slice_size.txt; the plain kernel sits at the fabric-request wall from about 64 MiB on)
inline uint64_t phase_shape_max_bytes(PhaseShape shape, int lg) { return phase_rule(shape, lg).max_bytes; }

// A read needs at least one whole cycle over its table's slices, and a cycle is not shorter than the refill of the table: small
// batches of a large table are served faster by the plain kernel (profiles/r03/phased_batch_size.txt, second part: 64 MiB
// one-word, 250 bp, 2 049 reads per call 78 us phased against 65 plain, 4 096: 105 / 116; 120 MiB: even at 16 384 reads, 1.11-1.17 x
// at 65 536).  Tables of up to 32 MiB pay from the first batch above the latency kernel's (rb_engine::phase_min_reads).
inline size_t phase_min_reads_for(uint64_t table_bytes)
{
    if (table_bytes > (64ull << 20)) return 32768;
    if (table_bytes > (32ull << 20)) return 4096;
    return 0;
}

}  // namespace rbplan

// dump_phase_plan.cpp -- the planner of the clock-phased gathers (readbouncer_amd/csrc/rb_phase_plan.h) over a grid of kernel
// shapes, block widths, table sizes and read lengths, on a CPU.  `dump_phase_plan` prints one FNV-1a digest per (shape, lg) over
// every planner output of the grid plus a few rows in full; tests/test_phase_plan.py compares them with
// tests/golden/phase_plan.txt, so that a change of the rules is a visible change of a fixture (the rules
// were fitted on one box over 62 sessions and nothing guarded them).  `dump_phase_plan full` prints every row.
#include <cinttypes>
#include <cstdio>
#include <cstring>

#include "../../readbouncer_amd/csrc/rb_phase_plan.h"

using namespace rbplan;

int main(int argc, char **argv)
{
    const bool full = argc > 1 && !std::strcmp(argv[1], "full");
    static const double mibs[] = {1, 1.25, 1.5, 2, 2.5, 3, 3.5, 4, 4.5, 5, 6, 7, 7.5, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 18.5, 19, 20, 22, 24,
                                  28, 32, 40, 48, 56, 64, 80, 96, 112, 127, 128, 129};
    static const uint32_t kmers[] = {50, 100, 138, 150, 188, 200, 214, 215, 238, 256, 257, 288, 300, 313, 314, 348, 384, 385, 418, 440, 488, 512, 513, 700, 988};
    for (int shape_no = 0; shape_no < kPhaseShapes; ++shape_no) {
        const PhaseShape shape = (PhaseShape)shape_no;
        for (int lg = (shape_no >= 4 ? 2 : 0); lg <= (shape_no >= 4 ? 2 : 1); ++lg) {
            uint64_t h = 1469598103934665603ull;
            auto mix = [&](uint64_t v) { for (int i = 0; i < 8; ++i) { h ^= (v >> (8 * i)) & 0xFF; h *= 1099511628211ull; } };
            for (double mib : mibs) {
                const uint64_t bytes = (uint64_t)(mib * 1048576.0);
                for (uint32_t km : kmers) {
                    const double fill = phase_fill(shape, km);
                    const uint32_t sl = phase_slice_log2(shape, lg, bytes, km);
                    const uint64_t mn = phase_shape_min_bytes(shape, lg, fill), mx = phase_shape_max_bytes(shape, lg);
                    mix((uint64_t)(fill * 1e9)); mix(sl); mix(mn); mix(mx); mix(phase_min_reads_for(bytes));
                    if (full) std::printf("shape %d lg %d mib %.2f kmers %u fill %.6f slice_log2 %u min %" PRIu64 " max %" PRIu64 " min_reads %zu\n", shape_no, lg, mib, km, fill, sl, mn, mx, phase_min_reads_for(bytes));
                    for (uint32_t s2 = 19; s2 <= 22; ++s2)
                        for (uint32_t n = 1; n <= 32; ++n) {
                            const uint64_t t = phase_window_ticks(shape, lg, s2, n, km);
                            mix(t);
                            if (full && (mib == 16 || mib == 64)) std::printf("  ticks shape %d lg %d kmers %u slice_log2 %u n %u -> %" PRIu64 "\n", shape_no, lg, km, s2, n, t);
                        }
                }
            }
            std::printf("digest shape %d lg %d %016" PRIx64 "  (%s)\n", shape_no, lg, h, phase_rule(shape, lg).name);
        }
    }
    // a few rows in full: the planner's answer for the shapes the README benchmark and BASELINE's read length produce
    struct Row { int shape, lg; double mib; uint32_t km; } rows[] = {{1, 0, 10.4, 238}, {1, 1, 18.9, 238}, {3, 0, 10.4, 348}, {3, 1, 18.9, 348}, {2, 0, 20, 488},
                                                                    {0, 0, 20, 988}, {5, 2, 24, 238}, {4, 2, 24, 348}, {6, 2, 24, 238}, {7, 2, 24, 348}, {1, 0, 64, 238}, {3, 0, 100, 348}};
    for (const Row &r : rows) {
        const uint64_t bytes = (uint64_t)(r.mib * 1048576.0);
        const PhaseShape sh = (PhaseShape)r.shape;
        const uint32_t sl = phase_slice_log2(sh, r.lg, bytes, r.km);
        const uint32_t n = (uint32_t)((bytes + (1ull << sl) - 1) >> sl);
        std::printf("row shape %d lg %d mib %.1f kmers %u: slice_log2 %u n_slices %u ticks %" PRIu64 " range [%" PRIu64 ", %" PRIu64 "]\n", r.shape, r.lg, r.mib, r.km, sl, n,
                    phase_window_ticks(sh, r.lg, sl, n, r.km), phase_shape_min_bytes(sh, r.lg, phase_fill(sh, r.km)), phase_shape_max_bytes(sh, r.lg));
    }
    return 0;
}

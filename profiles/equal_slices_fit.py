#!/usr/bin/env python3
"""Round 6, after session 20: slices of EQUAL length, shorter than an L2, for the two-word LDS-offset builds (ibf_count_max_phased_multi_kernel
<1, INV, 4 | 6, 2>).  The 18.9 MiB merged table of bench.py's deplete_target / targets3 legs runs 8.22 ms per 1 M reads of 250 bp in five
slices of 4 MiB (the last one 2.9 MiB, under the same window as the others) and 7.6-7.9 in seven or eight equal ones; 360 bp: 12.7 -> 10.9.
This script measures the grid a planner rule needs: single two-word filters (128 bins, AND form) of several sizes x read lengths x slice
counts x cycle lengths (cycle = slices x window, in ticks of 10 ns).  One engine per slice count (RB_PHASE_N_SLICES is read when an engine
is made; RB_TUNING_ENV=1), the window forced through rb_engine_set_phased; every setting's raw maxima equal the first one's.

  RB_TUNING_ENV=1 python3 profiles/equal_slices_fit.py [--sizes 5,9,13,18.9,24,31] [--lens 200,250,300,360] [--targets 2.0,2.4,2.75,3.2]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RB_TUNING_ENV"] = "1"
from readbouncer_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="5,7,9,11,13,16,18.9,22,26,31")
ap.add_argument("--lens", default="200,250,300,360")
ap.add_argument("--targets", default="2.0,2.4,2.75,3.2", help="slice lengths aimed for, MiB (the slice count is the table over this, rounded)")
ap.add_argument("--cycles", default="2800,3100,3400,3700,4000,4300,4600,5000,5400,5900,6400,7000")
ap.add_argument("--reads", type=int, default=500_000)
ap.add_argument("--words", type=int, default=2)
args = ap.parse_args()
dev = torch.device("cuda:0")
CYCLES = [int(x) for x in args.cycles.split(",")]


def k1_ms(eng, seqs, offs, lens, n, L, mc, ref, warm=1):
    for it in range(3 + warm):
        if it == warm:
            eng.kernel_time()
        eng.classify_device(seqs.data_ptr(), offs.data_ptr(), lens.data_ptr(), n, L, d_maxcount=mc.data_ptr())
    torch.cuda.synchronize()
    ms, calls = eng.kernel_time()
    if ref[0] is None:
        ref[0] = mc.clone()
    assert torch.equal(ref[0], mc), "the settings disagree"
    return ms / calls * 1e6 / n


reads_cache = {}
W = args.words
stride = 1 if W == 1 else 2 if W == 2 else 4
for mib in [float(x) for x in args.sizes.split(",")]:
    n_blocks = int(mib * (1 << 20) / (8 * stride)) - 3
    d = capi.DeviceIBF.create(0, 64 * W, 3, 13, W * 64 * n_blocks)
    d.fill_synth(3)
    for L in [int(x) for x in args.lens.split(",")]:
        N = args.reads
        if L not in reads_cache:
            reads_cache[L] = synth.make_reads_device(5, N, L, None, dev)
            torch.cuda.synchronize()
        seqs, offs, lens = reads_cache[L]
        mc = torch.zeros((N, 1), dtype=torch.int16, device=dev)
        torch.cuda.synchronize()  # (torch zeroes on its stream; the engine launches on its own)
        ref = [None]
        counts = [0] + sorted({max(1, int(round(mib / float(t)))) for t in args.targets.split(",")})
        scale = 1.0 if L <= 268 else 1.3
        for n_eq in counts:
            os.environ["RB_PHASE_N_SLICES"] = str(n_eq)
            eng = capi.Engine(0, [d], [])
            eng.set_timing(True)
            plan = eng.plan(0, N, L)
            t_rule = k1_ms(eng, seqs, offs, lens, N, L, mc, ref, warm=3)
            n_sl = plan["phase_slices"] if plan["phased"] else 0
            row = []
            if n_sl:
                for c in CYCLES:
                    ticks = int(min(2000, max(100, c * scale / n_sl)))
                    eng.set_phased(1 << 18, 1 << 32, ticks, 0, 1)
                    p2 = eng.plan(0, N, L)
                    if not p2["phased"] or p2["phase_slices"] != n_sl:
                        continue
                    row.append((ticks, k1_ms(eng, seqs, offs, lens, N, L, mc, ref)))
            eng.destroy()
            best = min(row, key=lambda x: x[1]) if row else (0, 0.0)
            print("%d-word %5.1f MiB %3d bp  n_eq %2d -> %2d slices of %5d KiB  %-28s rule %4d ticks %6.2f | %s | best %6.2f at %d (cycle %d)"
                  % (W, mib, L, n_eq, n_sl, (plan.get("phase_slice_bytes", 0) or 0) >> 10, plan["kernel"][-28:], plan["phase_window_ticks"] if plan["phased"] else 0, t_rule,
                     "  ".join("%d:%.2f" % r for r in row), best[1], best[0], best[0] * n_sl), flush=True)
    d.free()
os.environ.pop("RB_PHASE_N_SLICES", None)

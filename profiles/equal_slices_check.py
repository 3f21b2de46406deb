#!/usr/bin/env python3
"""The equal-length slices of the four-word one-lane builds (rb_phase_plan.h, phase_equal_slices) against the 4 MiB slices they
replace, and the window rule's neighbourhood with them: per (table MiB, read length)

  rule      K1 ms per 1 M reads as the engine plans it now (equal slices, the rule's window),
  sweep     the same slices with the window forced to 0.8 ... 1.25 x the rule's (is the rule next to a cliff?); "best" is the
            smoothed minimum of that sweep,
  pow2      slices of 4 MiB (rb_engine_set_phase_slices(22)) with the window the rule gives for THEIR count.

  python3 profiles/equal_slices_check.py [--points 28:250,32:250,...] [--reads 500000]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from readbouncer_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--points", default="24:250,28:250,32:250,37.73:250,44:250,28:360,32:360,37.73:360,44:360,37.73:200,37.73:430")
ap.add_argument("--reads", type=int, default=500_000)
ap.add_argument("--bins", type=int, default=256, help="bins of the filter (129-192: the three-word builds, 193-256: the four-word builds)")
args = ap.parse_args()
dev = torch.device("cuda:0")
FACTORS = (0.8, 0.9, 1.0, 1.1, 1.25)


def k1_ms(eng, seqs, offs, lens, n, L, mc, ref, warm=2):
    for it in range(3 + warm):
        if it == warm:
            eng.kernel_time()
        eng.classify_device(seqs.data_ptr(), offs.data_ptr(), lens.data_ptr(), n, L, d_maxcount=mc.data_ptr())
    torch.cuda.synchronize()
    ms, calls = eng.kernel_time()
    if ref[0] is None:
        ref[0] = mc.clone()
    assert torch.equal(ref[0], mc), "the forms disagree"
    return ms / calls * 1e6 / n


reads = {}
worst = 0.0
for point in args.points.split(","):
    mib, L = point.split(":")
    mib, L = float(mib), int(L)
    N = args.reads
    if L not in reads:
        reads[L] = synth.make_reads_device(5, N, L, None, dev)
    seqs, offs, lens = reads[L]
    mc = torch.zeros((N, 1), dtype=torch.int16, device=dev)
    n_blocks = int(mib * (1 << 20) / 32) - 3
    d = capi.DeviceIBF.create(0, args.bins, 3, 13, 64 * ((args.bins + 63) // 64) * n_blocks)
    d.fill_synth(3)
    eng = capi.Engine(0, [d], [])
    eng.set_timing(True)
    ref = [None]
    plan = eng.plan(0, N, L)
    t_rule = k1_ms(eng, seqs, offs, lens, N, L, mc, ref, warm=4)
    sweep = {}
    for f in FACTORS:
        ticks = int(plan["phase_window_ticks"] * f)
        eng.set_phased(1 << 18, 1 << 32, ticks, 0, 1)
        assert eng.plan(0, N, L)["phase_slices"] == plan["phase_slices"]
        sweep[ticks] = k1_ms(eng, seqs, offs, lens, N, L, mc, ref)
    eng.set_phased()
    eng.set_phase_slices(22, 32)
    p2 = eng.plan(0, N, L)
    t_pow2 = k1_ms(eng, seqs, offs, lens, N, L, mc, ref)
    eng.set_phase_slices(0, 32)
    t_rule = min(t_rule, k1_ms(eng, seqs, offs, lens, N, L, mc, ref))
    eng.destroy()
    d.free()
    # (the yardstick of profiles/phase_rule_check.py: the sweep smoothed along the window length -- half the point, a quarter of each
    # neighbour -- because a rule must not sit next to a cliff)
    ts = sorted(sweep)
    sm = {t: 0.5 * sweep[t] + 0.25 * sweep[ts[max(i - 1, 0)]] + 0.25 * sweep[ts[min(i + 1, len(ts) - 1)]] for i, t in enumerate(ts)}
    best = min(sm.values())
    worst = max(worst, t_rule / min(best, t_pow2) - 1.0)
    print("%5.2f MiB %3d bp: rule %6.2f (%s, %d slices of %d KiB, %d ticks) | windows %s | 4 MiB slices %6.2f (%d slices, %d ticks) | rule vs best %+5.1f %%"
          % (mib, L, t_rule, plan["phase_shape_name"], plan["phase_slices"], plan["phase_slice_bytes"] >> 10, plan["phase_window_ticks"],
             "  ".join("%d:%.2f" % kv for kv in sorted(sweep.items())), t_pow2, p2["phase_slices"], p2["phase_window_ticks"],
             (t_rule / min(best, t_pow2) - 1.0) * 100), flush=True)
print("worst rule vs best: %+.1f %%" % (worst * 100))
sys.exit(1 if worst > 0.08 else 0)

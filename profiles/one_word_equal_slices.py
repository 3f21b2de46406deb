#!/usr/bin/env python3
"""Equal-length slices for ONE-word tables of 40-127 MiB (DESIGN 8.4 of round 4: a 64 MiB table gained 7-8 % at 12-14 slices with a longer
window; no rule yet).  Per (table MiB, read length): K1 ms per 1 M reads with the planner's rule (slices of 4 MiB, its window), and with
n = ceil(table / target MiB) equal slices for a few targets and window cycles (window = cycle / n ticks).  RB_PHASE_N_SLICES is an
environment switch of measurement processes (RB_TUNING_ENV=1), read when an engine is created.

  python3 profiles/one_word_equal_slices.py [--points 48:250,64:250,...] [--reads 1000000]
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
ap.add_argument("--points", default="40:250,48:250,56:250,64:250,80:250,96:250,127:250,48:360,64:360,96:360,127:360,64:200,64:300")
ap.add_argument("--reads", type=int, default=1_000_000)
ap.add_argument("--targets", default="4.57,5.33,6.4")
ap.add_argument("--cycles", default="7000,8000,9000,10000,11500")
ap.add_argument("--bins", type=int, default=64)
args = ap.parse_args()
dev = torch.device("cuda:0")


def k1_ms(eng, seqs, offs, lens, n, L, mc, ref, warm=2, timed=4):
    for it in range(timed + warm):
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
W = (args.bins + 63) // 64
for point in args.points.split(","):
    mib, L = point.split(":")
    mib, L = float(mib), int(L)
    N = args.reads
    if L not in reads:
        reads[L] = synth.make_reads_device(5, N, L, None, dev)
    seqs, offs, lens = reads[L]
    mc = torch.zeros((N, 1), dtype=torch.int16, device=dev)
    n_blocks = int(mib * (1 << 20) / (8 * W)) - 3
    d = capi.DeviceIBF.create(0, args.bins, 3, 13, 64 * W * n_blocks)
    d.fill_synth(3)
    ref = [None]
    os.environ.pop("RB_PHASE_N_SLICES", None)
    eng = capi.Engine(0, [d], [])
    eng.set_timing(True)
    plan = eng.plan(0, N, L)
    t_rule = k1_ms(eng, seqs, offs, lens, N, L, mc, ref, warm=4)
    eng.destroy()
    if not plan["phased"]:
        print("%6.1f MiB %3d bp: not phased (%s) %.2f" % (mib, L, plan["kernel"], t_rule), flush=True)
        d.free()
        continue
    out = []
    best = (t_rule, "rule")
    for target in [float(x) for x in args.targets.split(",")]:
        n_sl = int(-(-mib // target))
        if n_sl >= plan["phase_slices"] or n_sl < 2:
            continue
        os.environ["RB_PHASE_N_SLICES"] = str(n_sl)
        eng = capi.Engine(0, [d], [])
        eng.set_timing(True)
        row = []
        scale = 1.0 if L <= 260 else 1.24  # (the six-tile build's windows are that much longer: 150 + 6800 / n against 150 + 5500 / n)
        for cyc in [float(x) for x in args.cycles.split(",")]:
            ticks = int(cyc * scale / n_sl)
            eng.set_phased(1 << 18, 1 << 32, ticks, 0, 1)
            p2 = eng.plan(0, N, L)
            if p2["phase_slices"] != n_sl:
                row.append("%d:? (%d slices)" % (ticks, p2["phase_slices"]))
                continue
            t = k1_ms(eng, seqs, offs, lens, N, L, mc, ref)
            row.append("%d:%.2f" % (ticks, t))
            if t < best[0]:
                best = (t, "%d slices of %.2f MiB, %d ticks (cycle %d)" % (n_sl, mib / n_sl, ticks, int(cyc * scale)))
        eng.destroy()
        out.append("%d slices [%s]" % (n_sl, "  ".join(row)))
    os.environ.pop("RB_PHASE_N_SLICES", None)
    d.free()
    print("%6.1f MiB %3d bp: rule %6.2f (%s, %d slices, %d ticks) | %s | best %.2f (%+.1f %%): %s"
          % (mib, L, t_rule, plan["phase_shape_name"], plan["phase_slices"], plan["phase_window_ticks"], " | ".join(out), best[0],
             (best[0] / t_rule - 1) * 100, best[1]), flush=True)

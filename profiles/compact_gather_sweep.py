#!/usr/bin/env python3
"""Round 5 experiment: the compacted window loop (rb_kernels.hip, phased_window_loop_compact, -DRB_COMPACT=1 builds) against the shipped
predicated one, on the narrow shapes of bench.py.  One process per library (RB_AMD_LIBRARY picks it); per workload the K1 time per
1 M reads at the rule's window and at 0.35 ... 1.25 x it, and a SHA-1 of the raw maxima so that the libraries can be compared bit for bit.

  RB_AMD_LIBRARY=readbouncer_amd/exp/libreadbouncer_amd_g6c128.so python3 profiles/compact_gather_sweep.py [--workloads a,b] [--reads N]
"""
import argparse
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from readbouncer_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workloads", default="deplete_target,targets3,readme,readme360,c1,w1_64mib")
ap.add_argument("--reads", type=int, default=1_000_000)
ap.add_argument("--factors", default="0.35,0.45,0.55,0.65,0.75,0.85,1.0,1.15")
args = ap.parse_args()
dev = torch.device("cuda:0")
SHAPES = {  # name -> (deplete keys, target keys, read length)
    "deplete_target": (["mock_t3"], ["mock_t1"], 250),
    "targets3": ([], ["mock_t1", "mock_t2", "mock_t3"], 250),
    "readme": (["mock_deplete"], ["mock_t1", "mock_t2", "mock_t3"], 250),
    "readme360": (["mock_deplete"], ["mock_t1", "mock_t2", "mock_t3"], 360),
    "c1": (["c1"], [], 250),
    "w1_64mib": (["w1_64mib"], [], 250),
}
SEEDS = {"mock_deplete": (11, 110), "mock_t1": (12, 111), "mock_t2": (13, 112), "mock_t3": (14, 113), "c1": (1, 10), "w1_64mib": (15, 114)}
filters = {}


def flt(key):
    if key not in filters:
        filters[key] = synth.build_device_filter(0, synth.WORKLOADS[key], fill_seed=SEEDS[key][0], plant_seed=SEEDS[key][1], n_segments=512)
    return filters[key]


def k1_ms(eng, seqs, offs, lens, n, L, mc, warm=2, timed=4):
    for it in range(timed + warm):
        if it == warm:
            eng.kernel_time()
        eng.classify_device(seqs.data_ptr(), offs.data_ptr(), lens.data_ptr(), n, L, d_maxcount=mc.data_ptr())
    torch.cuda.synchronize()
    ms, calls = eng.kernel_time()
    return ms / timed * 1e6 / n  # per step (a step may launch several count kernels), per 1 M reads


print("library:", os.environ.get("RB_AMD_LIBRARY", "shipped"), flush=True)
import numpy as np  # noqa: E402
for name in args.workloads.split(","):
    dk, tk, L = SHAPES[name]
    dep, tgt = [flt(k)[0] for k in dk], [flt(k)[0] for k in tk]
    ref = np.concatenate([flt(k)[1] for k in dk + tk])
    N = args.reads
    seqs, offs, lens = synth.make_reads_device(77, N, L, ref, dev)
    nf = len(dep) + len(tgt)
    mc = torch.zeros((N, nf), dtype=torch.int16, device=dev)
    eng = capi.Engine(0, dep, tgt)
    eng.set_timing(True)
    plan = eng.plan(0, N, L)
    t_rule = k1_ms(eng, seqs, offs, lens, N, L, mc, warm=4)
    sha = hashlib.sha1(mc.cpu().numpy().tobytes()).hexdigest()[:16]
    sweep = {}
    if plan["phased"]:
        for f in [float(x) for x in args.factors.split(",")]:
            ticks = max(100, int(plan["phase_window_ticks"] * f))
            eng.set_phased(1 << 18, 1 << 32, ticks, 0, 1)
            p2 = eng.plan(0, N, L)
            if p2["phase_slices"] != plan["phase_slices"]:
                continue
            sweep[ticks] = k1_ms(eng, seqs, offs, lens, N, L, mc)
            assert hashlib.sha1(mc.cpu().numpy().tobytes()).hexdigest()[:16] == sha, "results moved with the window"
        eng.set_phased()
    best = min(list(sweep.values()) + [t_rule])
    print("%-15s %3d bp  %s  %d slices of %d KiB  rule %4d ticks: %6.2f ms/M reads (%5.1f M reads/s) | %s | best %6.2f (%5.1f M reads/s)  sha %s"
          % (name, L, plan["kernel"].replace("ibf_count_max_", ""), plan["phase_slices"], plan["phase_slice_bytes"] >> 10, plan["phase_window_ticks"],
             t_rule, 1e3 / t_rule, "  ".join("%d:%.2f" % kv for kv in sorted(sweep.items())), best, 1e3 / best, sha), flush=True)
    eng.destroy()
    del seqs, offs, lens, mc
    torch.cuda.empty_cache()

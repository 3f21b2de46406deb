#!/usr/bin/env python3
"""Round 6: the phased build that carries R reads per wave through a pass of the windows (rb_engine_set_reads_per_wave;
ibf_count_max_phased_multi_kernel, rb_kernels.hip) against the shipped one-read build, on the two-word shapes of bench.py.  Per
workload and R: K1 time per 1 M reads at the rule's window and at a range of factors of it (more reads per wave want longer windows),
and a SHA-1 of the raw maxima so that the builds are compared bit for bit.

  python3 profiles/multi_reads_sweep.py [--workloads deplete_target,targets3] [--reads N] [--rpw 0,1,2,3] [--factors ...] [--slices 0,22,...]
"""
import argparse
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from readbouncer_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workloads", default="deplete_target,targets3")
ap.add_argument("--reads", type=int, default=1_000_000)
ap.add_argument("--rpw", default="0,1,2,3")
ap.add_argument("--factors", default="0.7,0.85,1.0,1.15,1.3,1.5,1.75,2.0,2.4")
ap.add_argument("--slice-log2", default="0", help="comma list; 0 = the planner's slice size")
ap.add_argument("--skew", default="0", help="comma list of rb_engine_set_phase_xcd_skew modes (bit 0: slice skew, bit 1: time skew)")
args = ap.parse_args()
dev = torch.device("cuda:0")
SHAPES = {  # name -> (deplete keys, target keys, read length)
    "deplete_target": (["mock_t3"], ["mock_t1"], 250),
    "targets3": ([], ["mock_t1", "mock_t2", "mock_t3"], 250),
    "deplete_target360": (["mock_t3"], ["mock_t1"], 360),
    "targets3_360": ([], ["mock_t1", "mock_t2", "mock_t3"], 360),
    "readme": (["mock_deplete"], ["mock_t1", "mock_t2", "mock_t3"], 250),
    "readme360": (["mock_deplete"], ["mock_t1", "mock_t2", "mock_t3"], 360),
    "c1": (["c1"], [], 250),
    "c1_360": (["c1"], [], 360),
    "w1_64mib": (["w1_64mib"], [], 250),
    "mock_deplete": (["mock_deplete"], [], 250),
    "mock_deplete360": (["mock_deplete"], [], 360),
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
    return ms / timed * 1e6 / n  # per 1 M reads


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
    sha0 = None
    for sl, skew, rpw in [(a, b, c) for a in [int(x) for x in args.slice_log2.split(",")] for b in [int(x) for x in args.skew.split(",")]
                          for c in [int(x) for x in args.rpw.split(",")]]:
        if True:
            eng.set_reads_per_wave(rpw)
            eng.set_phase_xcd_skew(skew)
            eng.set_phased()
            eng.set_phase_slices(sl, 32)
            plan = eng.plan(0, N, L)
            t_rule = k1_ms(eng, seqs, offs, lens, N, L, mc, warm=4)
            sha = hashlib.sha1(mc.cpu().numpy().tobytes()).hexdigest()[:16]
            if sha0 is None:
                sha0 = sha
            assert sha == sha0, "results differ between builds"
            sweep = {}
            for f in [float(x) for x in args.factors.split(",")]:
                ticks = max(100, int(plan["phase_window_ticks"] * f))
                eng.set_phased(1 << 18, 1 << 32, ticks, 0, 1)
                p2 = eng.plan(0, N, L)
                if p2["phase_slices"] != plan["phase_slices"]:
                    continue
                sweep[ticks] = k1_ms(eng, seqs, offs, lens, N, L, mc)
                assert hashlib.sha1(mc.cpu().numpy().tobytes()).hexdigest()[:16] == sha0, "results moved with the window"
            best_t = min(sweep, key=sweep.get) if sweep else plan["phase_window_ticks"]
            best = min(list(sweep.values()) + [t_rule])
            print("%-15s %3d bp  R=%d skew=%d  %s  %2d slices of %5d KiB  rule %4d ticks: %6.2f ms/M | %s | best %6.2f at %d (%5.1f M reads/s)  sha %s"
                  % (name, L, rpw, skew, plan["kernel"].replace("ibf_count_max_", ""), plan["phase_slices"], plan["phase_slice_bytes"] >> 10,
                     plan["phase_window_ticks"], t_rule, "  ".join("%d:%.2f" % kv for kv in sorted(sweep.items())), best, best_t, 1e3 / best, sha), flush=True)
    eng.destroy()
    del seqs, offs, lens, mc
    torch.cuda.empty_cache()

#!/usr/bin/env python3
"""Results never depend on timing: the same batch launched again and again must give the same raw maxima and decisions, bit for bit --
the plain kernel on the 8 GiB filter (40 launches of 10 M reads) and the clock-phased kernels, whose waves pick their slices by the wall
clock (README shape at 250 and 360 bp, two-word and one-word tables: 1 500 launches of 1 M reads each, window lengths varied on the way)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from readbouncer_amd import capi, synth
dev = torch.device("cuda:0")


def digest(*ts):
    h = 0
    for t in ts:
        v = t.reshape(-1).view(torch.uint8).to(torch.int64)
        w = torch.arange(1, v.numel() + 1, device=v.device, dtype=torch.int64) % 1000003
        h = (h * 1000003 + int((v * w).sum())) & ((1 << 62) - 1)
    return h


def soak(name, dep_keys, tgt_keys, n, L, launches, seeds, windows=(None,)):
    filters = {k: synth.build_device_filter(0, synth.WORKLOADS[k], fill_seed=seeds[k][0], plant_seed=seeds[k][1], n_segments=512 if k.startswith(("mock", "w1")) else 2048) for k in dep_keys + tgt_keys}
    ref = np.concatenate([filters[k][1] for k in dep_keys + tgt_keys])
    seqs, offs, lens = synth.make_reads_device(99, n, L, ref, dev)
    nf = len(dep_keys) + len(tgt_keys)
    mc = torch.zeros((n, nf), dtype=torch.int16, device=dev)
    dec = torch.zeros(n, dtype=torch.uint8, device=dev)
    eng = capi.Engine(0, [filters[k][0] for k in dep_keys], [filters[k][0] for k in tgt_keys])
    ref_d, t0, bad = None, time.time(), 0
    for i in range(launches):
        w = windows[i % len(windows)]
        if w is not None:
            eng.set_phased(1 << 18, 1 << 32, w, 0, 1)
        mc.zero_(); dec.zero_()
        if os.environ.get("RB_SOAK_RACY") != "1":  # (RB_SOAK_RACY=1: the harness as it was up to session 33, to see WHAT a differing launch holds)
            torch.cuda.synchronize()  # torch zeroes on ITS stream, the engine launches on its own (non-blocking) one: without this the zeroing races the kernels
        eng.classify_device(seqs.data_ptr(), offs.data_ptr(), lens.data_ptr(), n, L, d_maxcount=mc.data_ptr(), d_decision=dec.data_ptr())
        torch.cuda.synchronize()
        d = digest(mc, dec)
        if ref_d is None:
            ref_d, ref_mc, ref_dec = d, mc.clone(), dec.clone()
        if d != ref_d:  # say what differs: how many entries, where, and what they hold (zeros would point at the harness, not at a kernel)
            bad += 1
            diff = (mc != ref_mc).nonzero()
            print("   launch %d (window %s): %d raw maxima and %d decisions differ; first: %s got %s expected %s"
                  % (i, w, diff.shape[0], int((dec != ref_dec).sum()), diff[:4].tolist(),
                     [int(mc[a, b]) for a, b in diff[:4].tolist()], [int(ref_mc[a, b]) for a, b in diff[:4].tolist()]), flush=True)
    print("%-28s %5d launches of %8d reads (%d bp), %5.1f s: %d launches differ from the first; digest %x; decisions %s"
          % (name, launches, n, L, time.time() - t0, bad, ref_d, torch.bincount(dec.to(torch.int64), minlength=3).tolist()), flush=True)
    eng.destroy()
    for f, _ in filters.values():
        f.free()
    return bad


ONLY = os.environ.get("RB_SOAK_ONLY", "")
S = {"c3": (4, 40), "zymo": (6, 60), "mock_deplete": (11, 110), "mock_t1": (12, 111), "mock_t2": (13, 112), "mock_t3": (14, 113), "w1_64mib": (15, 114), "c1": (1, 10)}
bad = soak("c4 (8 GiB + 600 bins)", ["c3"], ["zymo"], 10_000_000, 360, 40, S) if (not ONLY or ONLY in "c4 (8 GiB + 600 bins)") else 0
bad += soak("README shape 250 bp", ["mock_deplete"], ["mock_t1", "mock_t2", "mock_t3"], 1_000_000, 250, 1500, S, windows=(None, None, 300, 450, 700, 150)) if (not ONLY or ONLY in "README shape 250 bp") else 0
bad += soak("README shape 360 bp", ["mock_deplete"], ["mock_t1", "mock_t2", "mock_t3"], 1_000_000, 360, int(os.environ.get("RB_SOAK_README360", "800")), S, windows=(None, 250, 420, 600)) if (not ONLY or ONLY in "README shape 360 bp") else 0
bad += soak("deplete + target (two-word)", ["mock_t3"], ["mock_t1"], 1_000_000, 250, 1500, S, windows=(None, None, 500, 900, 1300, 200)) if (not ONLY or ONLY in "deplete + target (two-word)") else 0
bad += soak("one-word 64 MiB, equal slices", ["w1_64mib"], [], 1_000_000, 250, 1000, S, windows=(None, 400, 750, 1100)) if (not ONLY or ONLY in "one-word 64 MiB, equal slices") else 0
bad += soak("config-1 geometry 360 bp", ["c1"], [], 1_000_000, 360, 1000, S, windows=(None, 600, 1250, 1800)) if (not ONLY or ONLY in "config-1 geometry 360 bp") else 0
print("TOTAL differing launches:", bad)
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""The determinism soak with RANDOM window lengths (50 ... 2 000 ticks, a new one every launch), random XCD skew modes and random slice cuts
(the rule's, 2^k-byte slices, 2 ... 12 equal slices) on the LDS-offset builds: README shape (four-word blocks) and the two-word merged table
at 250 and 360 bp, the one-word 19.8 MB table at 360 bp.  Every launch must equal the first bit for bit; a launch that does not is described
(how many raw maxima and decisions differ, where, what they hold).  Harness synchronised (profiles/r06/README.md, soak rows).

  python3 profiles/soak_random_windows.py [--launches 6000]
"""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from readbouncer_amd import capi, synth
ap = argparse.ArgumentParser()
ap.add_argument("--launches", type=int, default=6000)
args = ap.parse_args()
dev = torch.device("cuda:0")
S = {"mock_deplete": (11, 110), "mock_t1": (12, 111), "mock_t2": (13, 112), "mock_t3": (14, 113), "c1": (1, 10)}


def soak(name, dep_keys, tgt_keys, n, L, launches, seed):
    rng = np.random.default_rng(seed)
    filters = {k: synth.build_device_filter(0, synth.WORKLOADS[k], fill_seed=S[k][0], plant_seed=S[k][1], n_segments=512) for k in dep_keys + tgt_keys}
    ref = np.concatenate([filters[k][1] for k in dep_keys + tgt_keys])
    seqs, offs, lens = synth.make_reads_device(99, n, L, ref, dev)
    nf = len(dep_keys) + len(tgt_keys)
    mc = torch.zeros((n, nf), dtype=torch.int16, device=dev)
    dec = torch.zeros(n, dtype=torch.uint8, device=dev)
    eng = capi.Engine(0, [filters[k][0] for k in dep_keys], [filters[k][0] for k in tgt_keys])
    torch.cuda.synchronize()
    ref_mc = ref_dec = None
    bad, t0 = 0, time.time()
    for i in range(launches):
        how = "rule"
        if i:
            ticks = int(rng.integers(50, 2001))
            eng.set_phased(1 << 18, 1 << 32, ticks, 0, 1)
            eng.set_phase_xcd_skew(int(rng.integers(0, 4)))
            cut = int(rng.integers(0, 3))
            if cut == 0:
                eng.set_phase_slices(0, 32); eng.set_phase_equal_slices(0)
            elif cut == 1:
                eng.set_phase_equal_slices(0); eng.set_phase_slices(int(rng.integers(19, 23)), 32)
            else:
                eng.set_phase_slices(0, 32); eng.set_phase_equal_slices(int(rng.integers(2, 13)))
            how = "%d ticks, cut %d" % (ticks, cut)
        mc.zero_(); dec.zero_()
        torch.cuda.synchronize()
        eng.classify_device(seqs.data_ptr(), offs.data_ptr(), lens.data_ptr(), n, L, d_maxcount=mc.data_ptr(), d_decision=dec.data_ptr())
        torch.cuda.synchronize()
        if ref_mc is None:
            ref_mc, ref_dec = mc.clone(), dec.clone()
            kern = eng.plan(0, n, L)["kernel"]
        elif not (torch.equal(mc, ref_mc) and torch.equal(dec, ref_dec)):
            bad += 1
            diff = (mc != ref_mc).nonzero()
            print("   launch %d (%s): %d raw maxima and %d decisions differ; first: %s got %s expected %s"
                  % (i, how, diff.shape[0], int((dec != ref_dec).sum()), diff[:4].tolist(),
                     [int(mc[a, b]) for a, b in diff[:4].tolist()], [int(ref_mc[a, b]) for a, b in diff[:4].tolist()]), flush=True)
    print("%-34s %-36s %6d launches of %8d reads (%d bp), %6.1f s: %d launches differ from the first; decisions %s"
          % (name, kern, launches, n, L, time.time() - t0, bad, torch.bincount(ref_dec.to(torch.int64), minlength=3).tolist()), flush=True)
    eng.destroy()
    for f, _ in filters.values():
        f.free()
    return bad


bad = 0
bad += soak("README shape 360 bp", ["mock_deplete"], ["mock_t1", "mock_t2", "mock_t3"], 500_000, 360, args.launches, 1)
bad += soak("README shape 250 bp", ["mock_deplete"], ["mock_t1", "mock_t2", "mock_t3"], 500_000, 250, args.launches, 2)
bad += soak("deplete + target 360 bp (two-word)", ["mock_t3"], ["mock_t1"], 500_000, 360, args.launches, 3)
bad += soak("deplete + target 250 bp (two-word)", ["mock_t3"], ["mock_t1"], 500_000, 250, args.launches, 4)
bad += soak("config-1 geometry 360 bp (22-bit)", ["c1"], [], 500_000, 360, args.launches, 5)
print("TOTAL differing launches:", bad)
sys.exit(1 if bad else 0)

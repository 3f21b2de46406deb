#!/usr/bin/env python3
"""What rb_engine_calibrate finds on this box: K1 ms per 1 M reads with the planner's windows and after calibration, for the README
shape (packed merged table), three targets alone, and single narrow filters at lengths between the fitted ones."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from readbouncer_amd import capi, synth  # noqa: E402

dev = torch.device("cuda:0")
N = 1_000_000


SPREAD = {}


def k1(eng, seqs, offs, lens, L, mc):
    """median K1 ms of seven launches after one untimed (the spread is kept for the report)"""
    ts = []
    for it in range(8):
        eng.kernel_time()
        eng.classify_device(seqs.data_ptr(), offs.data_ptr(), lens.data_ptr(), N, L, d_maxcount=mc.data_ptr())
        torch.cuda.synchronize()
        ms, calls = eng.kernel_time()
        if it:
            ts.append(ms / calls)
    ts.sort()
    SPREAD["last"] = (ts[0], ts[-1])
    return ts[len(ts) // 2]


def case(name, deplete, target, L):
    seqs, offs, lens = synth.make_reads_device(5, N, L, None, dev)
    mc = torch.zeros((N, len(deplete) + len(target)), dtype=torch.int16, device=dev)
    torch.cuda.synchronize()  # torch filled these on ITS stream; the engine launches on its own (non-blocking) stream
    eng = capi.Engine(0, deplete, target)
    eng.set_timing(True)
    before = k1(eng, seqs, offs, lens, L, mc)
    sp0 = SPREAD["last"]
    ref = mc.clone()
    plan0 = eng.plan(0, N, L)
    nt, nc = (0, 0) if os.environ.get("RB_NO_CAL") == "1" else eng.calibrate(int(os.environ.get("RB_CAL_N", str(N))), L, 0.0)  # at the batch size it is then run with
    torch.cuda.synchronize()
    plan1 = eng.plan(0, N, L)
    after = k1(eng, seqs, offs, lens, L, mc)
    assert torch.equal(ref, mc)
    sp1 = SPREAD["last"]
    print("%-44s %3d bp: %6.2f ms [%.2f-%.2f] -> %6.2f ms [%.2f-%.2f] (%+5.1f %%)  tables %d changed %d  window %d -> %d ticks (%s)"
          % (name, L, before, sp0[0], sp0[1], after, sp1[0], sp1[1], (after / before - 1) * 100, nt, nc, plan0["phase_window_ticks"],
             plan1["phase_window_ticks"], plan0["phase_shape_name"]), flush=True)
    eng.destroy()


mock = {}
for i, key in enumerate(("mock_deplete", "mock_t1", "mock_t2", "mock_t3")):
    mock[key] = synth.build_device_filter(0, synth.WORKLOADS[key], fill_seed=11 + i, plant_seed=110 + i, n_segments=512)[0]
for L in (250, 360, 200, 300):
    case("README shape (packed merged table, 4 words)", [mock["mock_deplete"]], [mock["mock_t1"], mock["mock_t2"], mock["mock_t3"]], L)
    case("three targets (packed merged table, 2 words)", [], [mock["mock_t1"], mock["mock_t2"], mock["mock_t3"]], L)
for W, mib in ((1, 13), (2, 13), (2, 45), (1, 45)):
    stride = W
    n_blocks = int(mib * (1 << 20) / (8 * stride)) - 3
    d = capi.DeviceIBF.create(0, 64 * W, 3, 13, W * 64 * n_blocks)
    d.fill_synth(3)
    for L in (200, 250, 300):
        case("%d-word filter of %d MiB" % (W, mib), [d], [], L)
    d.free()

#!/usr/bin/env python3
"""Slice size of the clock-phased gathers.  Until session 25 the engine cut a table into at most 8 slices by a SHIFT of the
block number, so a slice was between 1/8 and 1/4 of the table: 1 MiB for an 8 MiB table, 2 MiB for 8-16 MiB, 4 MiB for 16-32 MiB.
Fewer, larger slices mean fewer passes over a read's lookups.  This sweeps slice size (RB_PHASE_SLICE_LOG2) x window length
for single filters; K1 ms per 1 M reads (hipEvents, rb_engine_kernel_time).  "rule" = what the engine does on its own.
Usage (GPU box): python profiles/r03/slice_size_sweep.py WORDS READ_LENS SIZES_MIB SLICE_LOG2S TICKS   (comma lists)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from readbouncer_amd import capi, synth  # noqa: E402

dev = torch.device("cuda:0")
W = int(sys.argv[1])
LENS = [int(x) for x in sys.argv[2].split(",")]
SIZES = [float(x) for x in sys.argv[3].split(",")]
SLICES = [int(x) for x in sys.argv[4].split(",")]
TICKS = [int(x) for x in sys.argv[5].split(",")]
print("ticks: rule | " + " ".join("%6d" % t for t in TICKS))
for L in LENS:
    N = 1_000_000 if L <= 400 else 400_000
    seqs, offs, lens = synth.make_reads_device(5, N, L, None, dev)
    mc = torch.zeros((N, 1), dtype=torch.int16, device=dev)
    for mb in SIZES:
        n_blocks = int(mb * (1 << 20) / (8 * W)) - 3
        d = capi.DeviceIBF.create(0, 64 * W, 3, 13, W * 64 * n_blocks)
        d.fill_synth(3)
        ref = None
        for lg2 in [0, -1] + SLICES:
            if lg2 > 0:
                os.environ["RB_PHASE_SLICE_LOG2"] = str(lg2)
            else:
                os.environ.pop("RB_PHASE_SLICE_LOG2", None)
            eng = capi.Engine(0, [d], [])
            eng.set_timing(True)
            if lg2 == -1:
                eng.set_phased(0, 0, 0, 0)  # the plain kernel
            row = []
            for ticks in ((0,) if lg2 <= 0 else TICKS):
                if ticks:
                    eng.set_phased(1 << 20, 1 << 30, ticks, 0)
                for it in range(4):
                    if it == 1:
                        eng.kernel_time()
                    eng.classify_device(seqs.data_ptr(), offs.data_ptr(), lens.data_ptr(), N, L, d_maxcount=mc.data_ptr())
                torch.cuda.synchronize()
                ms, calls = eng.kernel_time()
                row.append(ms / calls * 1e6 / N)
                if ref is None:
                    ref = mc.clone()
                assert torch.equal(ref, mc)
            eng.destroy()
            label = "rule" if lg2 == 0 else "plain kernel" if lg2 == -1 else ("slices of %d MiB" % (1 << (lg2 - 20)) if lg2 >= 20 else "slices of %d KiB" % (1 << (lg2 - 10)))
            print("%4d bp %d-word %6.1f MiB %-15s: %s" % (L, W, mb, label, " ".join("%6.2f" % x for x in row)), flush=True)
        d.free()

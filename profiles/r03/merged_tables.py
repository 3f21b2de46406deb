#!/usr/bin/env python3
"""A/B of the merged table on pairs of LARGE filters (the fuse rule of plan_merged was measured on the README shape's narrow
tables): two filters of one hash geometry, far beyond the L2s and the Infinity Cache; 1 M reads of 360 bp per launch, hipEvent
time of the count kernels (rb_engine_kernel_time), merge off against merge on; maxima compared.
Usage (GPU box): python profiles/r03/merged_tables.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from readbouncer_amd import capi, synth  # noqa: E402

dev = torch.device("cuda:0")
N, L = 1_000_000, 360
seqs, offs, lens = synth.make_reads_device(5, N, L, None, dev)
CASES = [(64, 2, 1 << 25), (128, 2, 1 << 24), (256, 2, 1 << 24), (512, 2, 1 << 23), (64, 2, 1 << 21)]
if len(sys.argv) > 1 and sys.argv[1] == "sweep":  # where does a pair (a triple) start to pay?  table sizes across the phased range and beyond
    CASES = [(64, m, int(mb * 1e6 / 8)) for m in (2, 3) for mb in (1, 2, 4, 8, 12, 24, 33, 48, 64, 128)]
    CASES += [(128, 2, int(mb * 1e6 / 16)) for mb in (4, 8, 16, 24, 33, 48, 64)]
for bins, n_members, blocks in CASES:
    W = (bins + 63) // 64
    n_blocks = blocks - 3
    fs = []
    for i in range(n_members):
        d = capi.DeviceIBF.create(0, bins, 3, 13, W * 64 * n_blocks)
        d.fill_synth(11 + i)
        fs.append(d)
    eng = capi.Engine(0, fs[:1], fs[1:])
    eng.set_timing(True)
    out = {}
    for mode in (0, 2):
        eng.set_merge(mode)
        mc = torch.zeros((N, n_members), dtype=torch.int16, device=dev)
        for it in range(5):
            if it == 2:
                eng.kernel_time()
            eng.classify_device(seqs.data_ptr(), offs.data_ptr(), lens.data_ptr(), N, L, d_maxcount=mc.data_ptr())
        torch.cuda.synchronize()
        ms, calls = eng.kernel_time()
        out[mode] = (ms / calls, mc.clone())
    same = bool(torch.equal(out[0][1], out[2][1]))
    print("%d x %4d bins (%d words each), %6.1f MB per table: apart %7.2f ms, merged %7.2f ms (%.2fx), info %s, equal %s"
          % (n_members, bins, W, W * 8 * n_blocks / 1e6, out[0][0], out[2][0], out[0][0] / out[2][0], eng.merge_info(), same), flush=True)
    eng.destroy()
    for d in fs:
        d.free()

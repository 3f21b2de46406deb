#!/usr/bin/env python3
"""Reads per call from which the phased form pays on the LARGER tables it serves since session 28 (a read needs at least one
whole cycle over the table's slices, 33-110 us): one-word filters of 10.5 / 32 / 64 / 120 MiB, 250 bp, batches of 2 049 ... 262 144
reads; K1 us per call (hipEvents, rb_engine_kernel_time), built-in rule against the same kernel without phases.
Usage (GPU box): python profiles/r03/phased_batch_size_large_tables.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from readbouncer_amd import capi, synth  # noqa: E402

dev = torch.device("cuda:0")
L = int(sys.argv[1]) if len(sys.argv) > 1 else 250
NMAX = 262144
seqs, offs, lens = synth.make_reads_device(5, NMAX, L, None, dev)
mc = torch.zeros((NMAX, 1), dtype=torch.int16, device=dev)
for W, mb in ((1, 10.5), (1, 32.0), (1, 64.0), (1, 120.0), (2, 20.0), (2, 64.0)):
    n_blocks = int(mb * (1 << 20) / (8 * W)) - 3
    d = capi.DeviceIBF.create(0, 64 * W, 3, 13, W * 64 * n_blocks)
    d.fill_synth(3)
    eng = capi.Engine(0, [d], [])
    eng.set_timing(True)
    for n in (2049, 4096, 8192, 16384, 65536, 262144):
        row = []
        for phased in (True, False):
            if phased:
                eng.set_phased()
            else:
                eng.set_phased(0, 0, 300, 3)  # no phases, same kernel
            for it in range(13):
                if it == 3:
                    eng.kernel_time()
                eng.classify_device(seqs.data_ptr(), offs.data_ptr(), lens.data_ptr(), n, L, d_maxcount=mc.data_ptr())
            torch.cuda.synchronize()
            ms, calls = eng.kernel_time()
            row.append(ms / calls * 1e3)
        print("%d bp %d-word %6.1f MiB  %7d reads per call: phased %8.1f us, plain %8.1f us  (%.2fx)" % (L, W, mb, n, row[0], row[1], row[1] / row[0]), flush=True)
    eng.destroy()
    d.free()

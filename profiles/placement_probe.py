#!/usr/bin/env python3
"""Config 3 at the reference's own sizing (4.72 GB) runs in one of two modes from process to process (3.25 or 3.17 M reads/s; its
no-compute probe 6.93 or 6.77 TB/s: profiles/r04/c3np2_bimodal.txt) -- "where the allocation lands".  Does the mode vary between
allocations of ONE process?  Allocates the table several times (earlier ones kept, so that every allocation gets other pages), probes each
with rb_dibf_probe_read_peak, then frees all and allocates once more.  If it varies within a process, a filter could be placed by trial."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from readbouncer_amd import capi, synth  # noqa: E402

w = synth.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c3np2"]
bits = synth.filter_bits(w)
keep = []
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 8):
    d = capi.DeviceIBF.create(0, w["n_bins"], w["h"], w["k"], bits)
    d.fill_synth(4)
    torch.cuda.synchronize()
    runs = [d.probe_read_peak(1024, True, 24, target_ms=120.0)[0] for _ in range(3)]
    print("allocation %d at 0x%x: probe %s GB/s" % (i, d.device_words(), " ".join("%.0f" % x for x in runs)), flush=True)
    keep.append(d)
for d in keep:
    d.free()
torch.cuda.synchronize()
for i in range(3):
    d = capi.DeviceIBF.create(0, w["n_bins"], w["h"], w["k"], bits)
    d.fill_synth(4)
    torch.cuda.synchronize()
    print("after freeing all, allocation %d at 0x%x: probe %.0f GB/s" % (i, d.device_words(), d.probe_read_peak(1024, True, 24, target_ms=120.0)[0]), flush=True)
    d.free()

#!/usr/bin/env python3
"""IBF::load_filter at scale (src/IBF/IBFBuild.cpp:329-396): how long does a .ibf take from a file into HBM, ready to classify?
Makes a filter of the given workload on the device (design-load synthetic bits), writes it as a reference-format .ibf into DIR (default /dev/shm:
the file is then in memory, which is what a second start of the program sees of a file on disk), and times
  rb_dibf_open            file -> HBM, streamed through pinned staging (no host image)
  rb_ibf_open + upload    file -> host image -> HBM (what the mirror's load_filter + an upload would do)
  rb_ibf_store            host image -> file
three times each, with the probe of the opened table and a bit-for-bit comparison against the filter the file was written from.
Usage: python3 profiles/load_throughput.py [workload=c3] [dir=/dev/shm]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from readbouncer_amd import capi, synth  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
d = sys.argv[2] if len(sys.argv) > 2 else "/dev/shm"
path = os.path.join(d, "rb_load_throughput_%d.ibf" % os.getpid())
t = time.perf_counter()
dev, _ = synth.build_device_filter(0, synth.WORKLOADS[wl], fill_seed=4, plant_seed=40)
info = dev.info
print("%s: %d bins, %d blocks, %.2f GiB of payload; made on the device in %.1f s" % (wl, info["n_bins"], info["n_blocks"], info["n_words"] * 8 / 2**30, time.perf_counter() - t), flush=True)
try:
    t = time.perf_counter(); host = dev.download(); t_down = time.perf_counter() - t
    size = info["n_words"] * 8
    for rep in range(2):
        t = time.perf_counter(); host.store(path); t_store = time.perf_counter() - t
        print("rb_dibf_download %.2f s (%.1f GB/s)   rb_ibf_store %.2f s (%.1f GB/s)" % (t_down, size / t_down / 1e9, t_store, size / t_store / 1e9), flush=True)
    want = host.words().copy() if size <= (2 << 30) else None
    ref_sum = int(np.bitwise_xor.reduce(host.words()))
    host.close(); del host
    capi.set_placement_tries(1)  # (timed apart below: the placement trial is a cost of its own)
    for rep in range(3):
        t = time.perf_counter(); f = capi.DeviceIBF.open(0, path); dt = time.perf_counter() - t
        print("rb_dibf_open (one allocation)      %.2f s  %.2f GB/s" % (dt, size / dt / 1e9), flush=True)
        if rep == 2:
            h2 = f.download(); ok = int(np.bitwise_xor.reduce(h2.words())) == ref_sum and (want is None or np.array_equal(want, h2.words()))
            print("   round trip file -> HBM -> host: %s" % ("identical" if ok else "DIFFERENT"), flush=True)
            h2.close(); del h2
        del f
    capi.set_placement_tries(5)
    for rep in range(2):
        t = time.perf_counter(); f = capi.DeviceIBF.open(0, path); dt = time.perf_counter() - t
        print("rb_dibf_open (placement by trial)  %.2f s  %.2f GB/s   placement %s" % (dt, size / dt / 1e9, f.placement()), flush=True)
        del f
    for rep in range(2):
        t = time.perf_counter(); h = capi.HostIBF.open(path); t1 = time.perf_counter() - t
        t = time.perf_counter(); f = capi.DeviceIBF.upload(0, h); t2 = time.perf_counter() - t
        print("rb_ibf_open %.2f s (%.2f GB/s) + rb_dibf_upload %.2f s (%.2f GB/s)" % (t1, size / t1 / 1e9, t2, size / t2 / 1e9), flush=True)
        del f; h.close(); del h
finally:
    if os.path.exists(path):
        os.remove(path)

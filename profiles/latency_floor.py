#!/usr/bin/env python3
"""Fixed cost of one rb_classify_batch call (host buffers in, decisions out) by batch size and filter count."""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from readbouncer_amd import capi, synth

w = synth.WORKLOADS["c2"]
d, ref = synth.build_device_filter(0, w, 2, 20)
small = [capi.DeviceIBF.create(0, 60, 3, 13, 64 * 200003) for _ in range(3)]
buf, offs, lens = synth.make_reads(3, 4096, 360, ref)
for name, dep, tgt in (("1 filter", [d], []), ("1+1 filters", [d], small[:1]), ("1+3 filters", [d], small)):
    eng = capi.Engine(0, dep, tgt)
    for n in (1, 8, 64, 256, 1024):
        sub = np.ascontiguousarray(buf[: n * 360]); so, sl = offs[:n].copy(), lens[:n].copy()
        for _ in range(20):
            eng.classify(sub, so, sl)
        ts = []
        for _ in range(300):
            a = time.perf_counter(); eng.classify(sub, so, sl); ts.append((time.perf_counter() - a) * 1e6)
        ts = np.sort(ts)
        print("%-12s n=%4d  p50 %.1f us  p99 %.1f us" % (name, n, ts[150], ts[296]))

#!/usr/bin/env python3
"""200 single-read rb_classify_batch calls against the c2 filter; run under `rocprofv3 --kernel-trace` to see where the
fixed cost of a call goes (profiles/latency_trace_report.py turns the trace into per-call gaps)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from readbouncer_amd import capi, synth
d, ref = synth.build_device_filter(0, synth.WORKLOADS["c2"], 2, 20)
buf, offs, lens = synth.make_reads(3, 64, 360, ref)
eng = capi.Engine(0, [d], [])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sub = np.ascontiguousarray(buf[: n * 360]); so, sl = offs[:n].copy(), lens[:n].copy()
for _ in range(50):
    eng.classify(sub, so, sl)
ts = []
for _ in range(200):
    a = time.perf_counter(); eng.classify(sub, so, sl); ts.append((time.perf_counter() - a) * 1e6)
    time.sleep(0.0002)
print("host p50 %.1f us" % np.median(ts))

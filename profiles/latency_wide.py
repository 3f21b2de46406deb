#!/usr/bin/env python3
"""Latency of one rb_classify_batch call against WIDE filters (config 3/4: 8192 bins, 1 KiB blocks, 8 GiB) by batch
size and by the multi-workgroup setting of the latency kernel (workgroups per read x shares per 64-k-mer tile)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from readbouncer_amd import capi, synth  # noqa: E402

dep, ref = synth.build_device_filter(0, synth.WORKLOADS["c3"], 4, 40)
tgt, _ = synth.build_device_filter(0, synth.WORKLOADS["zymo"], 6, 60)
buf, offs, lens = synth.make_reads(5, 4096, 360, ref)
settings = [(1, 1), (4, 2), (6, 4), (8, 4), (12, 8), (16, 8)]
for name, d, t in (("c3 (deplete only)", [dep], []), ("c4 (deplete + target)", [dep], [tgt])):
    eng = capi.Engine(0, d, t)
    for n in (1, 4, 14, 64, 256):
        sub = np.ascontiguousarray(buf[: n * 360]); so, sl = offs[:n].copy(), lens[:n].copy()
        row = []
        for parts, shares in settings:
            eng.set_split_parts(parts, shares)
            for _ in range(30):
                eng.classify(sub, so, sl)
            ts = []
            for _ in range(300):
                a = time.perf_counter(); eng.classify(sub, so, sl); ts.append((time.perf_counter() - a) * 1e6)
            ts = np.sort(ts)
            row.append("%dx%d %.0f/%.0f" % (parts, shares, ts[150], ts[296]))
        print("%-22s n=%4d  p50/p99 us: %s" % (name, n, "  ".join(row)), flush=True)

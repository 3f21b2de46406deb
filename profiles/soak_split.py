#!/usr/bin/env python3
"""Soak of the multi-workgroup latency kernel on the 8 GiB filter (real HBM latencies, all XCDs): thousands of
micro-batches of random size and random (parts, shares) settings; every maxcount is compared with the throughput
form's result for the same read (which test_gpu_parity pins to the oracle).  A lost update or a stale partial
counter plane would show up as a mismatch.  Usage: soak_split.py [calls]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from readbouncer_amd import capi, synth  # noqa: E402

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dep, ref = synth.build_device_filter(0, synth.WORKLOADS["c3"], 4, 40)
tgt, _ = synth.build_device_filter(0, synth.WORKLOADS["zymo"], 6, 60)
wide2 = capi.DeviceIBF.create(0, 2500, 3, 13, 40 * 64 * 3_000_017)  # 40 word columns, 0.96 GB, generic modulus
wide2.fill_synth(9)
rng = np.random.default_rng(123)
n_pool = 4096
lens_pool = rng.integers(0, 700, size=n_pool).astype(np.uint32)
lens_pool[:64] = 360
buf, offs, _ = synth.make_reads(5, n_pool, 700, ref)
eng = capi.Engine(0, [dep, wide2], [tgt])
eng.set_split_threshold(0)
expect = eng.classify(buf, offs, lens_pool)[0]  # throughput form
eng.set_split_threshold(2048)
bad = 0
checked = 0
t0 = time.time()
for c in range(calls):
    if c % 50 == 0:
        eng.set_split_parts(int(rng.choice([1, 2, 3, 4, 6, 8, 16])), int(rng.choice([1, 2, 4, 8])))
    n = int(rng.choice([1, 2, 3, 5, 9, 14, 20, 33, 64, 100, 300]))
    s = int(rng.integers(0, n_pool - n))
    mc = eng.classify(buf, offs[s:s + n], lens_pool[s:s + n])[0]
    checked += n
    if not np.array_equal(mc, expect[s:s + n]):
        bad += int((mc != expect[s:s + n]).any(axis=1).sum())
print("soak_split: %d calls, %d reads x 3 filters checked, %d mismatching reads, %.1f s" % (calls, checked, bad, time.time() - t0))
sys.exit(1 if bad else 0)

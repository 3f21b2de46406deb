#!/usr/bin/env python3
"""The reference's threading model on the GPU engine: T host threads classify ONE read per call (check_unblock per read,
src/main/adaptive_sampling.hpp:745-751).  One engine per thread (what the C++ mirror does: engines borrow the filters
and own their streams/workspaces) against one engine shared by all threads (calls serialise on its staging buffers)."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from readbouncer_amd import capi, synth  # noqa: E402

dep, ref = synth.build_device_filter(0, synth.WORKLOADS["c3"], 4, 40)
tgt, _ = synth.build_device_filter(0, synth.WORKLOADS["zymo"], 6, 60)
buf, offs, lens = synth.make_reads(5, 4096, 360, ref)
calls = 1500


def run(n_threads, shared):
    engines = [capi.Engine(0, [dep], [tgt])] * n_threads if shared else [capi.Engine(0, [dep], [tgt]) for _ in range(n_threads)]
    lat = [None] * n_threads

    def work(t):
        eng = engines[t]
        ts = np.zeros(calls)
        for i in range(calls):
            j = (t * calls + i) % 4096
            a = time.perf_counter()
            eng.classify(buf[j * 360:(j + 1) * 360], offs[:1], lens[:1])
            ts[i] = time.perf_counter() - a
        lat[t] = ts
    for e in set(engines):
        e.classify(buf[:360], offs[:1], lens[:1])
    th = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    wall = time.perf_counter() - t0
    all_lat = np.sort(np.concatenate(lat)) * 1e6
    return n_threads * calls / wall, all_lat[len(all_lat) // 2], all_lat[int(len(all_lat) * 0.99)]


for shared in (False, True):
    for T in (1, 2, 4, 8, 16):
        rate, p50, p99 = run(T, shared)
        print("%-22s threads %2d: %8.0f reads/s   per-call p50 %.0f us  p99 %.0f us" % (
            "one engine, shared" if shared else "one engine per thread", T, rate, p50, p99), flush=True)

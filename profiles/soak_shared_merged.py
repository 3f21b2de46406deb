#!/usr/bin/env python3
"""Soak of the shared merged copy (rb_engine.hip, MergedTable): engines are created, used and destroyed on several host threads
while another thread keeps inserting into a member filter.  Every result a thread sees must be one the filters could have had at
some moment (the maxima of a read never go DOWN as sequences are added), and once the inserts have stopped every engine must
agree with the oracle on the final bits.

  python3 profiles/soak_shared_merged.py [seconds=10] [threads=6]
"""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from readbouncer_amd import capi  # noqa: E402
from oracle import pyoracle as po  # noqa: E402  (the checker)
import helpers as H  # noqa: E402

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rng = np.random.default_rng(77)
ref = H.random_dna(rng, 120000)
n_blocks = 500_009
filters = []
for i, bins in enumerate((122, 43, 29, 49)):
    W = (bins + 63) // 64
    d = capi.DeviceIBF.create(0, bins, 3, 13, W * 64 * n_blocks)
    d.add_sequence(ref[i * 5000:i * 5000 + 6000], 300)
    filters.append(d)
reads = []
for i in range(3000):
    L = int(rng.integers(60, 250))
    p = int(rng.integers(0, len(ref) - L))
    reads.append(ref[p:p + L] if i % 3 else H.random_dna(rng, L))
buf, offs, lens = H.pack_reads(reads)
stop = threading.Event()
errors = []
calls = [0] * n_threads
creations = [0] * n_threads


def worker(t):
    try:
        last = None
        eng = None
        while not stop.is_set():
            if eng is None or calls[t] % 40 == 39:  # engines come and go: the registry hands the copy out again and again
                if eng is not None:
                    eng.destroy()
                eng = capi.Engine(0, filters[:1], filters[1:])
                creations[t] += 1
            mc = eng.classify(buf, offs, lens)[0].astype(np.int32)
            if last is not None and (mc < last).any():
                errors.append("thread %d: a maximum went down while sequences were only added" % t)
                break
            last = mc
            calls[t] += 1
        if eng is not None:
            eng.destroy()
    except Exception as ex:  # noqa: BLE001
        errors.append("thread %d: %r" % (t, ex))


threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
for th in threads:
    th.start()
t0 = time.time()
inserts = 0
while time.time() - t0 < seconds and not errors:
    i = inserts % 4
    p = 30000 + (inserts * 1500) % 80000
    filters[i].add_sequence(ref[p:p + 1500], 300)
    inserts += 1
    time.sleep(0.03)
stop.set()
for th in threads:
    th.join()
views, keep = [], []
for d in filters:
    h = d.download()
    keep.append(h)
    views.append(po.OracleIBF.wrap(h.info["n_bins"], 3, 13, h.info["n_bits"], h.words()))
exp = np.stack([po.batch_raw_max(v, buf, offs, lens, 8) for v in views], axis=1)
final = []
for t in range(3):
    e = capi.Engine(0, filters[:1], filters[1:])
    final.append(np.array_equal(e.classify(buf, offs, lens)[0], exp))
    e.destroy()
print("soak_shared_merged: %.1f s, %d threads, %d classify calls, %d engines created, %d inserts; errors: %s; final state equals the oracle: %s"
      % (seconds, n_threads, sum(calls), sum(creations), inserts, errors or "none", all(final)))
sys.exit(0 if (not errors and all(final)) else 1)

#!/usr/bin/env python3
"""N engines on ONE GPU at throughput scale -- the reference's concurrency model (N classification threads behind one queue,
src/main/adaptive_sampling.hpp:745-751) through the mirror's "one engine per calling thread" (INTEGRATION.md section 2).

K = 1, 2, 4 host threads, one engine each, all engines borrowing the SAME rb_dibf filters; every thread classifies its own
batches of 65 536 reads.  Shapes:
  readme           the README benchmark shape (1 deplete + 3 target narrow filters): merged table, one gather per lookup
  readme_unmerged  the same filters with rb_engine_set_merge(0): four clock-phased launches per call, each assuming it owns the L2
  c4               deplete = 8 GiB GRCh38-scale filter + target = 600-bin filter (wide kernels, HBM bound)
Two forms per shape: `device` (reads resident in HBM, rb_classify_batch_device on the engine's own stream -- kernels only) and
`host` (rb_classify_batch: pinned-free host buffers in, results back -- what a calling thread of the reference would do).
Prints aggregate reads/s per K and the ratio to K = 1; outputs of every thread are compared with the K = 1 outputs.

  python3 profiles/engines_on_one_gpu.py [--shapes readme,readme_unmerged,c4] [--k 1,2,4] [--forms device,host]
                                         [--batch 65536] [--batches 24] [--arbiter 0|1]
"""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from readbouncer_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="readme,readme_unmerged,c4")
ap.add_argument("--k", default="1,2,4")
ap.add_argument("--forms", default="device,host")
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--batches", type=int, default=24)
ap.add_argument("--read-len", type=int, default=0)
ap.add_argument("--host-slice-mib", type=float, default=0.0, help="host form: rb_engine_set_host_slice_bytes (0: the engine's default)")
args = ap.parse_args()

import torch  # noqa: E402  (device memory for the resident form)

dev = torch.device("cuda:0")
SEEDS = {"c3": (4, 40), "zymo": (6, 60), "mock_deplete": (11, 110), "mock_t1": (12, 111), "mock_t2": (13, 112), "mock_t3": (14, 113)}
_filters = {}


def filt(key):
    if key not in _filters:
        fs, ps = SEEDS[key]
        _filters[key] = synth.build_device_filter(0, synth.WORKLOADS[key], fill_seed=fs, plant_seed=ps,
                                                  n_segments=512 if key.startswith("mock_") else 2048)
    return _filters[key]


def shape(name):
    if name.startswith("readme"):
        dep, tgt, L = ["mock_deplete"], ["mock_t1", "mock_t2", "mock_t3"], 250
    else:
        dep, tgt, L = ["c3"], ["zymo"], 360
    L = args.read_len or L
    d = [filt(k)[0] for k in dep]
    t = [filt(k)[0] for k in tgt]
    ref = np.concatenate([filt(k)[1] for k in dep + tgt])
    return d, t, ref, L


def run(name, form, K, expect):
    d, t, ref, L = shape(name)
    n, nb = args.batch, args.batches
    engines, work = [], []
    for k in range(K):
        e = capi.Engine(0, d, t)
        if name == "readme_unmerged":
            e.set_merge(0)
        if args.host_slice_mib:
            e.set_host_slice_bytes(int(args.host_slice_mib * (1 << 20)))
        engines.append(e)
        # every thread has the same reads (seed 99): outputs must equal the K = 1 run whatever ran beside them
        t_seq, t_off, t_len = synth.make_reads_device(99, n, L, ref, dev)
        if form == "device":
            t_max = torch.zeros((n, len(d) + len(t)), dtype=torch.int16, device=dev)
            t_dec = torch.zeros(n, dtype=torch.uint8, device=dev)
            work.append((t_seq, t_off, t_len, t_max, t_dec))
        else:
            buf = t_seq.cpu().numpy()
            work.append((buf, np.arange(n, dtype=np.uint64) * np.uint64(L), np.full(n, L, dtype=np.uint32)))
    torch.cuda.synchronize()
    outs = [None] * K

    def call(k):
        e, w = engines[k], work[k]
        if form == "device":
            e.classify_device(w[0].data_ptr(), w[1].data_ptr(), w[2].data_ptr(), n, L, d_maxcount=w[3].data_ptr(), d_decision=w[4].data_ptr())
            return None
        return e.classify(w[0], w[1], w[2])

    def worker(k):
        for _ in range(nb):
            r = call(k)
        outs[k] = r if form == "host" else (work[k][3].cpu().numpy().view(np.uint16), None, work[k][4].cpu().numpy())

    for k in range(K):
        call(k)
        call(k)  # merged copies, threshold tables, code objects
    th = [threading.Thread(target=worker, args=(k,)) for k in range(K)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    wall = time.perf_counter() - t0
    for k in range(K):
        if expect is not None:
            assert np.array_equal(outs[k][0], expect[0]) and np.array_equal(outs[k][2], expect[2]), (name, form, K, k)
    for e in engines:
        e.destroy()
    return K * nb * n / wall, (outs[0][0].copy(), None, outs[0][2].copy())


for name in args.shapes.split(","):
    for form in args.forms.split(","):
        base, expect = None, None
        for K in [int(x) for x in args.k.split(",")]:
            rate, out = run(name, form, K, expect)
            if expect is None:
                expect = out
            base = base or rate
            print("%-16s %-6s K=%d  %7.2f M reads/s aggregate   %.2f x K=1   (batches of %d reads, %d per thread; outputs equal)"
                  % (name, form, K, rate / 1e6, rate / base, args.batch, args.batches), flush=True)

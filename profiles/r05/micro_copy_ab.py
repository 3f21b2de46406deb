"""A/B of the micro-batch's way into HBM: the runtime's copy (a blit kernel, 11 us) against the engine's own copy kernel (RB_MICRO_COPY_KERNEL_BYTES,
a measurement switch under RB_TUNING_ENV=1; default 65536).  One process per setting; config 4's filters; host-to-host latency by batch size
and the config 5 replay."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from readbouncer_amd import capi, synth
dep, rd = synth.build_device_filter(0, synth.WORKLOADS["c3"], fill_seed=4, plant_seed=40)
tgt, rt = synth.build_device_filter(0, synth.WORKLOADS["zymo"], fill_seed=6, plant_seed=60)
L = 360
N = 300_000
seqs, _, _ = synth.make_reads_device(7000, N, L, np.concatenate([rd, rt]), torch.device("cuda:0"))
buf = seqs.cpu().numpy(); del seqs
offs = np.arange(N, dtype=np.uint64) * np.uint64(L); lens = np.full(N, L, dtype=np.uint32)
eng = capi.Engine(0, [dep], [tgt])
tag = os.environ.get("RB_MICRO_COPY_KERNEL_BYTES", "default")
import hashlib
for n in (1, 64, 256, 512, 1024, 2048):
    sub = np.ascontiguousarray(buf[: n * L]); so, sl = offs[:n].copy(), lens[:n].copy()
    for _ in range(30): out = eng.classify(sub, so, sl)
    ts = []
    for _ in range(500):
        a = time.perf_counter(); eng.classify(sub, so, sl); ts.append((time.perf_counter() - a) * 1e6)
    ts = np.sort(ts)
    print("%s n=%4d  p50 %.1f us  p99 %.1f us  sha %s" % (tag, n, ts[250], ts[494], hashlib.sha1(b"".join(x.tobytes() for x in out)).hexdigest()[:10]), flush=True)
rate, seconds = 150000.0, 2.0
n = int(rate * seconds)
arrival = np.cumsum(np.random.default_rng(7).exponential(1.0 / rate, size=n))
for rep in range(2):
    dec, lat, calls, service, elapsed = eng.replay_arrivals(buf[: n * L], L, arrival, max_batch=16384)
    print("%s c5 replay  p50 %.1f us  p99 %.1f us  p99.9 %.1f us  max %.1f us  mean batch %.1f  service p50 %.1f us  decisions %s"
          % (tag, np.percentile(lat, 50) * 1e6, np.percentile(lat, 99) * 1e6, np.percentile(lat, 99.9) * 1e6, lat.max() * 1e6, calls.mean(), np.percentile(service, 50) * 1e6, np.bincount(dec, minlength=3).tolist()), flush=True)

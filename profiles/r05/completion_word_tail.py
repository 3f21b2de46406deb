"""The tail of the completion-word path: rb_engine_set_completion_word modes 0 (stream wait in every call), 1 (word; the stream is waited for
every 256th call), n > 1 (that period instead; 2147483647 = never).  (An earlier version of the library also had "word + hipStreamQuery / hipStreamSynchronize
at the start of every next call": both cost what the word saves -- one read: stream wait 40.6, word 36.1, word + query 41.8, word + synchronise 45.4 us; the numbers are in
negative_results.md, entry 12.)  Per mode and batch size: p50 / p90 / p99 / p99.9 / max over 4 000 calls, how many
calls took more than p50 + 8 us and the gaps (in calls) between them; then the config 5 replay per mode, three times."""
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
MODES = [int(x) for x in os.environ.get("MODES", "0,2147483647,1,128").split(",")]
for n in (1, 8, 64, 256):
    sub = np.ascontiguousarray(buf[: n * L]); so, sl = offs[:n].copy(), lens[:n].copy()
    for mode in MODES * 2:
        eng.set_completion_word(mode)
        for _ in range(50): eng.classify(sub, so, sl)
        ts = np.empty(4000)
        for i in range(4000):
            a = time.perf_counter(); eng.classify(sub, so, sl); ts[i] = (time.perf_counter() - a) * 1e6
        p = np.percentile(ts, [50, 90, 99, 99.9])
        slow = np.nonzero(ts > p[0] + 8)[0]
        gaps = np.diff(slow)
        print("n=%4d mode %d  p50 %6.1f  p90 %6.1f  p99 %6.1f  p99.9 %6.1f  max %7.1f us   slow calls %4d  gaps median %s  first gaps %s" % (
            n, mode, p[0], p[1], p[2], p[3], ts.max(), len(slow), np.median(gaps) if len(gaps) else "-", gaps[:12].tolist()), flush=True)
rate, seconds = 150000.0, 2.0
n = int(rate * seconds)
arrival = np.cumsum(np.random.default_rng(7).exponential(1.0 / rate, size=n))
for rep in range(3):
    for mode in MODES:
        eng.set_completion_word(mode)
        dec, lat, calls, service, elapsed = eng.replay_arrivals(buf[: n * L], L, arrival, max_batch=16384)
        print("c5 replay mode %d  p50 %.1f  p99 %.1f  p99.9 %.1f  max %.1f us  mean batch %.1f  service p50 %.1f p99 %.1f us" % (
            mode, np.percentile(lat, 50) * 1e6, np.percentile(lat, 99) * 1e6, np.percentile(lat, 99.9) * 1e6, lat.max() * 1e6, calls.mean(), np.percentile(service, 50) * 1e6, np.percentile(service, 99) * 1e6), flush=True)

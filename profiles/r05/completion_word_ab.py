"""A/B of how the host learns that a micro-batch is done: hipStreamSynchronize against a word of page-locked memory that the call's last
kernel stores (rb_engine_set_completion_word), in one process, settings alternated.  Config 3's deplete filter alone and config 4
(deplete + target); host-to-host latency of rb_classify_batch by batch size, the SHA-1 of everything a call returns, the config 5 replay
and the live step in both settings."""
import hashlib, os, sys, time
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
for name, targets in (("deplete only", []), ("deplete + target (c4)", [tgt])):
    eng = capi.Engine(0, [dep], targets)
    for n in (1, 8, 64, 256, 512, 1024, 2048):
        sub = np.ascontiguousarray(buf[: n * L]); so, sl = offs[:n].copy(), lens[:n].copy()
        res = {}
        for rnd in range(2):
            for word in (0, 1):
                eng.set_completion_word(word)
                for _ in range(30): out = eng.classify(sub, so, sl)
                ts = []
                for _ in range(400):
                    a = time.perf_counter(); eng.classify(sub, so, sl); ts.append((time.perf_counter() - a) * 1e6)
                res.setdefault(word, []).extend(ts)
                res[("sha", word)] = hashlib.sha1(b"".join(x.tobytes() for x in out)).hexdigest()[:10]
        p = {f: np.percentile(res[f], [50, 99]) for f in (0, 1)}
        print("%-22s n=%4d  stream wait p50 %6.1f p99 %6.1f us | completion word p50 %6.1f p99 %6.1f us  (%+.1f us)  sha %s %s" % (
            name, n, p[0][0], p[0][1], p[1][0], p[1][1], p[1][0] - p[0][0], res[("sha", 0)], "same" if res[("sha", 0)] == res[("sha", 1)] else "DIFFERENT " + res[("sha", 1)]), flush=True)
    if targets:
        rate, seconds = 150000.0, 2.0
        n = int(rate * seconds)
        arrival = np.cumsum(np.random.default_rng(7).exponential(1.0 / rate, size=n))
        for word in (0, 1, 0, 1):
            eng.set_completion_word(word)
            dec, lat, calls, service, elapsed = eng.replay_arrivals(buf[: n * L], L, arrival, max_batch=16384)
            print("c5 replay word=%d  p50 %.1f us  p99 %.1f us  p99.9 %.1f us  max %.1f us  mean batch %.1f  service p50 %.1f us  decisions %s"
                  % (word, np.percentile(lat, 50) * 1e6, np.percentile(lat, 99) * 1e6, np.percentile(lat, 99.9) * 1e6, lat.max() * 1e6, calls.mean(), np.percentile(service, 50) * 1e6, np.bincount(dec, minlength=3).tolist()), flush=True)
    del eng

"""A/B of who decides for a micro-batch: the decision kernel behind the latency form of K1 (two dependent launches) against K1 itself
(rb_engine_set_fold_decide: the workgroup that writes a read's last raw maximum decides for it).  In one process, alternating, on
three filter sets: config 3's deplete filter alone, config 4 (deplete + one target), deplete + three targets.  Host-to-host latency of
rb_classify_batch by batch size, the SHA-1 of everything a call returns, and the config 5 replay in both settings."""
import hashlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from readbouncer_amd import capi, synth
dep, rd = synth.build_device_filter(0, synth.WORKLOADS["c3"], fill_seed=4, plant_seed=40)
tgts = [synth.build_device_filter(0, synth.WORKLOADS["zymo"], fill_seed=6 + i, plant_seed=60 + i) for i in range(3)]
L = 360
N = 300_000
seqs, _, _ = synth.make_reads_device(7000, N, L, np.concatenate([rd] + [t[1] for t in tgts]), torch.device("cuda:0"))
buf = seqs.cpu().numpy(); del seqs
offs = np.arange(N, dtype=np.uint64) * np.uint64(L); lens = np.full(N, L, dtype=np.uint32)
for name, nt in (("deplete only", 0), ("deplete + target (c4)", 1), ("deplete + 3 targets", 3)):
    eng = capi.Engine(0, [dep], [t[0] for t in tgts[:nt]])
    for n in (1, 8, 64, 256, 1024, 2048):
        sub = np.ascontiguousarray(buf[: n * L]); so, sl = offs[:n].copy(), lens[:n].copy()
        res = {}
        for rnd in range(2):  # alternate, so that neither setting owns the warm or the cold half
            for fold in (0, 1):
                eng.set_fold_decide(fold)
                for _ in range(30): out = eng.classify(sub, so, sl)
                ts = []
                for _ in range(400):
                    a = time.perf_counter(); eng.classify(sub, so, sl); ts.append((time.perf_counter() - a) * 1e6)
                res.setdefault(fold, []).extend(ts)
                res[("sha", fold)] = hashlib.sha1(b"".join(x.tobytes() for x in out)).hexdigest()[:10]
        p = {f: np.percentile(res[f], [50, 99]) for f in (0, 1)}
        print("%-22s n=%4d  two launches p50 %6.1f p99 %6.1f us | folded p50 %6.1f p99 %6.1f us  (%+.1f us)  sha %s %s" % (
            name, n, p[0][0], p[0][1], p[1][0], p[1][1], p[1][0] - p[0][0], res[("sha", 0)], "same" if res[("sha", 0)] == res[("sha", 1)] else "DIFFERENT " + res[("sha", 1)]), flush=True)
    if nt == 1:
        rate, seconds = 150000.0, 2.0
        n = int(rate * seconds)
        arrival = np.cumsum(np.random.default_rng(7).exponential(1.0 / rate, size=n))
        for fold in (0, 1, 0, 1):
            eng.set_fold_decide(fold)
            dec, lat, calls, service, elapsed = eng.replay_arrivals(buf[: n * L], L, arrival, max_batch=16384)
            print("c5 replay fold=%d  p50 %.1f us  p99 %.1f us  p99.9 %.1f us  max %.1f us  mean batch %.1f  service p50 %.1f us  decisions %s"
                  % (fold, np.percentile(lat, 50) * 1e6, np.percentile(lat, 99) * 1e6, np.percentile(lat, 99.9) * 1e6, lat.max() * 1e6, calls.mean(), np.percentile(service, 50) * 1e6, np.bincount(dec, minlength=3).tolist()), flush=True)
    del eng

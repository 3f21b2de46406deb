import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from readbouncer_amd import capi, synth
for wl, seeds in (("c2", (2, 20)), ("c3", (4, 40))):
    d, ref = synth.build_device_filter(0, synth.WORKLOADS[wl], *seeds)
    buf, offs, lens = synth.make_reads(3, 65536, 360, ref)
    eng = capi.Engine(0, [d], [])
    for thr in (2048, 8192, 32768):
        eng.set_split_threshold(thr)
        row = []
        for n in (512, 1024, 2048, 4096, 8192, 16384, 65536):
            sub = np.ascontiguousarray(buf[: n * 360]); so, sl = offs[:n].copy(), lens[:n].copy()
            for _ in range(5): eng.classify(sub, so, sl)
            ts = []
            for _ in range(40):
                a = time.perf_counter(); eng.classify(sub, so, sl); ts.append((time.perf_counter() - a) * 1e6)
            row.append("%d:%.0fus(%.1fM/s)" % (n, np.median(ts), n / np.median(ts)))
        print(wl, "split_threshold", thr, " ".join(row), flush=True)

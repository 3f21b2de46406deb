import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from readbouncer_amd import capi, synth
dev = torch.device("cuda:0")
d, ref = synth.build_device_filter(0, synth.WORKLOADS["c3"], fill_seed=4, plant_seed=40)
print("source placement", d.placement(), flush=True)
n, L = 1_000_000, 360
seqs, offs, lens = synth.make_reads_device(1234, n, L, ref, dev)
buf = seqs.cpu().numpy()
del seqs
o = np.arange(n, dtype=np.uint64) * np.uint64(L); l = np.full(n, L, dtype=np.uint32)
byts = synth.algorithmic_bytes_per_read(L, [(8192, 13, 3)])
for rep in range(4):
    for tries in (5, 1):
        capi.set_placement_tries(tries)
        pool = capi.Pool.from_device([0], [d], [])
        pool.classify(buf, o, l)
        pool.set_timing(True)
        for _ in range(3): pool.classify(buf, o, l)
        kt = pool.kernel_time()
        print("pool replica with %d tries: K1 %.1f ms in %d launches = %.0f GB/s" % (tries, kt[0][0], kt[0][1], byts * 3 * n / (kt[0][0] / 1e3) / 1e9), flush=True)
        pool.destroy()

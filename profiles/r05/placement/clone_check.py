import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from readbouncer_amd import capi, synth
dev = torch.device("cuda:0")
d, ref = synth.build_device_filter(0, synth.WORKLOADS["c3"], fill_seed=4, plant_seed=40)
print("source placement", d.placement(), flush=True)
N, L = 2_000_000, 360
seqs, offs, lens = synth.make_reads_device(1234, N, L, ref, dev)
mc = torch.zeros((N, 1), dtype=torch.int16, device=dev)
def k1(f, label):
    eng = capi.Engine(0, [f], [])
    eng.set_timing(True)
    for it in range(5):
        if it == 2: eng.kernel_time()
        eng.classify_device(seqs.data_ptr(), offs.data_ptr(), lens.data_ptr(), N, L, d_maxcount=mc.data_ptr())
    torch.cuda.synchronize()
    ms, calls = eng.kernel_time()
    print("%s: K1 %.2f ms per 2 M reads, probe now %.0f GB/s" % (label, ms / calls, f.probe_read_peak(1024, True, 24, target_ms=60.0)[0]), flush=True)
    eng.destroy()
k1(d, "source")
for i in range(3):
    c, peer, secs = d.clone_to_ex(0)
    print("clone %d placement" % i, c.placement(), "copy s %.2f" % secs, flush=True)
    k1(c, "clone %d" % i)
    k1(d, "source again")
    c.free()
# pool over the same filter: per-device K1 rate
pool = capi.Pool.from_device([0], [d], [])
print("pool created", flush=True)
n = 1_000_000
buf = seqs[: n * L].cpu().numpy()
o = np.arange(n, dtype=np.uint64) * np.uint64(L); l = np.full(n, L, dtype=np.uint32)
pool.classify(buf, o, l)
pool.set_timing(True)
for _ in range(3): pool.classify(buf, o, l)
kt = pool.kernel_time()
byts = synth.algorithmic_bytes_per_read(L, [(8192, 13, 3)])
print("pool K1:", kt, "GB/s", byts * 3 * n / (kt[0][0] / 1e3) / 1e9, flush=True)
capi.set_placement_tries(1)
pool2 = capi.Pool.from_device([0], [d], [])
pool2.classify(buf, o, l)
pool2.set_timing(True)
for _ in range(3): pool2.classify(buf, o, l)
kt = pool2.kernel_time()
print("pool (replica without trial) K1:", kt, "GB/s", byts * 3 * n / (kt[0][0] / 1e3) / 1e9, flush=True)

"""K1 rate by launch size on several plain allocations of one table (placement trials off): do the allocations that probe fast (steady
state) also serve SMALL launches faster?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from readbouncer_amd import capi, synth
key = sys.argv[1] if len(sys.argv) > 1 else "c3np2"
capi.set_placement_tries(1)
dev = torch.device("cuda:0")
w = synth.WORKLOADS[key]
n, L = 2_000_000, 360
cands = []
for i in range(6):
    d, ref = synth.build_device_filter(0, w, fill_seed=4, plant_seed=40)
    cands.append(d)
seqs, offs, lens = synth.make_reads_device(1234, n, L, ref, dev)
mc = torch.zeros((n, 1), dtype=torch.int16, device=dev)
byts = synth.algorithmic_bytes_per_read(L, [(w["n_bins"], w["k"], w["h"])])
sizes = (4096, 16384, 65536, 262144, 2_000_000)
print("allocation: probe GB/s | K1 GB/s at launches of " + " / ".join(str(s) for s in sizes))
for i, d in enumerate(cands):
    g = max(d.probe_read_peak(1024 if key != "grch38_f100k" else 4096, True, 24, target_ms=60.0)[0] for _ in range(2))
    eng = capi.Engine(0, [d], [])
    eng.set_timing(True)
    row = []
    for m in sizes:
        reps = max(3, min(40, 400_000 // m))
        for it in range(2):
            eng.kernel_time()
            for s in range(reps):
                lo = (s * m) % (n - m + 1)
                eng.classify_device(seqs.data_ptr(), offs[lo:].data_ptr(), lens[lo:].data_ptr(), m, L, d_maxcount=mc[lo:].data_ptr())
            torch.cuda.synchronize()
            ms, calls = eng.kernel_time()
        row.append(byts * reps * m / (ms / 1e3) / 1e9)
    print("allocation %d at 0x%x: probe %.0f | %s" % (i, d.device_words(), g, " / ".join("%.0f" % x for x in row)), flush=True)
    eng.destroy()

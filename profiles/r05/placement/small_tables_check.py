import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from readbouncer_amd import capi, synth
capi.set_placement_tries(1)
dev = torch.device("cuda:0")
for key, row in (("c2", 128), ("zymo", 128)):
    w = synth.WORKLOADS[key]
    n, L = 1_000_000, 360
    cands = []
    for i in range(8):
        d, ref = synth.build_device_filter(0, w, fill_seed=2, plant_seed=20)
        cands.append(d)
    seqs, offs, lens = synth.make_reads_device(1234, n, L, ref, dev)
    mc = torch.zeros((n, 1), dtype=torch.int16, device=dev)
    byts = synth.algorithmic_bytes_per_read(L, [(w["n_bins"], w["k"], w["h"])])
    for rep in range(2):
        for i, d in enumerate(cands):
            g = max(d.probe_read_peak(row, False, 24, target_ms=40.0)[0] for _ in range(2))
            eng = capi.Engine(0, [d], [])
            eng.set_timing(True)
            for it in range(8):
                if it == 3: eng.kernel_time()
                eng.classify_device(seqs.data_ptr(), offs.data_ptr(), lens.data_ptr(), n, L, d_maxcount=mc.data_ptr())
            torch.cuda.synchronize()
            ms, calls = eng.kernel_time()
            print("%s allocation %d at 0x%x: probe %.0f GB/s | K1 %.3f ms = %.4f of 8 TB/s" % (key, i, d.device_words(), g, ms / calls, byts * n / (ms / calls / 1e3) / 8e12), flush=True)
            eng.destroy()
    for d in cands: d.free()

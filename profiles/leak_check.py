import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from readbouncer_amd import capi, synth
from tests import helpers as H
rng = np.random.default_rng(1)
ref = H.random_dna(rng, 20000)
reads = [ref[i*50:i*50+400] for i in range(100)]
buf, offs, lens = H.pack_reads(reads)
def once(big):
    d = capi.DeviceIBF.create(0, 2500 if big else 64, 3, 13, (40 if big else 1) * 64 * 200003)
    d.add_sequence(ref, 500)
    t = capi.DeviceIBF.create(0, 100, 3, 13, 128 * 50021)
    eng = capi.Engine(0, [d], [t])
    eng.classify(buf, offs, lens)
    eng.classify(buf, offs[:3], lens[:3])
    big_buf = np.tile(buf, 300); big_offs = np.concatenate([offs + np.uint64(i * len(buf)) for i in range(300)]); big_lens = np.tile(lens, 300)
    eng.classify(big_buf, big_offs, big_lens)
    live = capi.Live(eng) if hasattr(capi, "Live") else None
    del live, eng, d, t
free0 = None
for it in range(60):
    once(it % 2 == 0)
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if it == 4: free0 = free
    if it % 10 == 9: print(it, "free MB", free >> 20, flush=True)
print("drift MB since iteration 4:", (free0 - free) >> 20)

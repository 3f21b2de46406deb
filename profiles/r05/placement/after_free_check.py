"""Is the slow stretch after a placement trial a transient?  K1 rate of 30 k-read launches on a trial-placed clone, measured in consecutive
stretches of 33 launches right after the clone was made (the trial frees up to four 8.6 GB candidates just before)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from readbouncer_amd import capi, synth
dev = torch.device("cuda:0")
d, ref = synth.build_device_filter(0, synth.WORKLOADS["c3"], fill_seed=4, plant_seed=40)
n, L = 1_000_000, 360
seqs, offs, lens = synth.make_reads_device(1234, n, L, ref, dev)
byts = synth.algorithmic_bytes_per_read(L, [(8192, 13, 3)])
mc = torch.zeros((n, 1), dtype=torch.int16, device=dev)
m = n // 33
for tries in (5, 1, 5, 1):
    capi.set_placement_tries(tries)
    t0 = time.time()
    c, _, _ = d.clone_to_ex(0)
    eng = capi.Engine(0, [c], [])
    eng.set_timing(True)
    out = []
    for stretch in range(8):
        eng.kernel_time()
        for s in range(33):
            eng.classify_device(seqs.data_ptr(), offs[s * m:].data_ptr(), lens[s * m:].data_ptr(), m, L, d_maxcount=mc[s * m:].data_ptr())
        torch.cuda.synchronize()
        ms, calls = eng.kernel_time()
        out.append("%.2fs:%.0f" % (time.time() - t0, byts * 33 * m / (ms / 1e3) / 1e9))
    print("clone with %d tries %s: seconds since the clone call : GB/s  %s" % (tries, c.placement(), "  ".join(out)), flush=True)
    eng.destroy()
    c.free()
    time.sleep(2.0)

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
o = np.arange(n, dtype=np.uint64) * np.uint64(L); l = np.full(n, L, dtype=np.uint32)
byts = synth.algorithmic_bytes_per_read(L, [(8192, 13, 3)])
mc = torch.zeros((n, 1), dtype=torch.int16, device=dev)
for rep in range(3):
    for tries in (5, 1):
        capi.set_placement_tries(tries)
        c, _, _ = d.clone_to_ex(0)
        eng = capi.Engine(0, [c], [])
        eng.set_timing(True)
        # (a) device-resident, 33 launches of ~30 k reads each (a pool call's slices)
        m = n // 33
        for it in range(2):
            eng.kernel_time()
            for s in range(33):
                eng.classify_device(seqs.data_ptr(), offs[s * m:].data_ptr(), lens[s * m:].data_ptr(), m, L, d_maxcount=mc[s * m:].data_ptr())
            torch.cuda.synchronize()
            ms, calls = eng.kernel_time()
        ra = byts * 33 * m / (ms / 1e3) / 1e9
        # (b) host buffers through rb_classify_batch (PCIe slices)
        eng.classify(buf, o, l)
        eng.kernel_time()
        for _ in range(3): eng.classify(buf, o, l)
        ms2, calls2 = eng.kernel_time()
        rb_ = byts * 3 * n / (ms2 / 1e3) / 1e9
        # (c) one launch
        eng.kernel_time()
        for _ in range(3): eng.classify_device(seqs.data_ptr(), offs.data_ptr(), lens.data_ptr(), n, L, d_maxcount=mc.data_ptr())
        torch.cuda.synchronize()
        ms3, calls3 = eng.kernel_time()
        rc = byts * 3 * n / (ms3 / 1e3) / 1e9
        print("clone with %d tries %s: 33 device launches %.0f GB/s | host batch (%d launches) %.0f GB/s | one launch %.0f GB/s" % (tries, c.placement(), ra, calls2 // 3, rb_, rc), flush=True)
        eng.destroy()
        c.free()

#!/usr/bin/env python3
"""Config 4 (deplete 8 GiB + 600-bin target, check_unblock, 2 M reads): does the load policy or the launch arrangement of the TARGET's
count kernel matter beside the NT-streaming deplete kernel?  (VERDICT r4 "Next" #6.)  K1 time per call for: the engine's default
(deplete non-temporal, target temporal: rb_engine_set_nt_threshold 512 MiB), both non-temporal, both temporal, and the two kernels one
after the other instead of side by side (rb_engine_set_overlap 0).  Results are identical by construction; the SHA-1 says so."""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from readbouncer_amd import capi, synth  # noqa: E402

dev = torch.device("cuda:0")
dep, rd = synth.build_device_filter(0, synth.WORKLOADS["c3"], fill_seed=4, plant_seed=40)
tgt, rt = synth.build_device_filter(0, synth.WORKLOADS["zymo"], fill_seed=6, plant_seed=60)
import numpy as np  # noqa: E402
N, L = 2_000_000, 360
seqs, offs, lens = synth.make_reads_device(1000, N, L, np.concatenate([rd, rt]), dev)
mc = torch.zeros((N, 2), dtype=torch.int16, device=dev)
dec = torch.zeros(N, dtype=torch.uint8, device=dev)
ref = None
for name, nt, overlap in (("default: deplete NT, target temporal, side by side", 512 << 20, 1), ("both NT", 64 << 20, 1), ("both temporal", 1 << 40, 1),
                          ("default policy, one after the other", 512 << 20, 0), ("default again", 512 << 20, 1)):
    eng = capi.Engine(0, [dep], [tgt])
    eng.set_nt_threshold(nt)
    eng.set_overlap(bool(overlap))
    eng.set_timing(True)
    for it in range(8):
        if it == 2:
            eng.kernel_time()
        eng.classify_device(seqs.data_ptr(), offs.data_ptr(), lens.data_ptr(), N, L, d_maxcount=mc.data_ptr(), d_decision=dec.data_ptr())
    torch.cuda.synchronize()
    ms, calls = eng.kernel_time()
    sha = hashlib.sha1(mc.cpu().numpy().tobytes() + dec.cpu().numpy().tobytes()).hexdigest()[:12]
    ref = ref or sha
    assert sha == ref
    per = ms / 6
    byts = synth.algorithmic_bytes_per_read(L, [(8192, 13, 3), (600, 13, 3)]) * N
    lines = N * 2 * 348 * 3 * (8 + 1)
    print("%-52s K1 %7.2f ms per 2 M reads  %.4f of 8 TB/s  %.2f G lines/s  sha %s" % (name, per, byts / (per / 1e3) / 8e12, lines / per / 1e6, sha), flush=True)
    eng.destroy()

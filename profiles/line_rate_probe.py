#!/usr/bin/env python3
"""What the chip delivers for RANDOM SINGLE 128-byte lines (the miss stream of the narrow-filter kernels) as a function of the
table size: rb_dibf_probe_read_peak with 128-byte rows -- no compute, 12 / 24 loads in flight per wave, 8 waves per SIMD -- on
tables of 10 MB ... 1 GB.  The README shape's merged table is 40 MB; its phased kernel makes 41.6 G line requests/s to the fabric.

  python3 profiles/line_rate_probe.py [sizes in MB, comma separated]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from readbouncer_amd import capi  # noqa: E402

print("table MB | G lines/s (12 in flight) | (24 in flight) | non-temporal, 24 in flight")
for mb in [float(x) for x in sys.argv[1].split(',')] if len(sys.argv) > 1 else (4, 10, 20, 40, 80, 160, 400, 1000, 4000):
    n_blocks = int(mb * 1000 * 1000) // 32
    d = capi.DeviceIBF.create(0, 256, 3, 13, 256 * n_blocks)
    d.fill_synth(3)
    row = []
    for nt, b in ((0, 12), (0, 24), (1, 24)):
        best = 0.0
        for _ in range(3):
            g, ms = d.probe_read_peak(128, nt, b, 0, 100.0)
            best = max(best, g)
        row.append(best / 128.0)
    print("%8.1f | %6.1f | %6.1f | %6.1f" % (mb, row[0], row[1], row[2]), flush=True)
    d.free()

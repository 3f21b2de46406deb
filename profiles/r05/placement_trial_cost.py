"""What the placement trial costs at load time, piece by piece: rb_dibf_create of config 3's 8 GiB table and of config 3 at the reference's sizing
(4.4 GiB) with 1..5 candidates, and the release of the filter.  (rb_dibf_create zeroes the table and waits for the device to settle right after the
trial; rb_dibf_open fills the table first and waits afterwards: profiles/load_throughput.py.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from readbouncer_amd import capi, synth
for wl in ("c3", "c3np2"):
    w = synth.WORKLOADS[wl]
    for tries in (1, 1, 2, 3, 5, 5):
        capi.set_placement_tries(tries)
        t = time.perf_counter()
        d = capi.DeviceIBF.create(0, w["n_bins"], w["h"], w["k"], synth.filter_bits(w))
        t1 = time.perf_counter() - t
        pl = d.placement()
        t = time.perf_counter(); del d; t2 = time.perf_counter() - t
        print("%-6s tries %d: create %.2f s  (probed %d, kept %.0f GB/s, worst %.0f)  free %.3f s" % (wl, tries, t1, pl[0], pl[1], pl[2], t2), flush=True)

#!/usr/bin/env python3
"""How long does it take to stand up the one-process pool (rb_pool) for a GRCh38-scale filter?
  (a) rb_pool_create: host image -> one PCIe upload per device (needs the 8 GiB image on the host)
  (b) rb_pool_create_from_files: the .ibf streamed once into device 0, then device-to-device copies to the others, all
      at once (xGMI between peers; on a one-GPU box the device list repeats device 0 and the copies are HBM -> HBM)
Usage: python3 profiles/pool_replication.py [n_devices] [GiB]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from readbouncer_amd import capi  # noqa: E402

n_dev = int(sys.argv[1]) if len(sys.argv) > 1 else 4
gib = int(sys.argv[2]) if len(sys.argv) > 2 else 8
real = capi.device_count()
devices = list(range(n_dev)) if real >= n_dev else [0] * n_dev
d = capi.DeviceIBF.create(0, 8192, 3, 13, gib << 33)
d.fill_synth(4)
t0 = time.time()
img = d.download()
t_down = time.time() - t0
path = "/dev/shm/rb_pool_probe.ibf"
t0 = time.time()
img.store(path)
t_store = time.time() - t0
d.free()
print("filter: 8192 bins, %d GiB; devices %s (%d real); download %.2f s, store to %s %.2f s" % (gib, devices, real, t_down, path, t_store))
t0 = time.time()
p = capi.Pool(devices, [img], [])
t_img = time.time() - t0
p.destroy()
img.close()
print("(a) rb_pool_create from the host image: %.2f s (%d PCIe uploads of %d GiB)" % (t_img, n_dev, gib))
t0 = time.time()
p = capi.Pool.from_files(devices, [path], [])
t_files = time.time() - t0
print("(b) rb_pool_create_from_files: %.2f s in all, of which %.3f s for the %d device-to-device copies (%.0f GB/s aggregate)"
      % (t_files, p.replication_seconds, n_dev - 1, (n_dev - 1) * gib * 1.0737 / max(p.replication_seconds, 1e-9)))
p.destroy()
os.remove(path)

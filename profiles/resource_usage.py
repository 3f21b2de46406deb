#!/usr/bin/env python3
"""Register/LDS/occupancy table of every kernel in rb_kernels.hip, from hipcc's own -Rpass-analysis=kernel-resource-usage
remarks (cross-compiles without a GPU).  Usage: python3 profiles/resource_usage.py > profiles/r02/kernel_resource_usage.txt"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "readbouncer_amd", "csrc", "rb_kernels.hip")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "--cuda-device-only", "-c", src,
       "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
err = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(src)).stderr
demangle = lambda n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m:
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
print("# hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage readbouncer_amd/csrc/rb_kernels.hip")
print("# %-96s %5s %5s %4s %7s %6s" % ("kernel", "VGPR", "SGPR", "occ", "scratch", "LDS"))
for r in rows:
    name = re.sub(r"\(.*", "", demangle(r["name"])).replace("void rb::", "")
    print("%-98s %5s %5s %4s %7s %6s" % (name[:98], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("Occupancy [waves/SIMD]"),
                                          r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]")))

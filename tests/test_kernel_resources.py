"""Occupancy classes of the count kernels, checked at build time on a CPU (hipcc -Rpass-analysis=kernel-resource-usage, no GPU).
The narrow-filter kernels live on the lookups a CU holds in registers: their builds are compiled against a wave budget
(`amdgpu_waves_per_eu`, rb_kernels.hip) and ONE register more can drop a build from seven to six waves per SIMD -- round 4 lost
16 % on two-word tables that way (73 instead of 72 VGPRs after an innocent change at the END of the kernel) and only a benchmark
noticed.  This pins waves per SIMD and "no scratch" for the builds the planner's window lengths were fitted with."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "readbouncer_amd", "csrc")

# kernel<template arguments> -> least waves per SIMD
EXPECT = {
    "ibf_count_max_kernel<6,2,10,3,1,0>": 3,      # config 3 / 4: 16-byte lanes, 12 KiB of gathers in flight per wave
    "ibf_count_max_kernel<6,2,10,3,1,1>": 3,      # ... its opt-in early-decision twin (last argument; round 6)
    "ibf_count_max_kernel<4,1,10,3,0,0>": 4,      # config 2
    "ibf_count_max_phased_kernel<0,10,1,4>": 8,   # one-word blocks, <= 256 k-mers
    "ibf_count_max_phased_kernel<0,10,3,4>": 6,   # one-word blocks, six tiles (360 bp)
    "ibf_count_max_phased_kernel<0,10,2,4>": 8,   # one-word blocks, rounds of three tiles
    "ibf_count_max_phased_kernel<1,10,1,4>": 7,   # two-word blocks, <= 256 k-mers (merged pairs and triples of small targets)
    "ibf_count_max_phased_kernel<1,10,3,4>": 4,
    "ibf_count_max_phased_kernel<2,10,1,3>": 5,   # three-word blocks, <= 256 k-mers
    "ibf_count_max_phased_kernel<2,10,1,4>": 4,   # four-word blocks (the README shape's packed merged table)
    "ibf_count_max_phased_kernel<2,10,2,3>": 5,
    "ibf_count_max_phased_kernel<2,10,2,4>": 4,
    "ibf_count_max_merged_kernel<3,10,0>": 4,
    # round 6: the builds with a read's block numbers packed in LDS <reads per wave, OR form over a complemented copy, tiles per strand, words per block>
    "ibf_count_max_phased_multi_kernel<1,1,4,2>": 8,   # merged pairs / triples of small targets, <= 256 k-mers (50 registers)
    "ibf_count_max_phased_multi_kernel<1,0,4,2>": 8,   # a two-word filter on its own
    "ibf_count_max_phased_multi_kernel<1,1,6,2>": 7,   # <= 384 k-mers: 360 bp prefixes (69 registers; the register build: 112, four waves)
    "ibf_count_max_phased_multi_kernel<1,0,6,2>": 7,
    "ibf_count_max_phased_multi_kernel<2,1,4,2>": 5,   # two reads per wave (91 registers): measured, not the default
    "ibf_count_max_phased_multi_kernel<1,1,4,4>": 5,   # four-word blocks: the README shape's packed table (92 registers; register build: 116, four waves)
    "ibf_count_max_phased_multi_kernel<1,1,6,4>": 4,   # ... one round of six tiles where the register build takes two rounds of three
    "ibf_count_max_phased_multi_kernel<1,0,4,1>": 8,   # one-word blocks
    "ibf_count_max_phased_multi_kernel<1,0,6,1>": 7,
}


def _demangle_args(mangled):
    """template arguments of _ZN2rb..._kernelI...EEv... as '<a,b,c>' (integers and bools only)"""
    m = re.search(r"_kernelI((?:L[ib]\d+E)+)E", mangled)
    return "<" + ",".join(re.findall(r"L[ib](\d+)E", m.group(1))) + ">" if m else ""


def test_occupancy_classes_and_no_scratch(tmp_path):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("no hipcc on this box: the occupancy classes are pinned where the library is built")
    p = subprocess.run([hipcc, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off", "--offload-arch=gfx950",
                        "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, "rb_kernels.hip"), "-o", str(tmp_path / "k.o")],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    found, cur = {}, None
    for line in p.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            base = re.search(r"\d+(ibf_count_max\w*_kernel)I", name)
            cur = (base.group(1) + _demangle_args(name)) if base else None
            if cur:
                found[cur] = {}
            continue
        if cur:
            for key, pat in (("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("vgpr", r" VGPRs: (\d+)")):
                m = re.search(pat, line)
                if m:
                    found[cur][key] = int(m.group(1))
    assert len(found) > 40, sorted(found)[:5]
    for k, waves in EXPECT.items():
        assert k in found, (k, sorted(x for x in found if x.startswith(k.split("<")[0]))[:20])
        assert found[k]["occ"] >= waves, (k, found[k])
    spilling = {k: v for k, v in found.items() if v.get("scratch", 0) and ("phased" in k or "merged" in k or k.startswith("ibf_count_max_kernel"))}
    assert not spilling, spilling

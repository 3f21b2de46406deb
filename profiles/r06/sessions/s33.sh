#!/bin/bash
# r06 session 33: results never depend on timing -- the soak of round 5 on this round's builds (LDS-offset builds, equal cut, 22-bit one-word numbers):
# thousands of launches of the same batch, window lengths varied on the way, every launch bit-identical
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06l2
mkdir -p $OUT
cd $R
( time timeout 2400 python3 profiles/soak_determinism.py ) > $OUT/soak_determinism.txt 2>&1
echo "exit $?"; grep -v amdgpu.ids $OUT/soak_determinism.txt | cut -c1-220
echo done

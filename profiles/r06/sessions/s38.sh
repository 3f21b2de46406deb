#!/bin/bash
# r06 session 38: the determinism soak with random windows, skew modes and slice cuts on every launch (profiles/soak_random_windows.py)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06l2
mkdir -p $OUT
cd $R
( time timeout 2700 python3 profiles/soak_random_windows.py --launches 6000 ) > $OUT/soak_random_windows.txt 2>&1
echo "exit $?"; grep -v amdgpu.ids $OUT/soak_random_windows.txt | cut -c1-300
echo done

#!/bin/bash
# r06 session 35: the harness as it was (no synchronise between torch's zeroing and the engine's launch), README shape at 360 bp only, 8 000 launches,
# with the report of what a differing launch holds: zeros where the first launch had counts = the zeroing overtook the kernel (the harness's race)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06l2
mkdir -p $OUT
cd $R
( time RB_SOAK_RACY=1 RB_SOAK_ONLY="README shape 360 bp" RB_SOAK_README360=8000 timeout 2400 python3 profiles/soak_determinism.py ) > $OUT/soak_determinism_racy.txt 2>&1
echo "exit $?"; grep -v amdgpu.ids $OUT/soak_determinism_racy.txt | cut -c1-300
echo done

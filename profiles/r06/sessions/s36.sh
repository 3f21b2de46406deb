#!/bin/bash
# r06 session 36: the racy harness once more, 25 000 launches of the README shape at 360 bp in the process state of session 33 (the c4 and the 250 bp
# soaks first), so that a differing launch, if one comes, is described (zeros = the harness's zeroing overtook the kernel)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06l2
mkdir -p $OUT
cd $R
( time RB_SOAK_RACY=1 RB_SOAK_ONLY="README shape" RB_SOAK_README360=25000 timeout 3000 python3 profiles/soak_determinism.py ) > $OUT/soak_determinism_racy_long.txt 2>&1
echo "exit $?"; grep -v amdgpu.ids $OUT/soak_determinism_racy_long.txt | cut -c1-300
echo done

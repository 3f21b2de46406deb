#!/bin/bash
# r05 session 17: fold restricted to one-filter engines and <= 512 reads: the whole GPU suite, then the A/B again
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05k
mkdir -p $OUT
cd $R
( time timeout 1800 python3 -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.txt 2>&1
tail -n 4 $OUT/pytest_gpu.txt | cut -c1-200
timeout 900 python3 profiles/r05/fold_decide_ab.py > $OUT/fold_decide_ab.txt 2>&1
grep -v amdgpu.ids $OUT/fold_decide_ab.txt | tail -40

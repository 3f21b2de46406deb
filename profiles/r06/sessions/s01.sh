#!/bin/bash
# r06 session 1: first run of the multi-read phased build -- parity test, then the window sweep of R = 0 (shipped) / 1 / 2 / 3 on the two-word shapes
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06a
mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "several_reads_per_wave" > $OUT/pytest_multi.txt 2>&1
tail -5 $OUT/pytest_multi.txt
timeout 1200 python3 profiles/multi_reads_sweep.py 2>&1 | tee $OUT/multi_reads_sweep.txt

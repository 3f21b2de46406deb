#!/bin/bash
# r06 session 2: parity of the multi-read build (test fixed), then XCD skew modes (0 none, 1 slice skew, 2 time skew, 3 both) x R = 0 / 2, windows swept
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06b
mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "several_reads_per_wave" > $OUT/pytest_multi.txt 2>&1
tail -5 $OUT/pytest_multi.txt
timeout 1500 python3 profiles/multi_reads_sweep.py --workloads deplete_target --rpw 0,2 --skew 0,1,2,3 --factors 0.5,0.6,0.7,0.8,0.9,1.0,1.1,1.2,1.35,1.5 2>&1 | tee $OUT/skew_sweep.txt

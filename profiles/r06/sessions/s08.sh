#!/bin/bash
# r06 session 8: streaming form with tickets of eight reads
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06h
mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "several_reads" > $OUT/pytest_new.txt 2>&1
tail -3 $OUT/pytest_new.txt
timeout 1200 python3 profiles/multi_reads_sweep.py --workloads deplete_target --rpw 1,33 --skew 0,2 --factors 0.8,0.9,0.95,1.0,1.05,1.1,1.2,1.3 2>&1 | grep -v amdgpu.ids | tee $OUT/stream_sweep.txt

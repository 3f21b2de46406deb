#!/bin/bash
# r06 session 3: the multi-read build with 32-bit k-mer values from staged triples and (merged copies) the complemented twin + OR form:
# parity, then R = 0 / 1 / 2 / 3 and the AND form of R = 2 (18), XCD time skew off / on, windows swept
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06c
mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "several_reads_per_wave" > $OUT/pytest_multi.txt 2>&1
tail -5 $OUT/pytest_multi.txt
timeout 1500 python3 profiles/multi_reads_sweep.py --workloads deplete_target,targets3 --rpw 0,1,2,18,3 --skew 0,2 --factors 0.5,0.6,0.7,0.8,0.9,1.0,1.1,1.2,1.35,1.5 2>&1 | tee $OUT/multi_sweep.txt

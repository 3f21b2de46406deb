#!/bin/bash
# r06 session 6: the streaming form of the one-read build (rb_engine_set_reads_per_wave(1 + 32)): parity, then against the per-read form
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06f
mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "device_thresholds or placed_by_trial or several_reads" > $OUT/pytest_new.txt 2>&1
tail -5 $OUT/pytest_new.txt
timeout 1200 python3 profiles/multi_reads_sweep.py --workloads deplete_target,targets3 --rpw 1,33 --skew 0,2 --factors 0.8,0.9,0.95,1.0,1.05,1.1,1.2,1.3 2>&1 | grep -v amdgpu.ids | tee $OUT/stream_sweep.txt

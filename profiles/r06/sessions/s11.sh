#!/bin/bash
# r06 session 11: prefetching waves with slices of 1 MiB (the current and the next slice are half of an L2 together) and of 512 KiB
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06k
mkdir -p $OUT
cd $R
export RB_TUNING_ENV=1
for sl in 20 19; do for pre in 0 1 2; do
  echo "== RB_STREAM_PREFETCH=$pre slices of 2^$sl"
  RB_STREAM_PREFETCH=$pre RB_MULTI_UB=1 timeout 600 python3 profiles/multi_reads_sweep.py --workloads deplete_target --rpw 33 --skew 2 --slice-log2 $sl --factors 0.5,0.6,0.7,0.8,0.9,1.0,1.15,1.3,1.5,1.8 2>&1 | grep -v amdgpu.ids
done; done 2>&1 | tee $OUT/stream_prefetch_small_slices.txt

#!/bin/bash
# r06 session 12: (a) prefetching waves, 16 / 32 / 64 per XCD, roles by the hardware's XCD number, 32 loads in flight; 2 MiB slices
# (b) the per-read R = 1 build: two slots per batch of gathers; equal-length slices (4 / 5 / 6 for the 18.9 MiB table)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06l
mkdir -p $OUT
cd $R
export RB_TUNING_ENV=1
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "several_reads" > $OUT/pytest_new.txt 2>&1
tail -3 $OUT/pytest_new.txt
for pre in 16 32 64; do
  echo "== RB_STREAM_PREFETCH=$pre slices of 2 MiB"
  RB_STREAM_PREFETCH=$pre timeout 600 python3 profiles/multi_reads_sweep.py --workloads deplete_target --rpw 33 --skew 2 --slice-log2 21 --factors 0.6,0.7,0.8,0.9,1.0,1.1,1.25,1.4,1.6 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $OUT/stream_prefetch_many.txt
echo "== per-read R = 1, two slots per batch"
RB_MULTI_UB=2 timeout 600 python3 profiles/multi_reads_sweep.py --workloads deplete_target --rpw 1 --skew 2 --factors 0.7,0.8,0.9,0.95,1.0,1.05,1.1,1.2 2>&1 | grep -v amdgpu.ids | tee $OUT/ub2.txt
for n in 4 5 6; do
  echo "== RB_PHASE_N_SLICES=$n (equal-length slices)"
  RB_PHASE_N_SLICES=$n timeout 600 python3 profiles/multi_reads_sweep.py --workloads deplete_target --rpw 1 --skew 2 --factors 0.7,0.8,0.9,0.95,1.0,1.05,1.1,1.2,1.3 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $OUT/equal_slices.txt

#!/bin/bash
# r06 session 10: the streaming form with guided tickets; then with waves that only prefetch the next slice (RB_STREAM_PREFETCH per XCD),
# 2 MiB and 1 MiB slices, one or two slots per batch of gathers
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06j
mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "several_reads" > $OUT/pytest_new.txt 2>&1
tail -3 $OUT/pytest_new.txt
export RB_TUNING_ENV=1
echo "== streaming, guided tickets, no prefetch"
timeout 600 python3 profiles/multi_reads_sweep.py --workloads deplete_target --rpw 1,33 --skew 0,2 --factors 0.9,0.95,1.0,1.05,1.1,1.15,1.2 2>&1 | grep -v amdgpu.ids | tee $OUT/stream_guided.txt
for pre in 0 2 4 8; do for ub in 1 2; do
  echo "== RB_STREAM_PREFETCH=$pre RB_MULTI_UB=$ub slices of 2 MiB"
  RB_STREAM_PREFETCH=$pre RB_MULTI_UB=$ub RB_PHASE_SLICE_LOG2=21 timeout 600 python3 profiles/multi_reads_sweep.py --workloads deplete_target --rpw 33 --skew 0,2 --slice-log2 21 --factors 0.6,0.7,0.8,0.9,1.0,1.1,1.25,1.4,1.6 2>&1 | grep -v amdgpu.ids
done; done 2>&1 | tee $OUT/stream_prefetch.txt

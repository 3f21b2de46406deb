#!/bin/bash
# r06 session 5: new tests (device thresholds against the reference-compiled fixture, placement accounting), then R = 1 OR-form experiments:
# cache-policy bits of the gathers, the fraction of a window by which the XCDs' clocks are apart, a fine window sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06e
mkdir -p $OUT
cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "device_thresholds or placed_by_trial or several_reads" > $OUT/pytest_new.txt 2>&1
tail -5 $OUT/pytest_new.txt
export RB_TUNING_ENV=1
for aux in 0 1 2 3 17; do
  echo "== RB_MULTI_AUX=$aux"
  RB_MULTI_AUX=$aux timeout 600 python3 profiles/multi_reads_sweep.py --workloads deplete_target --rpw 1 --skew 2 --factors 0.9,0.95,1.0,1.05,1.1 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $OUT/aux_sweep.txt
for div in 3 4 5 6 10 16; do
  echo "== RB_PHASE_TSKEW_DIV=$div"
  RB_PHASE_TSKEW_DIV=$div timeout 600 python3 profiles/multi_reads_sweep.py --workloads deplete_target --rpw 1,0 --skew 2 --factors 0.9,0.95,1.0,1.05,1.1 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $OUT/tskew_div_sweep.txt

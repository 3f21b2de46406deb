#!/bin/bash
# r06 session 21: the grid for a rule of equal-length slices shorter than an L2 (two-word LDS-offset builds): sizes x read lengths x slice
# counts x cycles (profiles/equal_slices_fit.py); then the merged OR-form shapes of the bench at 18.9 MiB with finer windows around the optima
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06t
mkdir -p $OUT
cd $R
export RB_TUNING_ENV=1
timeout 1500 python3 profiles/equal_slices_fit.py 2>&1 | grep -v amdgpu.ids | tee $OUT/equal_slices_fit_two_word.txt | cut -c1-150
F=0.7,0.76,0.82,0.88,0.94,1.0,1.06,1.12,1.18,1.25
for n in 7 8; do
  echo "== RB_PHASE_N_SLICES=$n"
  RB_PHASE_N_SLICES=$n timeout 600 python3 profiles/multi_reads_sweep.py --workloads deplete_target,targets3,deplete_target360,targets3_360 --rpw 1 --skew 2 --factors $F 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $OUT/equal_slices_merged_fine.txt | cut -c1-330
echo done

#!/bin/bash
# r06 session 20: equal-length slices that are NOT fewer than the 4 MiB cut (session 12's RB_PHASE_N_SLICES=5 / 6 fell back to the 4 MiB cut:
# the planner took an explicit count only when it was smaller).  Two-word merged table of 18.9 MiB: 5 x 3.78 MiB ... 10 x 1.9 MiB, at 250
# and 360 bp; the README shape's four-word table (39.5 MB; the rule: 8 equal slices): 9 ... 12.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06s
mkdir -p $OUT
cd $R
export RB_TUNING_ENV=1
F=0.6,0.7,0.8,0.85,0.9,0.95,1.0,1.05,1.1,1.2,1.35
for n in 0 5 6 7 8 10; do
  echo "== RB_PHASE_N_SLICES=$n"
  RB_PHASE_N_SLICES=$n timeout 600 python3 profiles/multi_reads_sweep.py --workloads deplete_target,deplete_target360 --rpw 1 --skew 2 --factors $F 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $OUT/equal_slices_two_word.txt | cut -c1-400
for n in 0 9 10 12; do
  echo "== RB_PHASE_N_SLICES=$n"
  RB_PHASE_N_SLICES=$n timeout 600 python3 profiles/multi_reads_sweep.py --workloads readme,readme360 --rpw 1 --skew 2 --factors $F 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $OUT/equal_slices_four_word.txt | cut -c1-400
for n in 0 5 6 7; do
  echo "== RB_PHASE_N_SLICES=$n (one-word table of 19.8 MB, the register builds)"
  RB_PHASE_N_SLICES=$n timeout 600 python3 profiles/multi_reads_sweep.py --workloads c1,c1_360 --rpw 0 --skew 2 --factors $F 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $OUT/equal_slices_one_word.txt | cut -c1-400
echo done

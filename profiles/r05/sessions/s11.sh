#!/bin/bash
# r05 session 11: does K1 follow the probe between allocations of one table in one process?
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s11
mkdir -p $OUT
cd $R
timeout 900 python3 profiles/placement_trial.py c3np2 5 > $OUT/placement_trial_c3np2.txt 2>&1
grep -v amdgpu.ids $OUT/placement_trial_c3np2.txt
timeout 900 python3 profiles/placement_trial.py grch38_f100k 5 > $OUT/placement_trial_grch38_f100k.txt 2>&1
grep -v amdgpu.ids $OUT/placement_trial_grch38_f100k.txt
timeout 900 python3 profiles/placement_trial.py c3 4 > $OUT/placement_trial_c3.txt 2>&1
grep -v amdgpu.ids $OUT/placement_trial_c3.txt

#!/bin/bash
# r05 session 25: what the placement trial costs at load time, by number of candidates
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05s
mkdir -p $OUT
cd $R
timeout 600 python3 profiles/r05/placement_trial_cost.py > $OUT/placement_trial_cost.txt 2>&1; grep -v amdgpu.ids $OUT/placement_trial_cost.txt

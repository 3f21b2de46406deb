#!/bin/bash
# r06 session 31: the equal-cut grid for four-word blocks of 13-30 MiB (the README shape's 39.5 MB table lost with more, shorter slices -- do smaller
# four-word tables, which the rule cuts at 4 MiB, behave like it or like the one- and two-word tables?)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06m
mkdir -p $OUT
cd $R
export RB_TUNING_ENV=1
timeout 1500 python3 profiles/equal_slices_fit.py --words 4 --sizes 13,18,24,30 --lens 250,360 --targets 2.4,3.0,3.6,4.4 --cycles 2800,3100,3400,3700,4000,4300,4600,5000,5400,5900,6400,7000 2>&1 | grep -v amdgpu.ids | tee $OUT/equal_slices_fit_four_word.txt | cut -c1-150
echo done

#!/bin/bash
# r06 session 23: the same grid for ONE-word tables of 13-48 MiB (register builds; the config-1 geometry of 19.8 MB ran 8.42 -> 7.59 ms per 1 M
# reads of 250 bp in six equal slices in session 20)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06v
mkdir -p $OUT
cd $R
export RB_TUNING_ENV=1
timeout 1500 python3 profiles/equal_slices_fit.py --words 1 --sizes 13,16,19.8,24,28,32,40,48 --targets 2.0,2.4,2.75,3.2,3.6 --cycles 2800,3100,3400,3700,4000,4300,4600,5000,5400,5900,6400,7000,7700 2>&1 | grep -v amdgpu.ids | tee $OUT/equal_slices_fit_one_word.txt | cut -c1-150
echo done

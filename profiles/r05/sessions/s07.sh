#!/bin/bash
# r05 session 7: one-word equal-length slices, second sweep: larger slices and longer cycles for 64-127 MiB
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s07
mkdir -p $OUT
cd $R
( time timeout 1200 python3 profiles/one_word_equal_slices.py --points 64:250,80:250,96:250,112:250,127:250,64:360,96:360,127:360 --targets 6.4,8,10,12.8 --cycles 9000,10500,12000,13500,15000,17000 ) > $OUT/one_word_equal_slices_large.txt 2>&1
grep -v amdgpu.ids $OUT/one_word_equal_slices_large.txt | cut -c1-900

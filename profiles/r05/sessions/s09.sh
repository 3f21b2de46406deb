#!/bin/bash
# r05 session 9: do slices longer than an L2 pay for large TWO-word tables as well?  (sweep only)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s09
mkdir -p $OUT
cd $R
( time timeout 1200 python3 profiles/one_word_equal_slices.py --bins 128 --points 32:250,48:250,64:250,80:250,96:250,48:360,64:360,80:360 --targets 4.6,5.5,6.5,8.0 --cycles 5000,6000,7000,8000,9500,11000 ) > $OUT/two_word_equal_slices.txt 2>&1
grep -v amdgpu.ids $OUT/two_word_equal_slices.txt | cut -c1-1000

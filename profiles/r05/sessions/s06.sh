#!/bin/bash
# r05 session 6: equal-length slices for one-word tables of 40-127 MiB (sweep behind a rule, DESIGN 8.4 of round 4)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s06
mkdir -p $OUT
cd $R
( time timeout 1200 python3 profiles/one_word_equal_slices.py ) > $OUT/one_word_equal_slices.txt 2>&1
grep -v amdgpu.ids $OUT/one_word_equal_slices.txt | cut -c1-700

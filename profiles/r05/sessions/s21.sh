#!/bin/bash
# r05 session 21: the tail of the completion-word path (p99 of some batch sizes and of the replay rose with it): stream wait, word, word + query / synchronise at the next call
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05o
mkdir -p $OUT
cd $R
timeout 1200 python3 profiles/r05/completion_word_tail.py > $OUT/completion_word_tail.txt 2>&1
grep -v amdgpu.ids $OUT/completion_word_tail.txt | cut -c1-330

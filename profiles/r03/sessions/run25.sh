#!/bin/bash
# round 3, GPU session 25: slice size x window length of the phased gathers, single filters of 6-32 MiB
set -u
O=gpurun_out/r03
mkdir -p $O
timeout 900 python profiles/r03/slice_size_sweep.py 1 > $O/slice_size_w1.txt 2>&1
cat $O/slice_size_w1.txt
timeout 900 python profiles/r03/slice_size_sweep.py 2 > $O/slice_size_w2.txt 2>&1
cat $O/slice_size_w2.txt

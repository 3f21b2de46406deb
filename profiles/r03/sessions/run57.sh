#!/bin/bash
# round 3, GPU session 57: small tables with reads that do not fill the shape (150 / 200 / 300 bp), builds with more waves per SIMD
set -u
O=gpurun_out/r03
mkdir -p $O
T="130,200,250,325,400,500,700"
timeout 900 python profiles/r03/slice_size_sweep.py 1 150,200,300 2,3,4,5,6,7 19,20,21 $T > $O/occ_small_short_w1.txt 2>&1
timeout 900 python profiles/r03/slice_size_sweep.py 2 150,200,300 2,3,4,5,6,7 19,20,21 $T > $O/occ_small_short_w2.txt 2>&1

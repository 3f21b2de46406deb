#!/bin/bash
# round 3, GPU session 55: window sweep for the builds with more waves per SIMD (one-word 250 / 360 bp, two-word 250 bp)
set -u
O=gpurun_out/r03
mkdir -p $O
T="250,325,400,500,600,700,850,1000,1200,1500,1800,2000"
S="7,8,9,10.5,12,14,16,18,20,24,28,32,40,48,64,96,127"
timeout 900 python profiles/r03/slice_size_sweep.py 1 250,360 $S 21,22 $T > $O/occ_fine_w1.txt 2>&1
timeout 900 python profiles/r03/slice_size_sweep.py 2 250 $S 21,22 $T > $O/occ_fine_w2.txt 2>&1
timeout 900 python profiles/r03/slice_size_sweep.py 1 250,360 1.5,2,3,4,5,6 19,20 100,130,160,200,250,325,400,500,600 > $O/occ_fine_small_w1.txt 2>&1
timeout 900 python profiles/r03/slice_size_sweep.py 2 250 1.5,2,3,4,5,6 19,20 100,130,160,200,250,325,400,500,600 > $O/occ_fine_small_w2.txt 2>&1

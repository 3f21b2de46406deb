#!/bin/bash
# round 3, GPU session 26: fine sweep of slice size x window length (and tables beyond 32 MiB) for the phased gathers
set -u
O=gpurun_out/r03
mkdir -p $O
T="250,325,400,500,600,700,850,1000,1200,1500,1800,2000"
S="7,8,9,10.5,12,14,16,18,20,24,28,32,40,48,64,96"
timeout 900 python profiles/r03/slice_size_sweep.py 1 250,360 $S 21,22 $T > $O/slice_fine_w1.txt 2>&1
timeout 900 python profiles/r03/slice_size_sweep.py 2 250,360 $S 21,22 $T > $O/slice_fine_w2.txt 2>&1
timeout 600 python profiles/r03/slice_size_sweep.py 1 500,1000 8,12,16,24,32,48 21,22 300,450,600,800,1000,1400 > $O/slice_fine_w1_long.txt 2>&1
timeout 600 python profiles/r03/slice_size_sweep.py 4 250,360 8,12,16,24,32,48,64 21,22 300,450,600,800,1000,1400 > $O/slice_fine_w4.txt 2>&1
tail -3 $O/slice_fine_w1.txt $O/slice_fine_w2.txt $O/slice_fine_w1_long.txt $O/slice_fine_w4.txt

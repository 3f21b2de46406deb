#!/bin/bash
# round 3, GPU session 38: the window rule on read lengths it was not fitted at (150 / 200 / 300 bp; fitted at 250 / 360 / 500 / 1000)
set -u
O=gpurun_out/r03
mkdir -p $O
T="250,325,400,500,600,700,850,1000,1200,1500,1800"
timeout 900 python profiles/r03/slice_size_sweep.py 1 150,200,300,430 8,10.5,20,32,64 21,22 $T > $O/other_lengths_w1.txt 2>&1
timeout 900 python profiles/r03/slice_size_sweep.py 2 150,200,300,430 10.5,19,32,64 21,22 $T > $O/other_lengths_w2.txt 2>&1
tail -2 $O/other_lengths_w1.txt

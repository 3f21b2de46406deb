#!/bin/bash
# round 3, GPU session 41: small tables (1-6 MiB), slices of 512 KiB / 1 MiB, short windows
set -u
O=gpurun_out/r03
mkdir -p $O
T="100,130,160,200,250,325,400,500"
timeout 900 python profiles/r03/slice_size_sweep.py 1 250,360 1,1.5,2,3,4,5,6 19,20 $T > $O/small2_w1.txt 2>&1
timeout 900 python profiles/r03/slice_size_sweep.py 2 250,360 1,1.5,2,3,4,5,6 19,20 $T > $O/small2_w2.txt 2>&1


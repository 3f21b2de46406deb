#!/bin/bash
# round 3, GPU session 40: below the phased range (tables of 3-6 MiB): phased with 1 / 2 MiB slices against the plain kernel
set -u
O=gpurun_out/r03
mkdir -p $O
T="250,325,400,500,600,700,850,1000,1200,1500,1800"
timeout 900 python profiles/r03/slice_size_sweep.py 1 250,360 2,3,4,5,6,7 20,21 $T > $O/small_w1.txt 2>&1
timeout 900 python profiles/r03/slice_size_sweep.py 2 250,360 2,3,4,5,6,7 20,21 $T > $O/small_w2.txt 2>&1

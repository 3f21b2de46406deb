#!/bin/bash
# round 3, GPU session 43: two-word blocks outside the phased range on the both-strands round without a clock
set -u
O=gpurun_out/r03
mkdir -p $O
timeout 900 python profiles/r03/slice_size_sweep.py 2 150,250,360,500 0.5,1,3,6,110,200 20 250 > $O/noclock_w2.txt 2>&1
grep -h "rule\|plain" $O/noclock_w2.txt

#!/bin/bash
# round 3, GPU session 42: the rule on small tables (phased from 1.5 MiB on: slices of 512 KiB / 1 MiB), suite
set -u
O=gpurun_out/r03
mkdir -p $O
timeout 900 python profiles/r03/slice_size_sweep.py 1 150,250,360,500 1,1.5,2,3,4,5,6,7 20 250 > $O/small3_w1.txt 2>&1
timeout 900 python profiles/r03/slice_size_sweep.py 2 150,250,360,500 1,1.5,2,3,4,5,6,7 20 250 > $O/small3_w2.txt 2>&1
grep -h "rule\|plain" $O/small3_w1.txt $O/small3_w2.txt | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9}' | paste - - 


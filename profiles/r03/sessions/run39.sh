#!/bin/bash
# round 3, GPU session 39: the window rule scaled by how full the kernel shape is (k-mers of the batch / k-mers the shape was fitted with)
set -u
O=gpurun_out/r03
mkdir -p $O
timeout 900 python profiles/r03/slice_size_sweep.py 1 150,200,250,300,360,430 8,10.5,20,32,64 22 600 > $O/fill_w1.txt 2>&1
timeout 900 python profiles/r03/slice_size_sweep.py 2 150,200,250,300,360,430 10.5,19,32,64 22 600 > $O/fill_w2.txt 2>&1
grep -h "rule" $O/fill_w1.txt $O/fill_w2.txt

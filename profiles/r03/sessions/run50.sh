#!/bin/bash
# round 3, GPU session 50: four- and three-word tables around 32 MiB (eight slices of 4 MiB): window length
set -u
O=gpurun_out/r03
mkdir -p $O
T="200,250,325,400,450,500,600,700,800"
timeout 900 python profiles/r03/slice_size_sweep.py 4 250,360 28,32,36 22 $T > $O/wide5_w4.txt 2>&1
timeout 900 python profiles/r03/slice_size_sweep.py 3 250,360 18,24,30 22 $T > $O/wide5_w3.txt 2>&1
cat $O/wide5_w4.txt $O/wide5_w3.txt
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "merged or merge" 2>&1 | tail -3

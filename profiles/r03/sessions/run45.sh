#!/bin/bash
# round 3, GPU session 45: four-word blocks, 250 bp: one round of four tiles per strand (<= 256 k-mers)
set -u
O=gpurun_out/r03
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "raw_max or fuzz or long_reads or packed" 2>&1 | tail -3
T="150,200,250,325,400,500,600,800,1000,1400"
timeout 900 python profiles/r03/slice_size_sweep.py 4 250 2,4,8,16,24,40,64 21,22 $T > $O/wide2_w4.txt 2>&1
timeout 900 python profiles/r03/slice_size_sweep.py 3 250 4,16,40 21,22 $T > $O/wide2_w3.txt 2>&1

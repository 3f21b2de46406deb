#!/bin/bash
# round 3, GPU session 46: four-word blocks, 360 / 500 bp: rounds of four tiles per strand (three waves per SIMD) instead of two
set -u
O=gpurun_out/r03
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "raw_max or fuzz or long_reads or packed" 2>&1 | tail -3
T="150,200,250,325,400,500,600,800,1000,1400"
timeout 900 python profiles/r03/slice_size_sweep.py 4 360,500 4,8,16,24,40 21,22 $T > $O/wide3_w4.txt 2>&1

#!/bin/bash
# round 3, GPU session 44: three- and four-word blocks through a both-strands build of the phased kernel (one lane per block, two 16-byte
# gathers per lookup, rounds of two tiles per strand): parity, then slice size x window length against the plain kernel
set -u
O=gpurun_out/r03
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "raw_max or fuzz or long_reads or packed" 2>&1 | tail -3
T="150,200,250,325,400,500,600,800,1000,1400"
timeout 900 python profiles/r03/slice_size_sweep.py 4 250,360 2,4,8,16,24,40,64 20,21,22 $T > $O/wide_w4.txt 2>&1
timeout 900 python profiles/r03/slice_size_sweep.py 3 250,360 4,16,40 20,21,22 $T > $O/wide_w3.txt 2>&1
tail -3 $O/wide_w4.txt

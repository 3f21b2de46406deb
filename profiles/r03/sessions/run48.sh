#!/bin/bash
# round 3, GPU session 48: three- / four-word blocks, 360 and 500 bp: rounds of THREE tiles per strand (113 registers, four waves)
set -u
O=gpurun_out/r03
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "raw_max or fuzz or long_reads or packed" 2>&1 | tail -3
T="150,200,250,325,400,500,600,800,1000,1400"
timeout 900 python profiles/r03/slice_size_sweep.py 4 300,360,500 4,8,12,16,24,40,47 21,22 $T > $O/wide4_w4.txt 2>&1

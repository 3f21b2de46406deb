#!/bin/bash
# round 3, GPU session 49: merged tables of two to four words through the one-lane-per-block builds of the phased kernel
set -u
O=gpurun_out/r03
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "merged or merge" 2>&1 | tail -5
timeout 800 python profiles/r03/merged_tables.py sweep > $O/merged_sweep2.txt 2>&1
cat $O/merged_sweep2.txt

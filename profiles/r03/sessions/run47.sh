#!/bin/bash
# round 3, GPU session 47: the rule for three- and four-word blocks against the plain kernel; GPU suite
set -u
O=gpurun_out/r03
mkdir -p $O
timeout 900 python profiles/r03/slice_size_sweep.py 4 150,250,360,500 2,4,6,8,12,16,24,40,47,64 22 500 > $O/wide_rule_w4.txt 2>&1
timeout 900 python profiles/r03/slice_size_sweep.py 3 150,250,360,500 2,4,8,16,24,40,64 22 500 > $O/wide_rule_w3.txt 2>&1
python -m pytest tests -q -m gpu -x 2>&1 | tail -3

#!/bin/bash
# round 3, GPU session 62: final tree with the three-tile rounds for one-word blocks: suite, rule check at 430 / 500 bp
set -u
O=gpurun_out/r03
mkdir -p $O
python -m pytest tests -q -m gpu -x 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python profiles/r03/slice_size_sweep.py 1 400,430,500 3,5.5,6,8,10.5,16,20,32,48,64 22 500 > $O/t3_rule.txt 2>&1
grep -h "rule" $O/t3_rule.txt

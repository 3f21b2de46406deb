#!/bin/bash
# r06 session 14: the tree with the LDS-offset builds and the XCD time skew as defaults: the whole -m gpu suite, the planner guard with two-word
# points inside the new builds' range, usage = "build" at genome scale
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06n
mkdir -p $OUT
cd $R
timeout 2400 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
timeout 1500 python3 profiles/phase_rule_check.py --reads 1000000 --points 2:250:19,2:360:19,2:200:13,2:300:13,2:250:8,2:360:8,2:250:28,2:360:28,2:200:24,2:300:30,2:150:19 --factors 0.6,0.7,0.8,0.9,1.0,1.1,1.2,1.35,1.5 > $OUT/phase_rule_check_two_word.txt 2>&1
echo "guard exit $?" >> $OUT/phase_rule_check_two_word.txt
cut -c1-300 $OUT/phase_rule_check_two_word.txt | tail -50
timeout 1500 python3 profiles/cli_build.py > $OUT/cli_build.txt 2>&1
cut -c1-400 $OUT/cli_build.txt

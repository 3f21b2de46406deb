#!/bin/bash
# Round 4, session 4: parity suite on the new host-batch path, planner guard with the rebuilt rules, CLI sweep
TAG=${1:-r04s4}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
T="timeout 1500"
( time $T python3 -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
( time $T python3 profiles/phase_rule_check.py ) > $OUT/phase_rule_check.txt 2>&1
echo "phase_rule_check rc=$?" >> $OUT/phase_rule_check.txt
grep -E "rule vs best|outside|rc=" $OUT/phase_rule_check.txt | cut -c1-260
( time $T python3 profiles/cli_readme250.py ) > $OUT/cli_throughput.txt 2>&1
cut -c1-400 $OUT/cli_throughput.txt
( time $T python3 profiles/engines_on_one_gpu.py --shapes readme,c4 --forms host ) > $OUT/engines_host.txt 2>&1
tail -8 $OUT/engines_host.txt

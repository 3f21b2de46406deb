#!/bin/bash
# Round 4, session 8: final tree -- parity suite, the driver's command, narrow-shape stats + PMC, CLI throughput (writer thread per file)
TAG=${1:-r04s8}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
T="timeout 1800"
( time $T python3 -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.txt 2>&1
tail -n 5 $OUT/pytest_gpu.txt | cut -c1-300
( time $T python3 profiles/cli_readme250.py ) > $OUT/cli_throughput.txt 2>&1
cut -c1-420 $OUT/cli_throughput.txt
bash profiles/r04/sessions/collect_r04_s7.sh $TAG

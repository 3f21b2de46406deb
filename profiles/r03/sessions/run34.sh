#!/bin/bash
# round 3, GPU session 34: the README benchmark end to end through the CLI (round 2: 3.1-3.3 M reads/s at max_chunks 1) and the
# CLI throughput on the c2 filter
set -u
O=gpurun_out/r03
mkdir -p $O
timeout 900 python profiles/cli_readme.py 2000000 > $O/cli_readme.txt 2>&1
cat $O/cli_readme.txt
RB_MERGE=0 timeout 900 python profiles/cli_readme.py 2000000 > $O/cli_readme_apart.txt 2>&1
cat $O/cli_readme_apart.txt

#!/bin/bash
# round 3, GPU session 24: from which table size on does a merged pair / triple pay?
set -u
O=gpurun_out/r03
mkdir -p $O
timeout 800 python profiles/r03/merged_tables.py sweep > $O/merged_sweep.txt 2>&1
cat $O/merged_sweep.txt

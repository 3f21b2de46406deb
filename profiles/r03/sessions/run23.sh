#!/bin/bash
# round 3, GPU session 23: merged-table bookkeeping (merge_info, the size cap) and the merged table on pairs of LARGE filters
set -u
O=gpurun_out/r03
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "merged" 2>&1 | tail -5 > $O/merged_tests.txt
cat $O/merged_tests.txt
timeout 600 python profiles/r03/merged_tables.py > $O/merged_big_tables.txt 2>&1
cat $O/merged_big_tables.txt

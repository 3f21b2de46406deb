#!/bin/bash
# r05 session 10: the two bench-structure tests that still asked for the old key name; the -m gpuperf set again; allocation placement of the
# 4.7 GB table within one process; deep parity of the final tree
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s10
mkdir -p $OUT
cd $R
( timeout 900 python3 -m pytest tests/test_bench_ranks.py -m gpu -x -q ) > $OUT/pytest_bench_ranks.txt 2>&1
tail -n 3 $OUT/pytest_bench_ranks.txt | cut -c1-300
( time timeout 1500 python3 -m pytest tests -m gpuperf -q ) > $OUT/pytest_gpuperf.txt 2>&1
tail -n 4 $OUT/pytest_gpuperf.txt | cut -c1-300
timeout 600 python3 profiles/placement_probe.py c3np2 8 > $OUT/placement_probe_c3np2.txt 2>&1
grep -v amdgpu.ids $OUT/placement_probe_c3np2.txt
timeout 600 python3 profiles/placement_probe.py grch38_f100k 6 > $OUT/placement_probe_grch38_f100k.txt 2>&1
grep -v amdgpu.ids $OUT/placement_probe_grch38_f100k.txt
( time bash profiles/r05/deep_parity_r05.sh ) > $OUT/deep_parity.txt 2>&1
cp gpurun_out/r05deep/*.err $OUT/ 2>/dev/null
cat $OUT/deep_parity.txt | cut -c1-330

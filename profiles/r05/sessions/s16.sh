#!/bin/bash
# r05 session 16: the latency kernel makes the decisions itself (FoldJob): parity test first, then the A/B by batch size
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05j
mkdir -p $OUT
cd $R
( time timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "latency_kernel_makes or raw_max_matches or fused_wide or host_api_micro or back_to_back" ) > $OUT/pytest_fold.txt 2>&1
tail -n 6 $OUT/pytest_fold.txt | cut -c1-300
timeout 900 python3 profiles/r05/fold_decide_ab.py > $OUT/fold_decide_ab.txt 2>&1
cat $OUT/fold_decide_ab.txt | tail -40

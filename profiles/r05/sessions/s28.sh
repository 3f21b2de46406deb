#!/bin/bash
# r05 session 28: pool replicas started from threads (testing build's switch on one GPU): the pool tests, then the whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05v
mkdir -p $OUT
cd $R
( time timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "pool" ) > $OUT/pytest_pool.txt 2>&1
tail -n 5 $OUT/pytest_pool.txt | cut -c1-300
( time timeout 1800 python3 -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.txt 2>&1
tail -n 4 $OUT/pytest_gpu.txt | cut -c1-200

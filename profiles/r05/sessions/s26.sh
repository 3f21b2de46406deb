#!/bin/bash
# r05 session 26: the tree with the threaded loader and replicas allocated side by side: the whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05t
mkdir -p $OUT
cd $R
( time timeout 1800 python3 -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.txt 2>&1
tail -n 4 $OUT/pytest_gpu.txt | cut -c1-200

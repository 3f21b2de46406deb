#!/bin/bash
# r05 session 15: the C99 example over the C ABI (tests/test_examples.py) on the GPU, then the whole GPU suite with it in
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05i
mkdir -p $OUT
cd $R
gcc -std=c99 -O1 -Iinclude examples/adaptive_sampling_c_abi.c -Lreadbouncer_amd -lreadbouncer_amd -Wl,-rpath,$R/readbouncer_amd -o /tmp/asc
( time /tmp/asc ) > $OUT/c_example.txt 2>&1; echo "example rc=$?"; cat $OUT/c_example.txt
( time timeout 1800 python3 -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.txt 2>&1
tail -n 4 $OUT/pytest_gpu.txt | cut -c1-200

#!/bin/bash
# r05 session 24: the loader with reader threads and the deferred wait of the placement trial: parity at 150-200 MB, then the 8 GiB timings again
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05r
mkdir -p $OUT
cd $R
( time timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stream_into_hbm or file or roundtrip or round_trip" ) > $OUT/pytest_load.txt 2>&1
tail -n 5 $OUT/pytest_load.txt | cut -c1-300


ls /dev/shm | head -3

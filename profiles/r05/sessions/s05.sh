#!/bin/bash
# r05 session 5: the doorbell kernel -- its tests (own timeout: a resident kernel that does not leave must not hold the box), then the latency figures
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s05
mkdir -p $OUT
cd $R
( time timeout 600 python3 -m pytest tests/test_gpu_doorbell.py -m gpu -x -q -s ) > $OUT/pytest_doorbell.txt 2>&1
tail -n 30 $OUT/pytest_doorbell.txt | cut -c1-300
( time timeout 600 python3 profiles/doorbell_latency.py ) > $OUT/doorbell_latency.txt 2>&1
grep -v amdgpu.ids $OUT/doorbell_latency.txt | cut -c1-300

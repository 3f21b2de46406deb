#!/bin/bash
# round 3, GPU session 32: full GPU suite and deep parity on the final tree (slice / window rule of session 28, merge cost model)
set -u
O=gpurun_out/r03
mkdir -p $O
python -m pytest tests -q -m gpu 2>&1 | tail -4 > $O/pytest_gpu_full_j.txt
cat $O/pytest_gpu_full_j.txt
bash profiles/deep_parity_r03.sh > $O/deep_parity_j.txt 2>&1
cat $O/deep_parity_j.txt

#!/bin/bash
# r05 session 20: the completion word of the micro-batch path: parity, then the A/B by batch size, then the whole GPU suite
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05n
mkdir -p $OUT
cd $R
( time timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "completion_word or latency_kernel_makes or host_api_micro or back_to_back" ) > $OUT/pytest_word.txt 2>&1
tail -n 6 $OUT/pytest_word.txt | cut -c1-300
timeout 900 python3 profiles/r05/completion_word_ab.py > $OUT/completion_word_ab.txt 2>&1
grep -v amdgpu.ids $OUT/completion_word_ab.txt | tail -40
( time timeout 1800 python3 -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.txt 2>&1
tail -n 4 $OUT/pytest_gpu.txt | cut -c1-200

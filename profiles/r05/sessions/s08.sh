#!/bin/bash
# r05 session 8: the one-word equal-length-slice rule in the planner: its GPU test, the rule against forced neighbours at sizes BETWEEN the
# fitted ones, the planner guard at its default points
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s08
mkdir -p $OUT
cd $R
( timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "equal_length or phased or measurement_aids" ) > $OUT/pytest_equal_slices.txt 2>&1
tail -n 3 $OUT/pytest_equal_slices.txt | cut -c1-300
( time timeout 1200 python3 profiles/one_word_equal_slices.py --points 52:250,60:250,72:250,88:250,104:250,120:250,60:360,88:360,120:360,72:200,72:300 --targets 4.0,5.0,6.0,7.0,9.0 --cycles 7500,8500,9500,10500,11500,12500 ) > $OUT/one_word_rule_check.txt 2>&1
grep -v amdgpu.ids $OUT/one_word_rule_check.txt | cut -c1-1000
( time timeout 1500 python3 profiles/phase_rule_check.py ) > $OUT/phase_rule_check.txt 2>&1
echo "phase_rule_check rc=$?" >> $OUT/phase_rule_check.txt
tail -n 22 $OUT/phase_rule_check.txt | cut -c1-300
RB_BENCH_DETAIL=$OUT/bench_w1_64mib.detail.json timeout 600 python3 bench.py --workload w1_64mib --steps 5 --warmup 2 --cpu-seconds 5 --no-latency > $OUT/bench_w1_64mib.json 2> /dev/null
cat $OUT/bench_w1_64mib.json | cut -c1-1500

#!/bin/bash
# r05 session 18: the tree with the folded decision: 1000-seed fuzz soak, 200 k micro-batches through the latency kernel, smoke, the driver's command
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05l
mkdir -p $OUT
cd $R
( time RB_FUZZ_SEEDS=1000 timeout 2400 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -x ) > $OUT/fuzz.txt 2>&1; tail -n 5 $OUT/fuzz.txt | cut -c1-200
( time timeout 900 python3 profiles/soak_split.py ) > $OUT/soak_split.txt 2>&1; grep -v amdgpu.ids $OUT/soak_split.txt | tail -n 6
( time timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.txt 2>&1; tail -n 2 $OUT/smoke.txt
( time RB_BENCH_DETAIL=$OUT/bench_default.json timeout 1800 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_default_line.json 2> $OUT/bench_default.err
echo "rc=$? line bytes $(wc -c < $OUT/bench_default_line.json)"; tail -n 4 $OUT/bench_default.err
cat $OUT/bench_default_line.json

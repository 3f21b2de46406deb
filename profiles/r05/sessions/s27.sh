#!/bin/bash
# r05 session 27: the final tree (threaded loader, replicas side by side): gpuperf, smoke, the driver's command
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05u
mkdir -p $OUT
cd $R
( time timeout 1500 python3 -m pytest tests -m gpuperf -q ) > $OUT/pytest_gpuperf.txt 2>&1
tail -n 4 $OUT/pytest_gpuperf.txt | cut -c1-200
( time timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.txt 2>&1; tail -n 2 $OUT/smoke.txt
( time RB_BENCH_DETAIL=$OUT/bench_default.json timeout 1800 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_default_line.json 2> $OUT/bench_default.err
echo "rc=$? line bytes $(wc -c < $OUT/bench_default_line.json)"; tail -n 4 $OUT/bench_default.err
cut -c1-1500 $OUT/bench_default_line.json

#!/bin/bash
# r05 session 1: the tree after the bench-line rework -- GPU suite, then the driver's own command, timed
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s01
mkdir -p $OUT
cd $R
( time timeout 1500 python3 -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.txt 2>&1
tail -n 5 $OUT/pytest_gpu.txt | cut -c1-300
( time RB_BENCH_DETAIL=$OUT/bench_default.detail.json timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 600 $OUT/bench_default.err
wc -c $OUT/bench_default.json
cat $OUT/bench_default.json
python3 - <<PY
import json
d=json.load(open("$OUT/bench_default.detail.json"))
print("bench_seconds", d.get("bench_seconds"))
for k,v in d["other_configs"].items():
    print(k, v.get("leg_seconds"), v.get("error"))
PY

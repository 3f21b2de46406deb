#!/bin/bash
# Deep parity of the round-2 tree: long oracle legs on every bench workload + a 2000-seed fuzz soak (new kernels forced).
OUT=gpurun_out/r02deep; mkdir -p $OUT
for w in readme c1 c2 c4 c3; do
  secs=60; [ $w = c3 -o $w = c4 ] && secs=90
  timeout 900 python3 bench.py --workload $w --steps 3 --warmup 1 --cpu-seconds $secs --no-latency --no-extras > $OUT/$w.json 2> $OUT/$w.err
  python3 -c "
import json
d=json.load(open('$OUT/$w.json')); print('$w', round(d['value']), d['parity'], d['config']['decisions'], 'cpu', round(d['cpu_baseline']['value']))"
done
RB_FUZZ_SEEDS=2000 timeout 1200 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python3 profiles/soak_split.py 2>&1 | tail -3
( time python3 bench.py --steps 20 --warmup 5 > $OUT/default.json 2> $OUT/default.err ) 2>&1 | grep real

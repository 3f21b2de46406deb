#!/bin/bash
# Deep parity of the round-3 tree: long oracle legs on every bench workload (the narrow ones at 250 AND 360 bp: the six-tile
# kernels), a 1000-seed x 2 N-rule fuzz soak (phased kernels forced), 200 k micro-batches through the latency kernel.
OUT=gpurun_out/r03deep; mkdir -p $OUT
run() { # tag secs args...
  local tag=$1 secs=$2; shift 2
  timeout 1200 python3 bench.py "$@" --steps 3 --warmup 1 --cpu-seconds $secs --no-latency > $OUT/$tag.json 2> $OUT/$tag.err
  python3 -c "
import json
d=json.loads([l for l in open('$OUT/$tag.json') if l.startswith('{')][-1]); print('$tag', round(d['value']), d['parity'], d['config']['decisions'], 'cpu', round(d['cpu_baseline']['value']))"
}
run readme250 60 --workload readme
run readme360 60 --workload readme --read-len 360
run readme600 40 --workload readme --read-len 600 --reads 500000
run c1 60 --workload c1
run c1_250 40 --workload c1 --read-len 250
run w1_64mib_250 40 --workload w1_64mib
run w1_64mib_360 40 --workload w1_64mib --read-len 360
run targets3_250 40 --workload targets3
run targets3_360 40 --workload targets3 --read-len 360
run deplete_target_250 40 --workload deplete_target
run c2 60 --workload c2
run c4 90 --workload c4
run c3 90 --workload c3 --reads 2000000
RB_FUZZ_SEEDS=1000 timeout 2400 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python3 profiles/soak_split.py 2>&1 | tail -3

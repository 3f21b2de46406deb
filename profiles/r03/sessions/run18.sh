#!/bin/bash
# round 3, GPU session 18: one-word 20 MB table at 250 bp (an interpolated point of the window rule)
set -u
O=gpurun_out/r03
mkdir -p $O
one() { local tag=$1; shift
  python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/k_$tag.json 2>> $O/tune.err
  python - <<PY
import json
d=json.load(open("$O/k_$tag.json"))
print("$tag", round(d["value"]/1e6,2), "M reads/s", round(d["roofline"]["avg_kernel_ms"],2), "ms")
PY
}
one c1_250_default --workload c1 --read-len 250
for ticks in 800 900 1000 1100 1200 1300; do one c1_250_t$ticks --workload c1 --read-len 250 --phased 6,32,$ticks,0; done
one dep_600_default --workload mock_deplete --read-len 600 --reads 500000
for ticks in 450 525 600 700; do one dep_600_t$ticks --workload mock_deplete --read-len 600 --reads 500000 --phased 6,32,$ticks,0; done

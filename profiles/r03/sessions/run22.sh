#!/bin/bash
# round 3, GPU session 22: 16 slices instead of 8 for the 20 MB tables (smaller slices, shorter windows)
set -u
O=gpurun_out/r03
mkdir -p $O
one() { local tag=$1; shift
  python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/g_$tag.json 2>> $O/tune.err
  python - <<PY
import json
d=json.load(open("$O/g_$tag.json"))
print("$tag", round(d["value"]/1e6,2), "M reads/s", round(d["roofline"]["avg_kernel_ms"],2), "ms")
PY
}
export RB_PHASE_MAX_SLICES=16
for ticks in 300 400 500 600 750; do
  one s16_c1_360_t$ticks --workload c1 --phased 6,32,$ticks,0
  one s16_c1_250_t$ticks --workload c1 --read-len 250 --phased 6,32,$ticks,0
  one s16_dep_250_t$ticks --workload mock_deplete --phased 6,32,$ticks,0
  one s16_dep_360_t$ticks --workload mock_deplete --read-len 360 --phased 6,32,$ticks,0
  one s16_t1_250_t$ticks --workload mock_t1 --phased 6,32,$ticks,0
done

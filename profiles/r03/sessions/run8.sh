#!/bin/bash
# round 3, GPU session 8: per-kernel window sweep (single filters)
set -u
O=gpurun_out/r03
mkdir -p $O
one() { local tag=$1; shift
  python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/u_$tag.json 2>> $O/tune.err
  python - <<PY
import json
d=json.load(open("$O/u_$tag.json"))
print("$tag", round(d["value"]/1e6,2), "M reads/s", round(d["roofline"]["avg_kernel_ms"],2), "ms")
PY
}
for ticks in 800 900 1000 1100 1250; do RB_SIX_TILES=3 one dep360_s3_t$ticks --workload mock_deplete --read-len 360 --phased 6,32,$ticks,0; done
for ticks in 525 600 675 750 825; do RB_SIX_TILES=1 one dep360_s2_t$ticks --workload mock_deplete --read-len 360 --phased 6,32,$ticks,0; done
for ticks in 575 625 675 725 775; do RB_SIX_TILES=1 one t1_360_s3_t$ticks --workload mock_t1 --read-len 360 --phased 6,32,$ticks,0; done
for ticks in 450 500 550 600; do RB_SIX_TILES=0 one c1_s2_t$ticks --workload c1 --phased 6,32,$ticks,0; done
for ticks in 900 1000 1100 1250; do RB_SIX_TILES=2 one c1_s3_t$ticks --workload c1 --phased 6,32,$ticks,0; done
for ticks in 650 700 750; do one dep250_t$ticks --workload mock_deplete --phased 6,32,$ticks,0; done

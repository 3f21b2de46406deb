#!/bin/bash
# round 3, GPU session 7: window-length sweep with the bounds-checked buffer gathers
set -u
O=gpurun_out/r03
mkdir -p $O
one() { local tag=$1; shift
  python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/s_$tag.json 2>> $O/tune.err
  python - <<PY
import json
d=json.load(open("$O/s_$tag.json"))
print("$tag", round(d["value"]/1e6,2), "M reads/s", round(d["roofline"]["avg_kernel_ms"],2), "ms")
PY
}
for ticks in 575 625 675 725 775 850; do
  one readme250_t$ticks --workload readme --phased 6,32,$ticks,0
  RB_SIX_TILES=1 one readme360_six1_t$ticks --workload readme --read-len 360 --phased 6,32,$ticks,0
  RB_SIX_TILES=3 one readme360_six3_t$ticks --workload readme --read-len 360 --phased 6,32,$ticks,0
  RB_SIX_TILES=0 one c1_six0_t$ticks --workload c1 --phased 6,32,$ticks,0
  RB_SIX_TILES=2 one c1_six2_t$ticks --workload c1 --phased 6,32,$ticks,0
  one mockdep250_t$ticks --workload mock_deplete --phased 6,32,$ticks,0
  one mockt1_250_t$ticks --workload mock_t1 --phased 6,32,$ticks,0
done

#!/bin/bash
# round 3, GPU session 20: slices served in the order of the CLOCK (a late wave joins the current slice): parity + do the cliffs go?
set -u
O=gpurun_out/r03
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_properties.py -m gpu -q -x 2>&1 | tail -3
one() { local tag=$1; shift
  python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/f_$tag.json 2>> $O/tune.err
  python - <<PY
import json
d=json.load(open("$O/f_$tag.json"))
print("$tag", round(d["value"]/1e6,2), "M reads/s", round(d["roofline"]["avg_kernel_ms"],2), "ms")
PY
}
for ticks in 450 550 650 750 850; do one t1_250_t$ticks --workload mock_t1 --phased 6,32,$ticks,0; done
for ticks in 600 700 800 900 1000; do one dep_250_t$ticks --workload mock_deplete --phased 6,32,$ticks,0; done
for ticks in 550 650 750 850 950; do one t1_360_t$ticks --workload mock_t1 --read-len 360 --phased 6,32,$ticks,0; done
for ticks in 700 800 900 1000 1100; do one dep_360_t$ticks --workload mock_deplete --read-len 360 --phased 6,32,$ticks,0; done
for ticks in 700 850 1000 1150 1300; do one c1_360_t$ticks --workload c1 --phased 6,32,$ticks,0; done
for ticks in 600 750 900 1050 1200; do one c1_250_t$ticks --workload c1 --read-len 250 --phased 6,32,$ticks,0; done

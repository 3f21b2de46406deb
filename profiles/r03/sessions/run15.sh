#!/bin/bash
# round 3, GPU session 15: complemented image (OR accumulation): parity + numbers + window check
set -u
O=gpurun_out/r03
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_properties.py -m gpu -q -x 2>&1 | tail -3
one() { local tag=$1; shift
  python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/n_$tag.json 2>> $O/tune.err
  python - <<PY
import json
d=json.load(open("$O/n_$tag.json"))
print("$tag", round(d["value"]/1e6,2), "M reads/s", round(d["roofline"]["avg_kernel_ms"],2), "ms")
PY
}
one readme250 --workload readme
one readme360 --workload readme --read-len 360
one c1 --workload c1
for ticks in 600 675 750 825; do one t1_250_t$ticks --workload mock_t1 --phased 6,32,$ticks,0; done
for ticks in 750 825 900 975; do one dep_250_t$ticks --workload mock_deplete --phased 6,32,$ticks,0; done
for ticks in 750 825 900 975; do one t1_360_t$ticks --workload mock_t1 --read-len 360 --phased 6,32,$ticks,0; done
for ticks in 900 975 1050 1125; do one dep_360_t$ticks --workload mock_deplete --read-len 360 --phased 6,32,$ticks,0; done
for ticks in 1100 1175 1250 1325; do one c1_t$ticks --workload c1 --phased 6,32,$ticks,0; done

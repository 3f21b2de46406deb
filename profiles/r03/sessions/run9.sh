#!/bin/bash
# round 3, GPU session 9: the built-in window rule (defaults) + width of the optima
set -u
O=gpurun_out/r03
mkdir -p $O
one() { local tag=$1; shift
  python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/v_$tag.json 2>> $O/tune.err
  python - <<PY
import json
d=json.load(open("$O/v_$tag.json"))
print("$tag", round(d["value"]/1e6,2), "M reads/s", round(d["roofline"]["avg_kernel_ms"],2), "ms")
PY
}
one readme250_default --workload readme
one readme360_default --workload readme --read-len 360
one c1_default --workload c1
one dep250_default --workload mock_deplete
one dep360_default --workload mock_deplete --read-len 360
one t1_250_default --workload mock_t1
one t1_360_default --workload mock_t1 --read-len 360
for ticks in 925 950 975 1025 1050 1075; do
  one dep360_s3_t$ticks --workload mock_deplete --read-len 360 --phased 6,32,$ticks,0
  one c1_s3_t$ticks --workload c1 --phased 6,32,$ticks,0
done
for ticks in 650 700; do one t1_360_s3_t$ticks --workload mock_t1 --read-len 360 --phased 6,32,$ticks,0; done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_properties.py -m gpu -q -x 2>&1 | tail -3

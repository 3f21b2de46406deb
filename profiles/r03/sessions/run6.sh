#!/bin/bash
# round 3, GPU session 6: bounds-checked buffer loads in the phased gathers: parity, then throughput by six-tile setting
set -u
O=gpurun_out/r03
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_properties.py -m gpu -q -x -k "raw_max or random_geometry or narrow or n_reads or packed or odd_stride" 2>&1 | tail -5
one() { local tag=$1; shift
  python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/b_$tag.json 2>> $O/tune.err
  python - <<PY
import json
d=json.load(open("$O/b_$tag.json"))
print("$tag", round(d["value"]/1e6,2), "M reads/s", round(d["roofline"]["avg_kernel_ms"],2), "ms", d["config"]["decisions"])
PY
}
one readme250 --workload readme
for six in 0 1 3; do
  export RB_SIX_TILES=$six
  one readme360_six$six --workload readme --read-len 360
  one c1_six$six --workload c1
done
unset RB_SIX_TILES
for ticks in 350 450 575 700; do
  one readme250_t$ticks --workload readme --phased 6,32,$ticks,0
  RB_SIX_TILES=3 one readme360_six3_t$ticks --workload readme --read-len 360 --phased 6,32,$ticks,0
done

#!/bin/bash
# round 3, GPU session 2: the full -m gpu suite; six-tile kernels at 360 bp (RB_SIX_TILES 0/1/2) with a window sweep
set -u
O=gpurun_out/r03
mkdir -p $O
( time python -m pytest tests -m gpu -q ) > $O/pytest_gpu.txt 2>&1
tail -8 $O/pytest_gpu.txt
one() { # env-name tag args...
  local tag=$1; shift
  python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/t_$tag.json 2>> $O/tune.err
  python - <<PY
import json
d=json.load(open("$O/t_$tag.json"))
print("$tag", round(d["value"]/1e6,2), "M reads/s", round(d["roofline"]["avg_kernel_ms"],2), "ms", d["config"]["decisions"])
PY
}
for six in 0 1 2; do
  export RB_SIX_TILES=$six
  one readme360_six$six --workload readme --read-len 360
  one c1_six$six --workload c1
done
export RB_SIX_TILES=2
for ticks in 450 575 700 850 1000; do
  one readme360_six2_t$ticks --workload readme --read-len 360 --phased 6,32,$ticks,0
  one c1_six2_t$ticks --workload c1 --phased 6,32,$ticks,0
done
unset RB_SIX_TILES
one readme250 --workload readme

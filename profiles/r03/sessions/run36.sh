#!/bin/bash
# round 3, GPU session 36: window length on the README deplete filter itself (122 bins, two-word blocks, 19.9 MiB), both slice sizes
set -u
O=gpurun_out/r03
mkdir -p $O
one() { local tag=$1; shift
  python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/k_$tag.json 2>> $O/k.err
  python - <<PY
import json
d=json.load(open("$O/k_$tag.json"))
print("$tag", round(d["value"]/1e6,2), "M reads/s", round(d["roofline"]["avg_kernel_ms"],2), "ms")
PY
}
one dep250_rule --workload mock_deplete
one dep360_rule --workload mock_deplete --read-len 360
for lg in 21 22; do
  export RB_PHASE_SLICE_LOG2=$lg
  for t in 300 350 400 450 500 600 700 750 800 850 900 950 1000 1100; do
    one dep250_s${lg}_t$t --workload mock_deplete --phased 6,128,$t,0
    one dep360_s${lg}_t$t --workload mock_deplete --read-len 360 --phased 6,128,$t,0
  done
done

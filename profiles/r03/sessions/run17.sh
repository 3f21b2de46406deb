#!/bin/bash
# round 3, GPU session 17: merged tables: parity, README shape with merge on/off (RB_MERGE env not available: bench has no flag -> use test hook env)
set -u
O=gpurun_out/r03
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "merged" 2>&1 | tail -5
one() { local tag=$1; shift
  python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/m_$tag.json 2>> $O/tune.err
  python - <<PY
import json
d=json.load(open("$O/m_$tag.json"))
print("$tag", round(d["value"]/1e6,2), "M reads/s", round(d["roofline"]["avg_kernel_ms"],2), "ms", d["roofline"]["kernel"])
PY
}
for m in 1 0; do
  export RB_MERGE=$m
  one readme250_merge$m --workload readme
  one readme360_merge$m --workload readme --read-len 360
  one readme1500_merge$m --workload readme --read-len 1500 --reads 200000
done

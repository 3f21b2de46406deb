#!/bin/bash
# round 3, GPU session 16: reads per call at which the phased kernels start to pay (README shape, 250 and 360 bp)
set -u
O=gpurun_out/r03
mkdir -p $O
one() { local tag=$1; shift
  python bench.py "$@" --warmup 3 --no-cpu-baseline --no-latency > $O/q_$tag.json 2>> $O/tune.err
  python - <<PY
import json
d=json.load(open("$O/q_$tag.json"))
print("$tag", round(d["value"]/1e6,2), "M reads/s", round(d["roofline"]["avg_kernel_ms"],3), "ms")
PY
}
for n in 2049 4096 8192 16384 65536; do
  for L in 250 360; do
    one readme${L}_n${n}_phased --workload readme --read-len $L --reads $n --steps 30 --phased 6,32,0,0
    one readme${L}_n${n}_plain --workload readme --read-len $L --reads $n --steps 30 --phased off
  done
done

#!/bin/bash
# round 3, GPU session 19: load-batch size vs occupancy in the PLAIN kernels (c3 / c4 / c2 / grch38): HALF = steps per batch
set -u
O=gpurun_out/r03
mkdir -p $O
one() { local tag=$1; shift
  python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/h_$tag.json 2>> $O/tune.err
  python - <<PY
import json
d=json.load(open("$O/h_$tag.json"))
print("$tag", round(d["value"]/1e6,3), "M reads/s", round(d["roofline"]["avg_kernel_ms"],2), "ms", round(d["roofline"]["frac"],4))
PY
}
for v in "8 2" "2 2" "4 4" "8 4"; do
  set -- $v
  touch readbouncer_amd/csrc/rb_kernels.hip
  make -C readbouncer_amd/csrc -j8 KFLAGS="-DRB_HALF1=$1 -DRB_HALF2=$2" > $O/build_var.log 2>&1 || { tail $O/build_var.log; exit 1; }
  one c3_H$1_$2 --workload c3 --reads 2000000
  one c2_H$1_$2 --workload c2
  one c4_H$1_$2 --workload c4
  one grch_H$1_$2 --workload grch38_f100k
done

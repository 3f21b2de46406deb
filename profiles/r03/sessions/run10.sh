#!/bin/bash
# round 3, GPU session 10: load-batch size vs occupancy in the 250 bp kernels (rebuilds the library per variant on the box)
set -u
O=gpurun_out/r03
mkdir -p $O
one() { local tag=$1; shift
  python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/w_$tag.json 2>> $O/tune.err
  python - <<PY
import json
d=json.load(open("$O/w_$tag.json"))
print("$tag", round(d["value"]/1e6,2), "M reads/s", round(d["roofline"]["avg_kernel_ms"],2), "ms")
PY
}
for v in "4 2" "2 1" "8 4"; do
  set -- $v
  touch readbouncer_amd/csrc/rb_kernels.hip
  make -C readbouncer_amd/csrc -j8 KFLAGS="-DRB_GATHER_B1=$1 -DRB_GATHER_KB1=$2" > $O/build_var.log 2>&1 || { tail $O/build_var.log; exit 1; }
  for ticks in 575 650 725 800 900; do
    one t1_B$1_t$ticks --workload mock_t1 --phased 6,32,$ticks,0
    one dep_KB$2_t$ticks --workload mock_deplete --phased 6,32,$ticks,0
  done
done

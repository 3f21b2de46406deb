#!/bin/bash
# round 3, GPU session 12: load-batch size of the GENERAL phased build (long reads / 3-8 word blocks)
set -u
O=gpurun_out/r03
mkdir -p $O
one() { local tag=$1; shift
  python bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $O/y_$tag.json 2>> $O/tune.err
  python - <<PY
import json
d=json.load(open("$O/y_$tag.json"))
print("$tag", round(d["value"]/1e6,3), "M reads/s", round(d["roofline"]["avg_kernel_ms"],2), "ms")
PY
}
for bg in 8 4 2; do
  touch readbouncer_amd/csrc/rb_kernels.hip
  make -C readbouncer_amd/csrc -j8 KFLAGS="-DRB_GATHER_BG=$bg" > $O/build_var.log 2>&1 || { tail $O/build_var.log; exit 1; }
  for ticks in 450 600 750; do
    one c1_len600_BG${bg}_t$ticks --workload c1 --read-len 600 --reads 500000 --phased 6,32,$ticks,0
    one t1_len1500_BG${bg}_t$ticks --workload mock_t1 --read-len 1500 --reads 200000 --phased 6,32,$ticks,0
    one dep_len1500_BG${bg}_t$ticks --workload mock_deplete --read-len 1500 --reads 200000 --phased 6,32,$ticks,0
  done
done

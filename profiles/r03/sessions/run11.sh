#!/bin/bash
# round 3, GPU session 11: B1=2, KB1=1, B3=2 (more waves per SIMD, smaller load batches): window sweep
set -u
O=gpurun_out/r03
mkdir -p $O
one() { local tag=$1; shift
  python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/x_$tag.json 2>> $O/tune.err
  python - <<PY
import json
d=json.load(open("$O/x_$tag.json"))
print("$tag", round(d["value"]/1e6,2), "M reads/s", round(d["roofline"]["avg_kernel_ms"],2), "ms")
PY
}
touch readbouncer_amd/csrc/rb_kernels.hip
make -C readbouncer_amd/csrc -j8 KFLAGS="-DRB_GATHER_B1=2 -DRB_GATHER_KB1=1 -DRB_GATHER_B3=2" > $O/build_var.log 2>&1 || { tail $O/build_var.log; exit 1; }
for ticks in 700 750 800; do one t1_250_t$ticks --workload mock_t1 --phased 6,32,$ticks,0; done
for ticks in 850 950 1050 1150; do one dep_250_t$ticks --workload mock_deplete --phased 6,32,$ticks,0; done
for ticks in 675 750 825 900 1000; do one t1_360_t$ticks --workload mock_t1 --read-len 360 --phased 6,32,$ticks,0; done
for ticks in 1000 1100 1200 1350; do one c1_t$ticks --workload c1 --phased 6,32,$ticks,0; done

#!/bin/bash
# round 3, GPU session 3: dedicated-pump probe; the one test that failed on a test bug; read-len 1500 throughput
set -u
O=gpurun_out/r03
mkdir -p $O
hipcc -O3 --offload-arch=gfx950 profiles/gather_probe.hip -o /tmp/gather_probe || exit 1
timeout 600 /tmp/gather_probe phased_pump > $O/gather_phased_pump.txt 2>&1
cat $O/gather_phased_pump.txt
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "odd_stride or pool" -s 2>&1 | tail -6
for wl in c3 c4 readme; do
  python bench.py --workload $wl --read-len 1500 --reads 200000 --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $O/len1500_$wl.json 2>> $O/tune.err
  python - <<PY
import json
d=json.load(open("$O/len1500_$wl.json"))
print("$wl 1500bp", round(d["value"]/1e6,3), "M reads/s", round(d["roofline"]["avg_kernel_ms"],2), "ms  frac", round(d["roofline"]["frac"],3))
PY
done

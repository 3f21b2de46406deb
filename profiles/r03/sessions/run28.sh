#!/bin/bash
# round 3, GPU session 28: the refitted window rule (base + cycle / n, per-shape size caps) against the plain kernel
set -u
O=gpurun_out/r03
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "cap_is_not_made or merge_when" 2>&1 | tail -40 > $O/rule2_tests.txt
cat $O/rule2_tests.txt
S="7,8,9,10.5,12,14,16,18,20,24,28,32,40,48,64,96,127"
timeout 600 python profiles/r03/slice_size_sweep.py 1 250,360 $S 22 600 > $O/rule2_w1.txt 2>&1
timeout 600 python profiles/r03/slice_size_sweep.py 2 250,360 $S 22 600 > $O/rule2_w2.txt 2>&1
timeout 600 python profiles/r03/slice_size_sweep.py 1 500,1000 8,12,16,24,32,48,64 22 600 > $O/rule2_w1_long.txt 2>&1
timeout 600 python profiles/r03/slice_size_sweep.py 2 500,1000 8,12,16,24,32,48,64 22 600 > $O/rule2_w2_long.txt 2>&1
for w in readme c1; do
 for L in 250 360; do
  python bench.py --workload $w --read-len $L --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/rule2_${w}_$L.json 2>> $O/rule2.err
  RB_MERGE=0 python bench.py --workload $w --read-len $L --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/rule2_${w}_${L}_apart.json 2>> $O/rule2.err
 done
done
python - <<PY
import json
for w in ("readme","c1"):
  for L in (250,360):
    for s in ("","_apart"):
        d=json.load(open("$O/rule2_%s_%d%s.json"%(w,L,s)))
        print(w,L,s, round(d["value"]/1e6,2), "M reads/s", round(d["roofline"]["avg_kernel_ms"],2), "ms")
PY

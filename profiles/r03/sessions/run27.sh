#!/bin/bash
# round 3, GPU session 27: the new slice / window rule (slices of 2 or 4 MiB, cycle / n windows, phased up to 128 MiB) and the
# merge cost model: parity, the rule against the sweep's best, the README shape
set -u
O=gpurun_out/r03
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -5 > $O/rule_tests.txt
cat $O/rule_tests.txt
S="7,8,9,10.5,12,14,16,18,20,24,28,32,40,48,64,96,127"
timeout 600 python profiles/r03/slice_size_sweep.py 1 250,360 $S 22 600 > $O/rule_w1.txt 2>&1
timeout 600 python profiles/r03/slice_size_sweep.py 2 250,360 $S 22 600 > $O/rule_w2.txt 2>&1
timeout 600 python profiles/r03/slice_size_sweep.py 1 500,1000 8,12,16,24,32,48,96 22 600 > $O/rule_w1_long.txt 2>&1
timeout 600 python profiles/r03/slice_size_sweep.py 2 500,1000 8,12,16,24,32,48,96 21,22 300,450,600,800,1000,1400 > $O/rule_w2_long.txt 2>&1
grep -h "rule\|plain" $O/rule_w1.txt $O/rule_w2.txt $O/rule_w1_long.txt
cat $O/rule_w2_long.txt
for w in readme readme_360bp c1; do
  python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/rule_$w.json 2>> $O/rule.err
  RB_MERGE=0 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/rule_${w}_apart.json 2>> $O/rule.err
done
python - <<PY
import json
for w in ("readme","readme_360bp","c1"):
    for s in ("","_apart"):
        d=json.load(open("$O/rule_%s%s.json"%(w,s)))
        print(w+s, round(d["value"]/1e6,2), "M reads/s", round(d["roofline"]["avg_kernel_ms"],2), "ms")
PY

#!/bin/bash
# round 3, GPU session 56: the refitted window rule for the builds with more waves per SIMD, against the sweep's best; suite
set -u
O=gpurun_out/r03
mkdir -p $O
S="1.5,2,3,4,5,6,7,8,9,10.5,12,14,16,18,20,24,28,32,40,48,64,96,127"
timeout 900 python profiles/r03/slice_size_sweep.py 1 150,200,250,300,360 $S 22 600 > $O/occ_rule_w1.txt 2>&1
timeout 900 python profiles/r03/slice_size_sweep.py 2 150,200,250,300,360 $S 22 600 > $O/occ_rule_w2.txt 2>&1
python -m pytest tests -q -m gpu -x 2>&1 | tail -3
for w in readme c1 targets3 deplete_target; do for L in 250 360; do
  python bench.py --workload $w --read-len $L --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/o_${w}_$L.json 2>> $O/o.err
  python -c "
import json; d=json.load(open('$O/o_${w}_$L.json')); print('$w $L', round(d['value']/1e6,2), 'M reads/s', round(d['roofline']['avg_kernel_ms'],2), 'ms')"
done; done
RB_MERGE=0 python bench.py --workload readme --steps 5 --warmup 2 --no-cpu-baseline --no-latency | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('readme apart 250', round(d['value']/1e6,2), round(d['roofline']['avg_kernel_ms'],2))"
RB_MERGE=0 python bench.py --workload readme --read-len 360 --steps 5 --warmup 2 --no-cpu-baseline --no-latency | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('readme apart 360', round(d['value']/1e6,2), round(d['roofline']['avg_kernel_ms'],2))"

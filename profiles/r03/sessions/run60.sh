#!/bin/bash
# round 3, GPU session 60: final tree with the three-word three-tile build: suite, smoke, the narrow workloads, rule check on three-word tables
set -u
O=gpurun_out/r03
mkdir -p $O
python -m pytest tests -q -m gpu -x 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for w in targets3 deplete_target readme c1; do for L in 250 360; do
  python bench.py --workload $w --read-len $L --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/q_${w}_$L.json 2>> $O/q.err
  python -c "
import json; d=json.load(open('$O/q_${w}_$L.json')); print('$w $L', round(d['value']/1e6,2), 'M reads/s', round(d['roofline']['avg_kernel_ms'],2), 'ms', d['roofline']['kernel'])"
done; done
timeout 900 python profiles/r03/slice_size_sweep.py 3 360,500 6,9,12,18,24,30,36 22 500 > $O/w3r_rule.txt 2>&1
grep -h "rule" $O/w3r_rule.txt

#!/bin/bash
# round 3, GPU session 63: two-word 250 bp tables of 15-26 MiB on 4 MiB slices (flat optimum) instead of the narrow dip of the 2 MiB slices
set -u
O=gpurun_out/r03
mkdir -p $O
timeout 900 python profiles/r03/slice_size_sweep.py 2 250 12,14,16,18,20,22,24,28 22 1000 > $O/w2_switch.txt 2>&1
grep -h "rule" $O/w2_switch.txt
for i in 1 2 3; do
python bench.py --workload deplete_target --steps 5 --warmup 2 --no-cpu-baseline --no-latency | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('deplete_target 250', round(d['value']/1e6,2), round(d['roofline']['avg_kernel_ms'],2))"
done
RB_MERGE=0 python bench.py --workload readme --steps 5 --warmup 2 --no-cpu-baseline --no-latency | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('readme apart 250', round(d['value']/1e6,2), round(d['roofline']['avg_kernel_ms'],2))"
python -m pytest tests -q -m gpu -x 2>&1 | tail -2

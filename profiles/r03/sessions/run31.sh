#!/bin/bash
# round 3, GPU session 31: full GPU suite + smoke on the tree after the slice / window rule, merge cost model and their tests
set -u
O=gpurun_out/r03
mkdir -p $O
python -m pytest tests -q -m gpu 2>&1 | tail -6 > $O/pytest_gpu_full_i.txt
cat $O/pytest_gpu_full_i.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for w in c1; do for L in 250 360; do
  python bench.py --workload $w --read-len $L --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/i_${w}_$L.json 2>> $O/i.err
  python -c "
import json; d=json.load(open('$O/i_${w}_$L.json')); print('$w $L', round(d['value']/1e6,2), 'M reads/s', round(d['roofline']['avg_kernel_ms'],2), 'ms')"
done; done

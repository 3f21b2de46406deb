#!/bin/bash
# round 3, GPU session 53: three-word blocks (three merged targets) with a build of their own: 94 registers, five waves per SIMD
set -u
O=gpurun_out/r03
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "raw_max or fuzz or merged or merge" 2>&1 | tail -3
for L in 250 360; do
  python bench.py --workload targets3 --read-len $L --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/n_targets3_$L.json 2>> $O/n.err
  python -c "
import json; d=json.load(open('$O/n_targets3_$L.json')); print('targets3 $L', round(d['value']/1e6,2), 'M reads/s', round(d['roofline']['avg_kernel_ms'],2), 'ms')"
done
T="500"
timeout 900 python profiles/r03/slice_size_sweep.py 3 250 2,3,6,9,12,18,24,30,36 22 $T > $O/w3_five_waves.txt 2>&1
grep -h "rule\|plain\|slices" $O/w3_five_waves.txt

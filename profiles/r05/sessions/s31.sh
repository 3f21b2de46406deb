#!/bin/bash
# r05 session 31: the two-word 250 bp build at eight waves per SIMD (64 registers, 52 bytes of scratch) against the shipped seven (72 registers, no scratch):
# more reads per pass = fewer table reloads per read -- does it outweigh the spills?  An experimental library (RB_AMD_LIBRARY), windows swept.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05y
mkdir -p $OUT
cd $R
for lib in libreadbouncer_amd.so libreadbouncer_amd_w8.so; do
  for w in targets3 deplete_target; do
    for rep in 1 2; do
      RB_BENCH_NO_SUPERVISOR=1 RB_AMD_LIBRARY=$R/readbouncer_amd/$lib RB_BENCH_DETAIL=$OUT/$lib.$w.$rep.json timeout 600 python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-latency > $OUT/$lib.$w.$rep.line 2> $OUT/$lib.$w.$rep.err
      python3 -c "
import json,sys
d=json.load(open('$OUT/$lib.$w.$rep.json')); r=d['roofline']
print('$lib $w rep $rep: %.1f M reads/s  K1 %.3f ms  request_bound_frac %.3f  parity %s' % (d['value']/1e6, r['avg_kernel_ms'], r['request_bound']['request_bound_frac'], d.get('parity',{}).get('decision_mismatches')))"
    done
  done
done 2>&1 | tee $OUT/waves8.txt

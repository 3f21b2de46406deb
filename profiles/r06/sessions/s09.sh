#!/bin/bash
# r06 session 9: streaming form, tickets of 8 / 32 reads, with and without staggered starts (experimental libraries)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06i
mkdir -p $OUT
cd $R
for v in stag1c8 stag1c32 stag0c32; do
  echo "== $v"
  RB_AMD_LIBRARY=$R/readbouncer_amd/exp/libreadbouncer_amd_$v.so timeout 600 python3 profiles/multi_reads_sweep.py --workloads deplete_target --rpw 33 --skew 0,2 --factors 0.9,0.95,1.0,1.05,1.1,1.15,1.2 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $OUT/stream_variants.txt

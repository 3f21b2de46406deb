#!/bin/bash
# r06 session 4: counters of the shipped two-word build against the multi-read builds (R = 1 and R = 2, OR form, XCD time skew), deplete_target
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06d
mkdir -p $OUT
cd $R
for cfg in "shipped 0 0" "r1_skew2 1 2" "r2_skew2 2 2"; do
  set -- $cfg
  bash profiles/r06/collect_pmc_multi.sh $1 deplete_target $2 $3 $OUT 2>&1 | tee -a $OUT/pmc_multi.txt
done

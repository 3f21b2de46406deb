#!/bin/bash
# r06 session 29: does the equal cut's window hold at other batch sizes?  (the rule was fitted at 500 000 reads per launch and checked at 1 M)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06o
mkdir -p $OUT
cd $R
export RB_TUNING_ENV=1
for n in 100000 250000 500000 2000000 4000000; do
  echo "== $n reads per launch"
  timeout 900 python3 profiles/multi_reads_sweep.py --reads $n --workloads deplete_target,deplete_target360,c1_360 --rpw 1 --skew 2 --factors 0.85,0.92,0.96,1.0,1.04,1.08,1.15 2>&1 | grep -v amdgpu.ids
done 2>&1 | tee $OUT/equal_cut_by_batch_size.txt | cut -c1-260
echo done

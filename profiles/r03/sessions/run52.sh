#!/bin/bash
# round 3, GPU session 52: counters and kernel stats of the narrow merged tables (three targets; deplete + target), merged and apart
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03n
mkdir -p $O
for w in targets3 deplete_target; do
  bash $R/profiles/collect_pmc.sh $w 1000000 $O/pmc_$w > /dev/null 2>&1
  RB_MERGE=0 bash $R/profiles/collect_pmc.sh $w 1000000 $O/pmc_${w}_apart > /dev/null 2>&1
done
cd /tmp && export TMPDIR=/tmp
for w in targets3 deplete_target; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$w -- python3 $R/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-latency > $O/stats_$w.log 2>&1
  RB_MERGE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_${w}_apart -- python3 $R/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-latency > $O/stats_${w}_apart.log 2>&1
done
cd $R
for w in targets3 deplete_target; do
  python bench.py --workload $w --steps 5 --warmup 2 > $O/bench_$w.json 2>> $O/err.txt
done
ls $O

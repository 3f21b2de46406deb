#!/bin/bash
# round 3, GPU session 30: counters of a 64 MiB one-word filter, phased (16 slices of 4 MiB) against the plain kernel
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03i
mkdir -p $O
bash $R/profiles/collect_pmc.sh w1_64mib 1000000 $O/pmc_w1_64mib > /dev/null 2>&1
bash $R/profiles/collect_pmc.sh w1_64mib 1000000 $O/pmc_w1_64mib_plain "--phased 0,0,300,3" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_w1_64mib -- python3 $R/bench.py --workload w1_64mib --steps 5 --warmup 1 --no-cpu-baseline --no-latency > $O/stats_w1_64mib.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_w1_64mib_plain -- python3 $R/bench.py --workload w1_64mib --phased 0,0,300,3 --steps 5 --warmup 1 --no-cpu-baseline --no-latency > $O/stats_w1_64mib_plain.log 2>&1
cd $R
python bench.py --workload w1_64mib --steps 5 --warmup 2 > $O/bench_w1_64mib.json 2>> $O/err.txt
python bench.py --workload w1_64mib --phased 0,0,300,3 --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/bench_w1_64mib_plain.json 2>> $O/err.txt
tail -c 600 $O/bench_w1_64mib.json; echo; tail -c 400 $O/bench_w1_64mib_plain.json
for w in w1_64mib w1_64mib_plain; do cat $O/pmc_$w/ea.summary.csv | cut -d, -f1-3,12-20 | head -4; done

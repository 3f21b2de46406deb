#!/bin/bash
# rocprofv3 --kernel-trace --stats summaries for the other workloads (config 3, 4, 5)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/stats
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -- python3 $R/bench.py --workload c3 --reads 2000000 --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $OUT/c3.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -- python3 $R/bench.py --workload c4 --reads 2000000 --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $OUT/c4.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5 -- python3 $R/bench.py --workload c5 --replay-seconds 1.0 > $OUT/c5.log 2>&1
for w in c3 c4 c5; do f=$(find $OUT/$w -name "*kernel_stats.csv" | head -1); echo "== $w"; grep -E "^\"Name|rb::" $f | cut -c1-60,200-330 | head -8; done

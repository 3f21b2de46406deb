#!/bin/bash
# r05 session 29: kernel tables of the headline workload and of the driver's command on the final tree (rocprofv3 --kernel-trace --stats)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05w
mkdir -p $OUT
T="timeout 1800"
cd /tmp && export TMPDIR=/tmp
export RB_BENCH_NO_SUPERVISOR=1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3 -- python3 $R/bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $OUT/stats_c3.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/stats_default.log 2>&1
unset RB_BENCH_NO_SUPERVISOR
find $OUT -name "*.db" -delete; find $OUT -path "*stats_*" -name "*kernel_trace.csv" -delete; find $OUT -path "*stats_*" -name "*agent_info.csv" -delete
for w in c3 default; do f=$(find $OUT/stats_$w -name "*kernel_stats.csv" | head -1); echo "== $w $f"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ibf_count" in r["Name"] or "probe" in r["Name"] or "decide" in r["Name"] or "copy_from_host" in r["Name"]:
        print("  ", r["Name"].split("(")[0][-70:], r["Calls"], "avg ms %.4f" % (float(r["AverageNs"])/1e6))
PY
done
tail -c 600 $OUT/stats_c3.log | cut -c1-600
du -sh $OUT

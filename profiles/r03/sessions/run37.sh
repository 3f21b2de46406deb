#!/bin/bash
# round 3, GPU session 37 (run twice; the second time on the final tree: two-word blocks take 4 MiB slices from 18.5 MiB on): GPU suite, and the README filters one after the
# other (RB_MERGE=0) again -- kernel stats and counters
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03k
mkdir -p $O
cd $R
python3 -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/pytest_gpu_full.txt
cat $O/pytest_gpu_full.txt
rm -rf $O/pmc_readme_phased $O/pmc_readme360_phased $O/stats_readme_phased $O/stats_readme360_phased
RB_MERGE=0 bash $R/profiles/collect_pmc.sh readme 1000000 $O/pmc_readme_phased > /dev/null 2>&1
RB_MERGE=0 bash $R/profiles/collect_pmc.sh readme 1000000 $O/pmc_readme360_phased "--read-len 360" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
RB_MERGE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_readme_phased -- python3 $R/bench.py --workload readme --steps 5 --warmup 1 --no-cpu-baseline --no-latency > $O/stats_readme_phased.log 2>&1
RB_MERGE=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_readme360_phased -- python3 $R/bench.py --workload readme --read-len 360 --steps 5 --warmup 1 --no-cpu-baseline --no-latency > $O/stats_readme360_phased.log 2>&1
for w in readme_phased readme360_phased; do f=$(find $O/stats_$w -name "*kernel_stats.csv" | head -1); echo "== $w"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ibf_count" in r["Name"]:
        print("  ", r["Name"].split("(")[0][-60:], r["Calls"], "avg ms %.4f" % (float(r["AverageNs"])/1e6))
PY
done

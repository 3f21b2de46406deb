#!/bin/bash
# Round 4, session 5: planner guard (final rules), CLI with its final defaults + page-cache write floor, full parity suite,
# the driver's command on the final tree, rocprofv3 stats + PMC passes of the headline (c3) and c3np2 for profiles/r04
TAG=${1:-r04s5}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
T="timeout 1800"
( time $T python3 -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.txt 2>&1
tail -3 $OUT/pytest_gpu.txt
( time $T python3 -m pytest tests -m gpuperf -q ) > $OUT/pytest_gpuperf.txt 2>&1
tail -3 $OUT/pytest_gpuperf.txt
( time $T python3 profiles/phase_rule_check.py ) > $OUT/phase_rule_check.txt 2>&1
echo "phase_rule_check rc=$?" >> $OUT/phase_rule_check.txt
grep -E "rule vs best|outside|rc=" $OUT/phase_rule_check.txt | cut -c1-300
( time $T python3 profiles/cli_readme250.py ) > $OUT/cli_throughput.txt 2>&1
cut -c1-420 $OUT/cli_throughput.txt
( time $T python3 bench.py --steps 20 --warmup 5 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -3 $OUT/bench_default.err
$T python3 bench.py --workload grch38_f100k --reads 2000000 --steps 3 --warmup 1 --cpu-seconds 8 --no-latency > $OUT/bench_grch38_f100k.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3 -- python3 $R/bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $OUT/stats_c3.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3np2 -- python3 $R/bench.py --workload c3np2 --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $OUT/stats_c3np2.log 2>&1
bash $R/profiles/collect_pmc.sh c3 10000000 $OUT/pmc_c3 > /dev/null 2>&1
for w in c3 c3np2; do f=$(find $OUT/stats_$w -name "*kernel_stats.csv" | head -1); echo "== $w"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ibf_count" in r["Name"] or "decide" in r["Name"] or "probe" in r["Name"]:
        print("  ", r["Name"].split("(")[0][-60:], r["Calls"], "avg ms %.4f" % (float(r["AverageNs"])/1e6))
PY
done
python3 - "$OUT/bench_default.json" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r=d["roofline"]; print("c3", round(d["value"]), round(r["frac"],4), r.get("frac_of_measured_read_peak"), d["parity"])
for k,v in d["other_configs"].items():
    rr=v.get("roofline") or {}
    print("  ", k, round(v.get("value",0)), rr.get("frac") and round(rr["frac"],4), rr.get("frac_of_measured_read_peak"), v.get("parity"), v.get("error"))
PY

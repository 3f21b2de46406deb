#!/bin/bash
# Round 4, session 1 (on the GPU box): parity suite, the default bench line with c3np2, N engines on one GPU, c3np2 + grch38_f100k
# rocprof stats and PMC passes.  bash profiles/collect_r04_s1.sh <tag>
TAG=${1:-r04s1}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
T="timeout 1500"
( time $T python3 -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
( time $T python3 bench.py --steps 10 --warmup 2 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -3 $OUT/bench_default.err
( time $T python3 profiles/engines_on_one_gpu.py ) > $OUT/engines_on_one_gpu.txt 2>&1
cat $OUT/engines_on_one_gpu.txt | tail -20
( time $T python3 -m pytest tests -m gpuperf -q ) > $OUT/pytest_gpuperf.txt 2>&1
tail -5 $OUT/pytest_gpuperf.txt
$T python3 bench.py --workload grch38_f100k --reads 2000000 --steps 3 --warmup 1 --cpu-seconds 8 --no-latency > $OUT/bench_grch38_f100k.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3np2 -- python3 $R/bench.py --workload c3np2 --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $OUT/stats_c3np2.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_grch38_f100k -- python3 $R/bench.py --workload grch38_f100k --reads 2000000 --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $OUT/stats_grch38_f100k.log 2>&1
bash $R/profiles/collect_pmc.sh c3np2 10000000 $OUT/pmc_c3np2 > /dev/null 2>&1
bash $R/profiles/collect_pmc.sh grch38_f100k 2000000 $OUT/pmc_grch38_f100k > /dev/null 2>&1
for w in c3np2 grch38_f100k; do f=$(find $OUT/stats_$w -name "*kernel_stats.csv" | head -1); echo "== $w"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ibf_count" in r["Name"] or "decide" in r["Name"]:
        print("  ", r["Name"].split("(")[0][-60:], r["Calls"], "avg ms %.4f" % (float(r["AverageNs"])/1e6))
PY
done
for f in $OUT/bench_*.json; do python3 - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], round(d["value"]), r.get("frac") and round(r["frac"],4), (r.get("read_peak_probe") or {}).get("GBps"), r.get("frac_of_measured_read_peak"), (d.get("cpu_baseline") or {}).get("value"), d.get("parity"))
    for k,v in (d.get("other_configs") or {}).items():
        rr=v.get("roofline") or {}
        print("    ", k, round(v.get("value",0)), rr.get("frac") and round(rr["frac"],4), (rr.get("read_peak_probe") or {}).get("GBps"), v.get("parity"), v.get("error"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
for w in c3np2 grch38_f100k; do echo "== pmc $w"; cat $OUT/pmc_$w/*.summary.csv | cut -c1-400; done

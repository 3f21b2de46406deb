#!/bin/bash
# Round-3 evidence in one go (on the GPU box): bash profiles/r03/collect_r03.sh <tag> ; then here:
#   RB_EVIDENCE_DATE=<date> python3 profiles/summarize.py gpurun_out/<tag> profiles/r03
TAG=${1:-r03e}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
T="timeout 1200"
( time python3 -m pytest tests -m gpu -q ) > $OUT/pytest_gpu_full.txt 2>&1
tail -4 $OUT/pytest_gpu_full.txt
# the driver's line: config 3 at 10 M reads per launch + other_configs (c4, c5 with the live-step leg, c2, readme, readme at 360 bp)
( time $T python3 bench.py --steps 20 --warmup 5 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -4 $OUT/bench_default.err
$T python3 bench.py --workload c1 --steps 5 --warmup 1 --cpu-seconds 5 --no-latency > $OUT/bench_c1.json 2> /dev/null
$T python3 bench.py --workload grch38_f100k --steps 3 --warmup 1 --cpu-seconds 8 > $OUT/bench_grch38_f100k.json 2> /dev/null
$T python3 bench.py --workload c5 --rate 18750 > $OUT/bench_c5_18750.json 2> /dev/null
# the default command with two ranks on the one GPU of this box (test hooks; RCCL refuses duplicate devices: gloo), batches / 20
RB_BENCH_BACKEND=gloo RB_BENCH_SAME_GPU=1 RB_BENCH_DUMP_DECISIONS=1 RB_BENCH_READS_DIVISOR=20 $T python3 bench.py --gpus 2 --steps 3 --warmup 1 --cpu-seconds 3 > $OUT/bench_gpus2_same_gpu_default.json 2> $OUT/bench_gpus2_same_gpu_default.err
RB_BENCH_DUMP_DECISIONS=1 RB_BENCH_READS_DIVISOR=20 $T python3 bench.py --gpus 1 --steps 3 --warmup 1 --cpu-seconds 3 > $OUT/bench_gpus1_default_div20.json 2> /dev/null
cd /tmp && export TMPDIR=/tmp
# rocprofv3 --kernel-trace --stats of the headline leg on its own (the default command runs the same kernel template on c4's
# 2 M-read launches and on micro-batches as well, which would mix into one average): 10 M reads per launch
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3 -- python3 $R/bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $OUT/stats_c3.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c4 -- python3 $R/bench.py --workload c4 --steps 5 --warmup 1 --no-cpu-baseline --no-latency > $OUT/stats_c4.log 2>&1
for w in c2 readme c1; do
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$w -- python3 $R/bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-latency > $OUT/stats_$w.log 2>&1
done
RB_TUNING_ENV=1 RB_MERGE=0 $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_readme_phased -- python3 $R/bench.py --workload readme --steps 5 --warmup 1 --no-cpu-baseline --no-latency > $OUT/stats_readme_phased.log 2>&1
RB_TUNING_ENV=1 RB_MERGE=0 $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_readme360_phased -- python3 $R/bench.py --workload readme --read-len 360 --steps 5 --warmup 1 --no-cpu-baseline --no-latency > $OUT/stats_readme360_phased.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_readme360 -- python3 $R/bench.py --workload readme --read-len 360 --steps 5 --warmup 1 --no-cpu-baseline --no-latency > $OUT/stats_readme360.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -- python3 $R/bench.py --workload c5 --replay-seconds 1.0 > $OUT/stats_c5.log 2>&1
# PMC passes (separate runs, counters only): c3 on the 10 M-read launch
bash $R/profiles/collect_pmc.sh c3 10000000 $OUT/pmc_c3 > /dev/null 2>&1
for w in c2 c4 readme c1; do bash $R/profiles/collect_pmc.sh $w 1000000 $OUT/pmc_$w > /dev/null 2>&1; done
bash $R/profiles/collect_pmc.sh readme 1000000 $OUT/pmc_readme360 "--read-len 360" > /dev/null 2>&1
RB_TUNING_ENV=1 RB_MERGE=0 bash $R/profiles/collect_pmc.sh readme 1000000 $OUT/pmc_readme_phased > /dev/null 2>&1
RB_TUNING_ENV=1 RB_MERGE=0 bash $R/profiles/collect_pmc.sh readme 1000000 $OUT/pmc_readme360_phased "--read-len 360" > /dev/null 2>&1
for w in c2 c3 c4 c5 readme readme360 readme_phased readme360_phased c1; do f=$(find $OUT/stats_$w -name "*kernel_stats.csv" | head -1); echo "== $w"; python3 - "$f" <<'PY'
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
    print(sys.argv[1].split("/")[-1], "n_gpus", d.get("n_gpus"), round(d["value"]), r.get("achieved") and round(r["achieved"]), r.get("frac") and round(r["frac"],3), (d.get("cpu_baseline") or {}).get("value"), d.get("parity"), (d.get("config") or {}).get("decisions_sha1"), {k:v for k,v in (d.get("latency") or {}).items() if k.startswith("p")})
    for k,v in (d.get("other_configs") or {}).items():
        rr=v.get("roofline") or {}
        print("    ", k, round(v.get("value",0)), rr.get("frac") and round(rr["frac"],3), v.get("parity"), (v.get("latency") or {}).get("p99_ms"), (v.get("live_step") or {}).get("p99_ms"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done

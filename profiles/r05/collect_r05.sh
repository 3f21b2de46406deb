#!/bin/bash
# Round-5 evidence in one go (on the GPU box): bash profiles/r05/collect_r05.sh <tag> ; then here:
#   RB_EVIDENCE_DATE=<date> python3 profiles/summarize.py gpurun_out/<tag> profiles/r05
# bench.py prints ONE bounded line since this round; the full result of every run is its RB_BENCH_DETAIL sidecar (kept as
# bench_<name>.json beside the line, bench_<name>_line.json).
TAG=${1:-r05f}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
T="timeout 1800"
( time $T python3 -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.txt 2>&1
tail -n 4 $OUT/pytest_gpu.txt | cut -c1-200
( time $T python3 -m pytest tests -m gpuperf -q ) > $OUT/pytest_gpuperf.txt 2>&1
tail -n 4 $OUT/pytest_gpuperf.txt | cut -c1-200
bench() { # name args...
  local name=$1; shift
  ( time RB_BENCH_DETAIL=$OUT/bench_$name.json $T python3 bench.py "$@" ) > $OUT/bench_${name}_line.json 2> $OUT/bench_$name.err
  echo "bench $name: rc=$? line $(wc -c < $OUT/bench_${name}_line.json) bytes; $(tail -n 3 $OUT/bench_$name.err | tr '\n' ' ')"
}
# the driver's command
bench default --gpus 1 --steps 20 --warmup 5
bench no_flags
bench c1 --workload c1 --steps 5 --warmup 1 --cpu-seconds 5 --no-latency
bench w1_64mib --workload w1_64mib --steps 5 --warmup 2 --cpu-seconds 5 --no-latency
bench pool --pool --steps 3
RB_BENCH_POOL_DEVICES=0,0 bench pool_two_workers_one_gpu --pool --steps 3
cd /tmp && export TMPDIR=/tmp
export RB_BENCH_NO_SUPERVISOR=1  # under the profiler the measuring process is the profiled one
for w in c3 c3np2 c4 c2 readme targets3 deplete_target c1 w1_64mib; do
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$w -- python3 $R/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $OUT/stats_$w.log 2>&1
done
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_readme360 -- python3 $R/bench.py --workload readme --read-len 360 --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $OUT/stats_readme360.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_grch38_f100k -- python3 $R/bench.py --workload grch38_f100k --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $OUT/stats_grch38_f100k.log 2>&1
# the driver's own command under the profiler (headline + every leg: the kernel table of the run that BENCH records)
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/stats_default.log 2>&1
unset RB_BENCH_NO_SUPERVISOR
bash $R/profiles/collect_pmc.sh c3 10000000 $OUT/pmc_c3 > /dev/null 2>&1
bash $R/profiles/collect_pmc.sh c3np2 10000000 $OUT/pmc_c3np2 > /dev/null 2>&1
bash $R/profiles/collect_pmc.sh grch38_f100k 2000000 $OUT/pmc_grch38_f100k > /dev/null 2>&1
for w in c2 c4 readme targets3 deplete_target c1 w1_64mib; do bash $R/profiles/collect_pmc.sh $w 1000000 $OUT/pmc_$w > /dev/null 2>&1; done
bash $R/profiles/collect_pmc.sh readme 1000000 $OUT/pmc_readme360 "--read-len 360" > /dev/null 2>&1
for w in c3 c3np2 c4 c2 readme readme360 targets3 deplete_target c1 w1_64mib grch38_f100k default; do f=$(find $OUT/stats_$w -name "*kernel_stats.csv" | head -1); echo "== $w"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ibf_count" in r["Name"]:
        print("  ", r["Name"].split("(")[0][-70:], r["Calls"], "avg ms %.4f" % (float(r["AverageNs"])/1e6))
PY
done
# gpurun brings back at most 64 MiB: the per-dispatch traces are not evidence (the stats tables are), and the default run's c5 leg alone
# launches 150 k kernels
find $OUT -name "*.db" -delete; find $OUT -path "*stats_*" -name "*kernel_trace.csv" -delete; find $OUT -path "*stats_*" -name "*agent_info.csv" -delete
find $OUT -path "*pmc_*" -name "*.csv" -size +2000k -delete; find $OUT -path "*pmc_*" -name "*kernel_trace.csv" -delete
du -sm $OUT | cut -f1 | xargs echo "MiB under $OUT:"
cd $R
for f in $OUT/bench_*.json; do case $f in *_line.json) continue;; esac; python3 - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    r=d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], "n_gpus", d.get("n_gpus"), round(d["value"]), r.get("frac") and round(r["frac"],4), r.get("frac_of_measured_read_peak"), (d.get("cpu_baseline") or {}).get("value"), d.get("parity"), "bench_seconds", d.get("bench_seconds"))
    for k,v in (d.get("other_configs") or {}).items():
        rr=v.get("roofline") or {}
        print("    ", k, round(v.get("value",0)), rr.get("frac") and round(rr["frac"],4), rr.get("frac_of_measured_read_peak"), (rr.get("request_bound") or {}).get("request_bound_frac"), v.get("parity"), (v.get("latency") or {}).get("p99_ms"), (v.get("live_step") or {}).get("p99_ms"), v.get("leg_seconds"), v.get("error"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done

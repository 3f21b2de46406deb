#!/bin/bash
# Round-6 evidence on the final tree: the GPU suites, smoke, the driver's command (and the run without flags), rocprofv3 kernel tables of the
# headline and of the narrow shapes whose kernels changed this round, the four-pass counter collection of those shapes and of the early-decision
# leg (-> traffic.json through summarize.py, BEFORE the bench lines that replay it), and the placement-off twins of the two reference-sized legs.  Usage: bash profiles/r06/collect_r06.sh [tag]
TAG=${1:-r06z}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
T="timeout 2400"
( time $T python3 -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.txt 2>&1
tail -n 4 $OUT/pytest_gpu.txt | cut -c1-200
( time $T python3 -m pytest tests -m gpuperf -q ) > $OUT/pytest_gpuperf.txt 2>&1
tail -n 4 $OUT/pytest_gpuperf.txt | cut -c1-200
( time $T python3 -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.txt 2>&1
tail -n 2 $OUT/smoke.txt
cd $R
for w in deplete_target targets3 readme c1 c3_early; do bash profiles/collect_pmc.sh $w 1000000 $OUT/pmc_$w > $OUT/pmc_$w.log 2>&1; done
bash profiles/collect_pmc.sh readme 1000000 $OUT/pmc_readme360 "--read-len 360" > $OUT/pmc_readme360.log 2>&1
find $OUT -name "*.db" -delete; find $OUT -path "*pmc_*" -name "*kernel_trace.csv" -delete; find $OUT -path "*pmc_*" -name "*agent_info.csv" -delete
# the counters go into profiles/traffic.json BEFORE the bench runs below replay them (roofline.traffic, request_bound): the line and the
# counters then belong to the same kernels and the same slicing (the builder re-runs summarize.py on the merged output at home)
cd $R
mkdir -p $OUT/r06
RB_EVIDENCE_DATE=$(date +%F) python3 profiles/summarize.py $OUT $OUT/r06 > $OUT/summarize.txt 2>&1; tail -n 8 $OUT/summarize.txt
cp profiles/traffic.json $OUT/traffic_after_pmc.json
bench() { # name args...
  local name=$1; shift
  ( time RB_BENCH_DETAIL=$OUT/bench_$name.json $T python3 bench.py "$@" ) > $OUT/bench_${name}_line.json 2> $OUT/bench_$name.err
  echo "bench $name: rc=$? line $(wc -c < $OUT/bench_${name}_line.json) bytes; $(tail -n 3 $OUT/bench_$name.err | tr '\n' ' ')"
}
bench default --gpus 1 --steps 20 --warmup 5
bench no_flags
bench c3_early --workload c3_early --no-cpu-baseline
# the two reference-sized tables with the placement trial off (what the trial buys on this ROCm)
RB_BENCH_PLACEMENT=off bench placement_off_c3np2 --workload c3np2 --steps 5 --warmup 1 --no-cpu-baseline --no-latency
RB_BENCH_PLACEMENT=off bench placement_off_grch38_f100k --workload grch38_f100k --steps 3 --warmup 1 --no-cpu-baseline --no-latency
bench placement_on_c3np2 --workload c3np2 --steps 5 --warmup 1 --no-cpu-baseline --no-latency
bench placement_on_grch38_f100k --workload grch38_f100k --steps 3 --warmup 1 --no-cpu-baseline --no-latency
cd /tmp && export TMPDIR=/tmp
export RB_BENCH_NO_SUPERVISOR=1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c3 -- python3 $R/bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $OUT/stats_c3.log 2>&1
for w in deplete_target targets3 readme c1; do
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$w -- python3 $R/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-latency > $OUT/stats_$w.log 2>&1
done
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_readme360 -- python3 $R/bench.py --workload readme --read-len 360 --steps 10 --warmup 2 --no-cpu-baseline --no-latency > $OUT/stats_readme360.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/stats_default.log 2>&1
unset RB_BENCH_NO_SUPERVISOR
find $OUT -name "*.db" -delete; find $OUT -path "*stats_*" -name "*kernel_trace.csv" -delete; find $OUT -path "*stats_*" -name "*agent_info.csv" -delete
for w in c3 deplete_target targets3 readme readme360 c1 default; do f=$(find $OUT/stats_$w -name "*kernel_stats.csv" | head -1); echo "== $w"; [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ibf_count" in r["Name"] or "probe" in r["Name"]:
        print("  ", r["Name"].split("(")[0][-70:], r["Calls"], "avg ms %.4f" % (float(r["AverageNs"])/1e6))
PY
done
for f in $OUT/bench_default.json $OUT/bench_no_flags.json $OUT/bench_placement_*.json; do python3 - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d.get("roofline") or {}
print(sys.argv[1].split("/")[-1], round(d["value"]), r.get("frac") and round(r["frac"],4), r.get("frac_of_measured_read_peak"), r.get("placement"), (d.get("cpu_baseline") or {}).get("value"), "bench_seconds", d.get("bench_seconds"))
for k,v in (d.get("other_configs") or {}).items():
    rr=v.get("roofline") or {}
    print("    ", k, round(v.get("value",0)), rr.get("frac") and round(rr["frac"],4), rr.get("frac_of_measured_read_peak"), (rr.get("request_bound") or {}).get("request_bound_frac"), (v.get("latency") or {}).get("p99_ms"), (v.get("live_step") or {}).get("p99_ms"), v.get("leg_seconds"), v.get("error"), (v.get("parity") or {}).get("checked_reads"))
PY
done
du -sm $OUT | cut -f1 | xargs echo "MiB under $OUT:"

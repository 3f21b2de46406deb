#!/bin/bash
# Round-5 final tree (after placement by trial went in): the GPU suites, the driver's command, and the kernel tables of the headline and of
# the two reference-sized filters again.  Everything else of profiles/r05 is from collect_r05.sh on the tree before (same kernels).
TAG=${1:-r05g}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
T="timeout 1800"
( time $T python3 -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.txt 2>&1
tail -n 4 $OUT/pytest_gpu.txt | cut -c1-200
( time $T python3 -m pytest tests -m gpuperf -q ) > $OUT/pytest_gpuperf.txt 2>&1
tail -n 4 $OUT/pytest_gpuperf.txt | cut -c1-200
( time $T python3 -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.txt 2>&1
tail -n 2 $OUT/smoke.txt
bench() { # name args...
  local name=$1; shift
  ( time RB_BENCH_DETAIL=$OUT/bench_$name.json $T python3 bench.py "$@" ) > $OUT/bench_${name}_line.json 2> $OUT/bench_$name.err
  echo "bench $name: rc=$? line $(wc -c < $OUT/bench_${name}_line.json) bytes; $(tail -n 3 $OUT/bench_$name.err | tr '\n' ' ')"
}
bench default --gpus 1 --steps 20 --warmup 5
bench no_flags
cd /tmp && export TMPDIR=/tmp
export RB_BENCH_NO_SUPERVISOR=1
for w in c3 c3np2; do
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$w -- python3 $R/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $OUT/stats_$w.log 2>&1
done
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_grch38_f100k -- python3 $R/bench.py --workload grch38_f100k --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $OUT/stats_grch38_f100k.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_default -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/stats_default.log 2>&1
unset RB_BENCH_NO_SUPERVISOR
find $OUT -name "*.db" -delete; find $OUT -path "*stats_*" -name "*kernel_trace.csv" -delete; find $OUT -path "*stats_*" -name "*agent_info.csv" -delete
for w in c3 c3np2 grch38_f100k default; do f=$(find $OUT/stats_$w -name "*kernel_stats.csv" | head -1); echo "== $w"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ibf_count" in r["Name"] or "probe" in r["Name"]:
        print("  ", r["Name"].split("(")[0][-70:], r["Calls"], "avg ms %.4f" % (float(r["AverageNs"])/1e6))
PY
done
cd $R
for f in $OUT/bench_default.json $OUT/bench_no_flags.json; do python3 - "$f" <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d.get("roofline") or {}
print(sys.argv[1].split("/")[-1], round(d["value"]), round(r["frac"],4), r.get("frac_of_measured_read_peak"), r.get("placement"), (d.get("cpu_baseline") or {}).get("value"), "bench_seconds", d.get("bench_seconds"))
for k,v in (d.get("other_configs") or {}).items():
    rr=v.get("roofline") or {}
    print("    ", k, round(v.get("value",0)), rr.get("frac") and round(rr["frac"],4), rr.get("frac_of_measured_read_peak"), (rr.get("request_bound") or {}).get("request_bound_frac"), rr.get("placement"), (v.get("latency") or {}).get("p99_ms"), (v.get("live_step") or {}).get("p99_ms"), v.get("leg_seconds"), v.get("error"))
PY
done
du -sm $OUT | cut -f1 | xargs echo "MiB under $OUT:"

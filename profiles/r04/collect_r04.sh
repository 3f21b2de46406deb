#!/bin/bash
# Round-4 evidence in one go (on the GPU box): bash profiles/r04/collect_r04.sh <tag> ; then here:
#   RB_EVIDENCE_DATE=<date> python3 profiles/summarize.py gpurun_out/<tag> profiles/r04
# (the round's working sessions used profiles/collect_r04_s<N>.sh; this is their union on the final tree)
TAG=${1:-r04f}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
T="timeout 1800"
( time $T python3 -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.txt 2>&1
tail -n 4 $OUT/pytest_gpu.txt | cut -c1-200
( time $T python3 -m pytest tests -m gpuperf -q ) > $OUT/pytest_gpuperf.txt 2>&1
tail -n 4 $OUT/pytest_gpuperf.txt | cut -c1-200
( time $T python3 profiles/phase_rule_check.py ) > $OUT/phase_rule_check.txt 2>&1
echo "phase_rule_check rc=$?" >> $OUT/phase_rule_check.txt
tail -n 2 $OUT/phase_rule_check.txt
# the driver's line
( time $T python3 bench.py --steps 20 --warmup 5 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -3 $OUT/bench_default.err
$T python3 bench.py --workload grch38_f100k --reads 2000000 --steps 3 --warmup 1 --cpu-seconds 8 --no-latency > $OUT/bench_grch38_f100k.json 2> /dev/null
$T python3 bench.py --workload c1 --steps 5 --warmup 1 --cpu-seconds 5 --no-latency > $OUT/bench_c1.json 2> /dev/null
( time $T python3 bench.py --pool --steps 3 ) > $OUT/bench_pool.json 2> $OUT/bench_pool.err
( time RB_BENCH_POOL_DEVICES=0,0 $T python3 bench.py --pool --steps 3 ) > $OUT/bench_pool_two_workers_one_gpu.json 2>> $OUT/bench_pool.err
# the default command with two ranks on the one GPU of this box (test hooks; RCCL refuses duplicate devices: gloo), batches / 20
RB_BENCH_BACKEND=gloo RB_BENCH_SAME_GPU=1 RB_BENCH_DUMP_DECISIONS=1 RB_BENCH_READS_DIVISOR=20 $T python3 bench.py --gpus 2 --steps 3 --warmup 1 --cpu-seconds 3 > $OUT/bench_gpus2_same_gpu_default.json 2> $OUT/bench_gpus2_same_gpu_default.err
RB_BENCH_DUMP_DECISIONS=1 RB_BENCH_READS_DIVISOR=20 $T python3 bench.py --gpus 1 --steps 3 --warmup 1 --cpu-seconds 3 > $OUT/bench_gpus1_default_div20.json 2> /dev/null
( time $T python3 profiles/engines_on_one_gpu.py ) > $OUT/engines_on_one_gpu.txt 2>&1
$T python3 profiles/calibrate_gain.py > $OUT/calibrate_gain.txt 2>&1
( time $T python3 profiles/cli_readme250.py ) > $OUT/cli_throughput.txt 2>&1
( time $T python3 profiles/cli_readme250.py 64000000 - quick ) > $OUT/cli_throughput_64M_reads.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for w in c3 c3np2 c4 c2 readme targets3 deplete_target c1; do
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$w -- python3 $R/bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $OUT/stats_$w.log 2>&1
done
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_readme360 -- python3 $R/bench.py --workload readme --read-len 360 --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $OUT/stats_readme360.log 2>&1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_grch38_f100k -- python3 $R/bench.py --workload grch38_f100k --reads 2000000 --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $OUT/stats_grch38_f100k.log 2>&1
bash $R/profiles/collect_pmc.sh c3 10000000 $OUT/pmc_c3 > /dev/null 2>&1
bash $R/profiles/collect_pmc.sh c3np2 10000000 $OUT/pmc_c3np2 > /dev/null 2>&1
bash $R/profiles/collect_pmc.sh grch38_f100k 2000000 $OUT/pmc_grch38_f100k > /dev/null 2>&1
for w in c2 c4 readme targets3 deplete_target c1; do bash $R/profiles/collect_pmc.sh $w 1000000 $OUT/pmc_$w > /dev/null 2>&1; done
bash $R/profiles/collect_pmc.sh readme 1000000 $OUT/pmc_readme360 "--read-len 360" > /dev/null 2>&1
for w in c3 c3np2 c4 c2 readme readme360 targets3 deplete_target c1 grch38_f100k; do f=$(find $OUT/stats_$w -name "*kernel_stats.csv" | head -1); echo "== $w"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ibf_count" in r["Name"]:
        print("  ", r["Name"].split("(")[0][-70:], r["Calls"], "avg ms %.4f" % (float(r["AverageNs"])/1e6))
PY
done
for f in $OUT/bench_*.json; do python3 - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], "n_gpus", d.get("n_gpus"), round(d["value"]), r.get("frac") and round(r["frac"],4), r.get("frac_of_measured_read_peak"), (d.get("cpu_baseline") or {}).get("value"), d.get("parity"), (d.get("config") or {}).get("decisions_sha1"))
    for k,v in (d.get("other_configs") or {}).items():
        rr=v.get("roofline") or {}
        print("    ", k, round(v.get("value",0)), rr.get("frac") and round(rr["frac"],4), rr.get("frac_of_measured_read_peak"), v.get("parity"), (v.get("latency") or {}).get("p99_ms"), (v.get("live_step") or {}).get("p99_ms"), v.get("error"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done

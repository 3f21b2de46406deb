#!/bin/bash
# Round-2 evidence in one go (on the GPU box): bash profiles/r02/collect_r02.sh <tag>
TAG=${1:-r02e}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
T="timeout 900"
$T python3 bench.py --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
$T python3 bench.py --workload c3 --steps 10 --warmup 2 --cpu-seconds 10 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
$T python3 bench.py --workload c4 --reads 2000000 --steps 5 --warmup 1 --cpu-seconds 10 > $OUT/bench_c4.json 2> $OUT/bench_c4.err
$T python3 bench.py --workload readme --steps 10 --warmup 2 --cpu-seconds 8 > $OUT/bench_readme.json 2> $OUT/bench_readme.err
$T python3 bench.py --workload readme --steps 10 --warmup 2 --no-cpu-baseline --no-latency --phased off --no-overlap > $OUT/bench_readme_plain_serial.json 2> /dev/null
$T python3 bench.py --workload c1 --steps 5 --warmup 1 --cpu-seconds 5 --no-latency > $OUT/bench_c1.json 2> /dev/null
$T python3 bench.py --workload c1 --steps 5 --warmup 1 --no-cpu-baseline --no-latency --phased off > $OUT/bench_c1_plain.json 2> /dev/null
$T python3 bench.py --workload grch38_f100k --steps 3 --warmup 1 --cpu-seconds 8 > $OUT/bench_grch38_f100k.json 2> /dev/null
$T python3 bench.py --workload c5 > $OUT/bench_c5_150k.json 2> /dev/null
$T python3 bench.py --workload c5 --rate 18750 > $OUT/bench_c5_18750.json 2> /dev/null
# two ranks on the one GPU of this box (RCCL refuses duplicate devices: collectives through gloo), launched by bench.py itself
RB_BENCH_BACKEND=gloo RB_BENCH_SAME_GPU=1 RB_BENCH_DUMP_DECISIONS=1 $T python3 bench.py --gpus 2 --workload c4 --reads 200000 --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $OUT/gpus2_same_gpu_c4.json 2> $OUT/gpus2_same_gpu_c4.err
RB_BENCH_BACKEND=gloo RB_BENCH_SAME_GPU=1 RB_BENCH_DUMP_DECISIONS=1 $T python3 bench.py --gpus 2 --bin-sharded --workload c4 --reads 200000 --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $OUT/gpus2_same_gpu_c4_binsharded.json 2> $OUT/gpus2_same_gpu_c4_binsharded.err
RB_BENCH_DUMP_DECISIONS=1 $T python3 bench.py --gpus 1 --workload c4 --reads 200000 --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $OUT/gpus1_c4.json 2> /dev/null
$T python3 profiles/latency_floor.py 2>/dev/null > $OUT/latency_floor.txt
$T python3 profiles/latency_wide.py 2>/dev/null > $OUT/latency_wide.txt
$T python3 profiles/pool_replication.py 4 8 > $OUT/pool_replication.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for w in c2 c3 c4 readme c1; do
  N=1000000; [ $w = c3 -o $w = c4 ] && N=2000000
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$w -- python3 $R/bench.py --workload $w --reads $N --steps 5 --warmup 1 --no-cpu-baseline --no-latency > $OUT/stats_$w.log 2>&1
done
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -- python3 $R/bench.py --workload c5 --replay-seconds 1.0 > $OUT/stats_c5.log 2>&1
for w in c2 c3 c4 readme c1; do bash $R/profiles/collect_pmc.sh $w 1000000 $OUT/pmc_$w > /dev/null 2>&1; done
bash $R/profiles/collect_pmc.sh grch38_f100k 500000 $OUT/pmc_grch38_f100k > /dev/null 2>&1
hipcc -O3 --offload-arch=gfx950 $R/profiles/hbm_peak.hip -o /tmp/hbm_peak 2>/dev/null && $T /tmp/hbm_peak > $OUT/hbm_peak.txt 2>&1; cat $OUT/hbm_peak.txt
for w in c2 c3 c4 c5 readme c1; do f=$(find $OUT/stats_$w -name "*kernel_stats.csv" | head -1); echo "== $w"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ibf_count" in r["Name"] or "decide" in r["Name"]:
        print("  ", r["Name"].split("(")[0][-60:], r["Calls"], "avg ms %.4f" % (float(r["AverageNs"])/1e6))
PY
done
for f in $OUT/bench_*.json $OUT/gpus*.json; do python3 - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    r=d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], "n_gpus", d.get("n_gpus"), round(d["value"]), r.get("achieved") and round(r["achieved"]), r.get("frac") and round(r["frac"],3), (d.get("cpu_baseline") or {}).get("value"), d.get("parity"), (d.get("config") or {}).get("decisions_sha1"), {k:v for k,v in (d.get("latency") or {}).items() if k.startswith("p")})
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
cat $OUT/pool_replication.txt

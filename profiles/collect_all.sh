#!/bin/bash
# Round evidence in one go: bench lines of every config, rocprofv3 --kernel-trace --stats of the same commands,
# and the separate --pmc passes for config 2 and 3.  Usage (on the GPU box): bash profiles/collect_all.sh <tag>
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python3 bench.py > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python3 bench.py --workload c3 --reads 2000000 --steps 5 --warmup 1 --cpu-seconds 10 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
python3 bench.py --workload c3np2 --reads 2000000 --steps 3 --warmup 1 --no-cpu-baseline --no-latency > $OUT/bench_c3np2.json 2> /dev/null
python3 bench.py --workload c4 --reads 2000000 --steps 5 --warmup 1 --cpu-seconds 10 > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python3 bench.py --workload grch38_f100k --steps 3 --warmup 1 --cpu-seconds 8 > $OUT/bench_grch38_f100k.json 2> /dev/null
python3 bench.py --workload c1 --steps 5 --warmup 1 --no-cpu-baseline --no-latency > $OUT/bench_c1.json 2> /dev/null
python3 bench.py --workload c5 > $OUT/bench_c5_150k.json 2> /dev/null
python3 bench.py --workload c5 --rate 18750 > $OUT/bench_c5_18750.json 2> /dev/null
python3 profiles/latency_floor.py 2>/dev/null > $OUT/latency_floor.txt
python3 profiles/latency_wide.py 2>/dev/null > $OUT/latency_wide.txt
cd /tmp && export TMPDIR=/tmp
for w in c2 c3 c4; do
  N=1000000; [ $w != c2 ] && N=2000000
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$w -- python3 $R/bench.py --workload $w --reads $N --steps 5 --warmup 1 --no-cpu-baseline --no-latency > $OUT/stats_$w.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c5 -- python3 $R/bench.py --workload c5 --replay-seconds 1.0 > $OUT/stats_c5.log 2>&1
bash $R/profiles/collect_pmc.sh c2 1000000 $OUT/pmc_c2 > /dev/null 2>&1
bash $R/profiles/collect_pmc.sh c3 1000000 $OUT/pmc_c3 > /dev/null 2>&1
bash $R/profiles/collect_pmc.sh c4 1000000 $OUT/pmc_c4 > /dev/null 2>&1
bash $R/profiles/collect_pmc.sh grch38_f100k 500000 $OUT/pmc_grch38_f100k > /dev/null 2>&1
hipcc -O3 --offload-arch=gfx950 $R/profiles/hbm_peak.hip -o /tmp/hbm_peak 2>/dev/null && /tmp/hbm_peak > $OUT/hbm_peak.txt 2>&1; cat $OUT/hbm_peak.txt
for w in c2 c3 c4 c5; do f=$(find $OUT/stats_$w -name "*kernel_stats.csv" | head -1); echo "== $w"; grep -E "rb::ibf_count" $f | sed -e 's/(rb::IbfDev[^"]*"/"/' | head -4; done
for f in $OUT/bench_*.json; do python3 - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    r=d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], round(d["value"]), r.get("achieved") and round(r["achieved"]), r.get("frac") and round(r["frac"],3), (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cores"), d.get("parity"), {k:v for k,v in (d.get("latency") or {}).items() if k.startswith("p")})
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done

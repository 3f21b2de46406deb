#!/bin/bash
# Round 4, session 6: bit-packed merged tables -- parity suite, the narrow-filter bench legs, CLI throughput with the page-cache write floor
TAG=${1:-r04s6}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
T="timeout 1800"
( time $T python3 -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.txt 2>&1
tail -n 8 $OUT/pytest_gpu.txt | cut -c1-300
for w in readme targets3 deplete_target; do
  $T python3 bench.py --workload $w --steps 10 --warmup 2 --cpu-seconds 5 --no-latency > $OUT/bench_$w.json 2> $OUT/bench_$w.err
done
$T python3 bench.py --workload readme --read-len 360 --steps 10 --warmup 2 --no-cpu-baseline --no-latency > $OUT/bench_readme_360bp.json 2> /dev/null
RB_MERGE=0 $T python3 bench.py --workload readme --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $OUT/bench_readme_unmerged.json 2> /dev/null
for f in $OUT/bench_*.json; do python3 - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r=d["roofline"]; print(sys.argv[1].split("/")[-1], round(d["value"]), round(r["frac"],4), round(r["avg_kernel_ms"],3), r.get("plan")[:1], d.get("parity"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done
( time $T python3 profiles/cli_readme250.py ) > $OUT/cli_throughput.txt 2>&1
cut -c1-420 $OUT/cli_throughput.txt

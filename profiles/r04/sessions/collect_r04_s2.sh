#!/bin/bash
# Round 4, session 2 (on the GPU box): parity suite on the reworked CLI / pool / planner, the planner guard, CLI throughput on the
# README shape, L2 counters of N engines on one GPU, the one-process pool legs.  bash profiles/collect_r04_s2.sh <tag>
TAG=${1:-r04s2}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
T="timeout 1500"
( time $T python3 -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
( time $T python3 profiles/phase_rule_check.py ) > $OUT/phase_rule_check.txt 2>&1
echo "phase_rule_check rc=$?" >> $OUT/phase_rule_check.txt
grep -E "rule vs best|outside|rc=" $OUT/phase_rule_check.txt | cut -c1-260
( time $T python3 profiles/cli_readme250.py ) > $OUT/cli_throughput.txt 2>&1
cat $OUT/cli_throughput.txt | cut -c1-420
( time $T python3 bench.py --pool --steps 3 ) > $OUT/bench_pool.json 2> $OUT/bench_pool.err
( time RB_BENCH_POOL_DEVICES=0,0 $T python3 bench.py --pool --steps 3 ) > $OUT/bench_pool_two_workers_one_gpu.json 2>> $OUT/bench_pool.err
tail -3 $OUT/bench_pool.err
cd /tmp && export TMPDIR=/tmp
for K in 1 4; do
  $T rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/l2_unmerged_k$K -- python3 $R/profiles/engines_on_one_gpu.py --shapes readme_unmerged --k $K --forms device --batches 6 > $OUT/l2_unmerged_k$K.log 2>&1
  f=$(find $OUT/l2_unmerged_k$K -name "*counter_collection.csv" | head -1)
  python3 - "$f" $K <<'PY'
import csv,sys
tot={}
for r in csv.DictReader(open(sys.argv[1])):
    if "ibf_count_max" in r["Kernel_Name"]:
        tot[r["Counter_Name"]]=tot.get(r["Counter_Name"],0.0)+float(r["Counter_Value"])
h,m=tot.get("TCC_HIT_sum",0),tot.get("TCC_MISS_sum",0)
print("readme_unmerged K=%s: TCC_HIT %.4g TCC_MISS %.4g hit rate %.3f" % (sys.argv[2],h,m,h/max(h+m,1)))
PY
done
for f in $OUT/bench_pool*.json; do python3 - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    for k,v in [("pool_c3",d)]+list((d.get("other_configs") or {}).items()):
        print(sys.argv[1].split("/")[-1], k, round(v["value"]), v["pool"], v["parity"])
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done

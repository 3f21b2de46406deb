cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s14; mkdir -p $O
( timeout 900 python3 -m pytest tests/test_bench_ranks.py -m gpu -q -k "pool" ) > $O/pytest_pool.txt 2>&1; tail -n 4 $O/pytest_pool.txt | cut -c1-300
RB_BENCH_POOL_CHILD=1 RB_BENCH_READS_DIVISOR=20 timeout 900 python3 bench.py --steps 3 --warmup 1 > $O/bench_default_child_div20.json 2> $O/bench_default_child_div20.err
python3 - $O/bench_default_child_div20.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "preflight", d["ranks"]["xgmi_preflight"])
for k in ("pool_c3","pool_c4"):
    v=d["other_configs"][k]; print(k, v.get("value"), v.get("in_child_process"), v.get("child_seconds"), v.get("error"), (v.get("parity") or {}))
PY

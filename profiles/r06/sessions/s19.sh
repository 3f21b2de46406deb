#!/bin/bash
# r06 session 19: the opt-in early-decision mode on the GPU: its parity test, the whole -m gpu suite on the tree that
# holds it, the bench leg c3_early alone, its counters (four --pmc passes), and the default bench line with the leg in it
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
O=$R/gpurun_out/r06e
mkdir -p $O
( time timeout 900 python3 -m pytest tests -m gpu -x -q -k "early_decision" ) > $O/pytest_early.txt 2>&1
tail -5 $O/pytest_early.txt
( time timeout 1500 python3 -m pytest tests -m gpu -q ) > $O/pytest_gpu.txt 2>&1
tail -8 $O/pytest_gpu.txt
( time timeout 600 python3 bench.py --workload c3_early --no-cpu-baseline ) > $O/bench_c3_early_line.json 2> $O/bench_c3_early.err
cut -c1-1500 $O/bench_c3_early_line.json; tail -3 $O/bench_c3_early.err
cp profiles/bench_detail.json $O/bench_c3_early.json 2>/dev/null
bash profiles/collect_pmc.sh c3_early 1000000 $O/pmc_c3_early > $O/pmc_c3_early.log 2>&1
tail -5 $O/pmc_c3_early.log; cat $O/pmc_c3_early/*.summary.csv | cut -c1-300 | head -40
cd $R
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_default_line.json 2> $O/bench_default.err
wc -c $O/bench_default_line.json; tail -3 $O/bench_default.err
cp profiles/bench_detail.json $O/bench_default.json 2>/dev/null
python3 - <<'PY'
import json
d = json.load(open("/root/repo/gpurun_out/r06e/bench_default.json"))
for l in d.get("other_configs", d.get("legs", [])) if isinstance(d.get("other_configs", []), list) else []:
    print(l.get("leg") or l.get("config", {}).get("workload", "")[:30], l.get("value"), l.get("k1_ms"), l.get("speedup_over_full_count"))
PY
echo done

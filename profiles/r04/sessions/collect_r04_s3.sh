#!/bin/bash
# Round 4, session 3: full parity suite, planner guard after the rule fixes, CLI throughput with chunked positional writes, default bench line
TAG=${1:-r04s3}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
T="timeout 1500"
( time $T python3 -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.txt 2>&1
tail -5 $OUT/pytest_gpu.txt
( time $T python3 profiles/phase_rule_check.py ) > $OUT/phase_rule_check.txt 2>&1
echo "phase_rule_check rc=$?" >> $OUT/phase_rule_check.txt
grep -E "rule vs best|outside|rc=" $OUT/phase_rule_check.txt | cut -c1-260
( time $T python3 profiles/cli_readme250.py ) > $OUT/cli_throughput.txt 2>&1
cat $OUT/cli_throughput.txt | cut -c1-520
( time $T python3 bench.py --steps 10 --warmup 2 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -3 $OUT/bench_default.err
python3 - "$OUT/bench_default.json" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r=d["roofline"]; print("c3", round(d["value"]), round(r["frac"],4), r.get("frac_of_measured_read_peak"), r.get("plan"), d["parity"])
for k,v in d["other_configs"].items():
    rr=v.get("roofline") or {}
    print("  ", k, round(v.get("value",0)), rr.get("frac") and round(rr["frac"],4), rr.get("frac_of_measured_read_peak"), rr.get("plan"), v.get("parity"), v.get("pool"), v.get("error"))
PY

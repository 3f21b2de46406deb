#!/bin/bash
# r06 session 32: the driver's command once more on the final tree, now that every traffic.json entry it replays is this round's
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06x
mkdir -p $OUT
cd $R
( time RB_BENCH_DETAIL=$OUT/bench_default.json timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench_default_line.json 2> $OUT/bench_default.err
echo "rc=$? $(wc -c < $OUT/bench_default_line.json) bytes; $(tail -n 3 $OUT/bench_default.err | tr '\n' ' ')"
python3 - <<'PY'
import json
d = json.load(open("/root/repo/gpurun_out/r06x/bench_default.json"))
print(d["value"], d["roofline"]["frac"], d["roofline"].get("traffic_source", "")[-40:])
for k, l in d["other_configs"].items():
    r = l.get("roofline") or {}
    print("  ", k, l.get("value"), r.get("avg_kernel_ms"), r.get("frac"), (r.get("request_bound") or {}).get("request_bound_frac"), (r.get("traffic_source") or "")[-24:])
PY
echo done

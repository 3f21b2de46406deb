#!/bin/bash
# round 3, GPU session 1: the full -m gpu suite, the default bench line, narrow-filter baselines with/without the XCD skew
set -u
O=gpurun_out/r03
mkdir -p $O
python -c 'import __graft_entry__ as g; g.build()' > $O/build.log 2>&1
( time python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt
( time python bench.py --steps 5 --warmup 2 ) > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_default.err
for skew in 0 1; do
  for wl in "readme" "readme --read-len 360" "c1"; do
    tag=$(echo "$wl" | tr -d ' -')
    RB_PHASE_XCD_SKEW=$skew python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/tune_${tag}_skew$skew.json 2>> $O/tune.err
    python - <<PY
import json
d=json.load(open("$O/tune_${tag}_skew$skew.json"))
print("$wl skew=$skew", round(d["value"]/1e6,2), "M reads/s", round(d["roofline"]["avg_kernel_ms"],2), "ms")
PY
  done
done

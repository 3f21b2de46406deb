#!/bin/bash
# round 3, GPU session 29: full GPU suite on the tree with the new slice / window rule and the merge cost model; default bench
set -u
O=gpurun_out/r03
mkdir -p $O
python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/pytest_gpu_full_h.txt
cat $O/pytest_gpu_full_h.txt
python bench.py > $O/bench_default_h.json 2> $O/bench_default_h.err
python - <<PY
import json
d=json.load(open("$O/bench_default_h.json"))
print(d["value"], d["roofline"]["frac"], {k:(v.get("value"),) for k,v in d.get("other_configs",{}).items()})
PY

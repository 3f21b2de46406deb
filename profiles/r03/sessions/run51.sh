#!/bin/bash
# round 3, GPU session 51: final tree: second-to-last evidence collection (deep parity incl. the narrow merged workloads), bench of those workloads
set -u
O=gpurun_out/r03
mkdir -p $O
for w in targets3 deplete_target; do for L in 250 360; do
  python bench.py --workload $w --read-len $L --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/m_${w}_$L.json 2>> $O/m.err
  RB_MERGE=0 python bench.py --workload $w --read-len $L --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/m_${w}_${L}_apart.json 2>> $O/m.err
  python - <<PY
import json
a=json.load(open("$O/m_${w}_$L.json")); b=json.load(open("$O/m_${w}_${L}_apart.json"))
print("$w $L: merged %.2f M reads/s (%.2f ms, %s), apart %.2f M (%.2f ms)" % (a["value"]/1e6, a["roofline"]["avg_kernel_ms"], a["roofline"]["kernel"], b["value"]/1e6, b["roofline"]["avg_kernel_ms"]))
PY
done; done
bash profiles/deep_parity_r03.sh > $O/deep_parity_j.txt 2>&1
cat $O/deep_parity_j.txt

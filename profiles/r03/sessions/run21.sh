#!/bin/bash
# round 3, GPU session 21: how many chunks/s ONE GPU takes through the replay before the 1 ms SLO goes (config 5 asks for 150 k over 8 GPUs)
set -u
O=gpurun_out/r03
mkdir -p $O
for rate in 150000 300000 600000 1000000 1500000 2000000 2500000; do
  python bench.py --workload c5 --rate $rate --replay-seconds 1.0 > $O/c5_rate$rate.json 2>> $O/tune.err
  python - <<PY
import json
d=json.load(open("$O/c5_rate$rate.json"))
l=d["latency"]; v=d["live_step"]
print("rate %8d  served %9.0f/s  p50 %.3f p99 %.3f p99.9 %.3f ms  batch mean %.1f max %d  | live step: kept_up %s p50 %.3f p99 %.3f ms" % ($rate, d["value"], l["p50_ms"], l["p99_ms"], l["p99.9_ms"], d["config"]["micro_batch_reads"]["mean"], d["config"]["micro_batch_reads"]["max"], v["kept_up"], v["p50_ms"], v["p99_ms"]))
PY
done

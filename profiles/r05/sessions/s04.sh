#!/bin/bash
# r05 session 4: (a) c4 target load policy / launch arrangement A/B; (b) the first multi-GPU run rehearsed on one GPU: the driver's command
# with N = 2, 4, 8 ranks on this GPU (gloo, same-GPU hook; batches / N so that the GPU does the work of one N = 1 run), wall time, HBM in use
# after every leg (all ranks' replicas on ONE device: the upper bound of what any device of a real node holds), size of the final line
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s04
mkdir -p $OUT
cd $R
timeout 600 python3 profiles/c4_target_loads.py > $OUT/c4_target_loads.txt 2>&1
cat $OUT/c4_target_loads.txt | grep -v amdgpu.ids
for n in 2 4 8; do
  ( time RB_BENCH_BACKEND=gloo RB_BENCH_SAME_GPU=1 RB_BENCH_READS_DIVISOR=$n RB_BENCH_DETAIL=$OUT/bench_gpus${n}_same_gpu.detail.json timeout 1500 python3 bench.py --gpus $n --steps 20 --warmup 5 ) > $OUT/bench_gpus${n}_same_gpu.json 2> $OUT/bench_gpus${n}_same_gpu.err
  echo "== N=$n rc=$? line bytes $(wc -c < $OUT/bench_gpus${n}_same_gpu.json)"; tail -n 4 $OUT/bench_gpus${n}_same_gpu.err
  python3 - $OUT/bench_gpus${n}_same_gpu.detail.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(" value %.0f  ms_per_step %.1f  bench_seconds %s  hbm_in_use_at_exit %.1f GB" % (d["value"], d["ms_per_step"], d.get("bench_seconds"), d.get("hbm_in_use_at_exit_bytes",0)/1e9))
for k,v in d["other_configs"].items():
    print("  ", k, "value %.0f" % v.get("value",0), "leg_s", v.get("leg_seconds"), "hbm after leg %.1f GB" % (v.get("hbm_in_use_after_leg_bytes",0)/1e9), v.get("error",""))
PY
done
# a rank that dies in the middle of the real run: a parseable line with an error, non-zero exit, no hang
( time RB_BENCH_BACKEND=gloo RB_BENCH_SAME_GPU=1 RB_BENCH_READS_DIVISOR=40 RB_BENCH_TEST_DIE_RANK=3 RB_BENCH_DETAIL=$OUT/dead_rank.detail.json timeout 600 python3 bench.py --gpus 4 --steps 2 --warmup 1 ) > $OUT/dead_rank.json 2> $OUT/dead_rank.err
echo "== dead rank: rc=$? line: $(cat $OUT/dead_rank.json | cut -c1-400)"; tail -n 3 $OUT/dead_rank.err

cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s26; mkdir -p $O
# the default command as two gloo ranks on the one GPU, pool legs forced into rank 0's child: store wait + child + line structure
( time RB_BENCH_POOL_CHILD=1 RB_BENCH_POOL_DEVICES=0,0 RB_BENCH_BACKEND=gloo RB_BENCH_SAME_GPU=1 RB_BENCH_READS_DIVISOR=20 timeout 1200 python3 bench.py --gpus 2 --steps 2 --warmup 1 ) > $O/bench_gpus2_child.json 2> $O/bench_gpus2_child.err; tail -n 5 $O/bench_gpus2_child.err
python3 - $O/bench_gpus2_child.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("n_gpus", d["n_gpus"], "value", d["value"], "preflight", d["ranks"]["xgmi_preflight"])
for k,v in d["other_configs"].items():
    print(k, v.get("n_gpus"), "%.3g" % v.get("value",0), v.get("in_child_process"), v.get("error"), (v.get("parity") or {}).get("pool_outputs_equal_single_engine"))
PY

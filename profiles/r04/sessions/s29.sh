cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s29; mkdir -p $O
for w in "readme 0" "readme 360" "targets3 0" "deplete_target 0" "c1 0" "w1_64mib 0"; do
  set -- $w
  timeout 300 python3 bench.py --workload $1 --read-len $2 --steps 10 --warmup 3 --no-cpu-baseline --no-latency > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err
  python3 - $O/bench_$1_$2.json $1 $2 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], sys.argv[3], "%.2f M reads/s" % (d["value"]/1e6), "K1 %.3f ms" % d["roofline"]["avg_kernel_ms"], json.dumps({k:(round(v,3) if isinstance(v,float) else v) for k,v in d["roofline"].get("request_roofline",{}).items() if k!="source"}))
except Exception as ex:
    print(sys.argv[2], sys.argv[3], "failed", ex)
PY
done

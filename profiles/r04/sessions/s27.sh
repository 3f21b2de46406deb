cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s27; mkdir -p $O
for w in c1 w1_64mib; do
  timeout 300 python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-latency > $O/bench_$w.json 2> $O/bench_$w.err
  python3 - $O/bench_$w.json $w <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], d["config"]["read_len"], "%.2f M reads/s" % (d["value"]/1e6), "K1 %.3f ms" % d["roofline"]["avg_kernel_ms"], json.dumps({k:v for k,v in d["roofline"].get("request_roofline",{}).items() if k!="source"}))
except Exception as ex:
    print(sys.argv[2], "failed", ex)
PY
done

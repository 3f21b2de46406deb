cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s50; mkdir -p $O
( time timeout 1200 python3 bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; tail -n 4 $O/bench_default.err
python3 - $O/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("c3 %.3f M reads/s frac %.3f of probe %.4f" % (d["value"]/1e6, d["roofline"]["frac"], d["roofline"].get("frac_of_measured_read_peak",0)))
for k,v in d["other_configs"].items():
    r=v.get("roofline") or {}
    q=r.get("request_roofline") or {}
    print(k, "%.2f M" % (v.get("value",0)/1e6), "frac", r.get("frac"), "K1", r.get("avg_kernel_ms"), "req", q.get("frac"), "err", v.get("error"), (v.get("parity") or {}).get("raw_max_mismatches"), (v.get("parity") or {}).get("decision_mismatches"))
PY
timeout 600 python3 profiles/engines_on_one_gpu.py --shapes readme > $O/engines_on_one_gpu.txt 2>&1; cut -c1-120 $O/engines_on_one_gpu.txt
for w in "readme 0" "readme 360"; do set -- $w; timeout 300 python3 bench.py --workload $1 --read-len $2 --steps 10 --warmup 3 --cpu-seconds 20 --no-latency > $O/bench_$1_$2.json 2> $O/bench_$1_$2.err; python3 - $O/bench_$1_$2.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["config"]["read_len"], d["value"], d["parity"], {k:v for k,v in d["roofline"]["request_roofline"].items() if k!="source"})
PY
done

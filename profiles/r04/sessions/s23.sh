cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s23; mkdir -p $O
( time timeout 1500 python3 -m pytest tests -m gpu -q ) > $O/pytest_gpu.txt 2>&1; tail -n 6 $O/pytest_gpu.txt | cut -c1-300
( time timeout 900 python3 -m pytest tests -m gpuperf -q ) > $O/pytest_gpuperf.txt 2>&1; tail -n 6 $O/pytest_gpuperf.txt | cut -c1-300
( time timeout 1200 python3 bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; tail -n 4 $O/bench_default.err
python3 - $O/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("c3 %.3f M reads/s frac %.3f of probe %.4f parity %s" % (d["value"]/1e6, d["roofline"]["frac"], d["roofline"].get("frac_of_measured_read_peak",0), d["parity"]))
for k,v in d["other_configs"].items():
    r=v.get("roofline") or {}
    print(k, "%.2f M" % (v.get("value",0)/1e6), "frac", r.get("frac"), "probe", r.get("frac_of_measured_read_peak"), "req", (r.get("request_roofline") or {}).get("frac"), "err", v.get("error"), (v.get("parity") or {}).get("raw_max_mismatches"), (v.get("parity") or {}).get("decision_mismatches"))
PY

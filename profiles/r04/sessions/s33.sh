cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s33; mkdir -p $O
( time timeout 1200 python3 bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; tail -n 4 $O/bench_default.err
python3 - $O/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("c3 %.3f M reads/s frac %.3f of probe %.4f parity %s" % (d["value"]/1e6, d["roofline"]["frac"], d["roofline"].get("frac_of_measured_read_peak",0), {k:d["parity"][k] for k in ("checked_reads","decision_mismatches","raw_max_mismatches")}))
for k,v in d["other_configs"].items():
    r=v.get("roofline") or {}
    q=r.get("request_roofline") or {}
    print(k, "%.2f M" % (v.get("value",0)/1e6), "frac", r.get("frac"), "probe", r.get("frac_of_measured_read_peak"), "req", q.get("frac"), q.get("bound"), q.get("sum_of_terms_over_kernel_ms"), "err", v.get("error"), (v.get("parity") or {}).get("raw_max_mismatches"), (v.get("parity") or {}).get("decision_mismatches"))
PY
( time timeout 1200 python3 -m pytest tests/test_bench_ranks.py -m "gpu or gpuperf" -q ) > $O/pytest_bench_ranks.txt 2>&1; tail -n 4 $O/pytest_bench_ranks.txt

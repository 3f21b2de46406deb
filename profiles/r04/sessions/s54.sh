cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s54; mkdir -p $O
( time timeout 1200 python3 bench.py --steps 20 --warmup 5 ) > $O/bench_default.json 2> $O/bench_default.err; tail -n 4 $O/bench_default.err
python3 - $O/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("c3 %.3f M reads/s frac %.3f of probe %.4f" % (d["value"]/1e6, d["roofline"]["frac"], d["roofline"].get("frac_of_measured_read_peak",0)))
for k,v in d["other_configs"].items():
    r=v.get("roofline") or {}
    q=r.get("request_roofline") or {}
    pl=(r.get("plan") or [{}])[0]
    print(k, "%.2f M" % (v.get("value",0)/1e6), "frac", r.get("frac"), "req", q.get("frac"), pl.get("phase_slices"), pl.get("phase_slice_bytes"), "err", v.get("error"), (v.get("parity") or {}).get("raw_max_mismatches"))
PY
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1

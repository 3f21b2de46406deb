cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s34; mkdir -p $O
# does the size the table is allocated with matter (page-table fragments)?  c3np2 (4.72 GB) and c3 (8 GiB + 64 B), 2 M reads per launch
for rep in 1 2; do
 for R in none 2 1024 0; do
  for w in c3np2 c3; do
    if [ $R = none ]; then unset RB_ALLOC_ROUND_MIB; else export RB_ALLOC_ROUND_MIB=$R; fi
    timeout 300 python3 bench.py --workload $w --reads 2000000 --steps 4 --warmup 1 --no-cpu-baseline --no-latency > $O/b.json 2> $O/b.err
    python3 - $O/b.json $w $R <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r=d["roofline"]
    print(sys.argv[2], "round", sys.argv[3], "%.3f M reads/s" % (d["value"]/1e6), "frac %.4f" % r["frac"], "probe %.0f GB/s" % r["read_peak_probe"]["GBps"], "of probe %.4f" % r["frac_of_measured_read_peak"])
except Exception as ex:
    print(sys.argv[2], sys.argv[3], "failed", ex)
PY
  done
 done
done

#!/bin/bash
# r05 session 12: placement by trial in the library: its test, then c3np2 / grch38_f100k / c3 legs several times (fresh process each: does the
# kept allocation always probe fast, and does K1 follow?)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s12
mkdir -p $OUT
cd $R
( timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "placed_by_trial or round_trip or clone or resize" ) > $OUT/pytest_placement.txt 2>&1
tail -n 3 $OUT/pytest_placement.txt | cut -c1-300
for i in 1 2 3 4 5 6; do
  for w in c3np2 grch38_f100k; do
    RB_BENCH_DETAIL=$OUT/${w}_$i.json timeout 600 python3 bench.py --workload $w --reads 2000000 --steps 3 --warmup 1 --no-cpu-baseline --no-latency > /dev/null 2> $OUT/${w}_$i.err
    python3 - $OUT/${w}_$i.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(sys.argv[1].split("/")[-1], "value %.0f frac %.4f probe %.0f of probe %.4f placement %s setup_s %.1f" % (d["value"], r["frac"], r["read_peak_probe"]["GBps"], r["frac_of_measured_read_peak"], r.get("placement"), d["setup_s"]))
PY
  done
done
RB_BENCH_DETAIL=$OUT/c3.json timeout 600 python3 bench.py --workload c3 --reads 2000000 --steps 3 --warmup 1 --no-cpu-baseline --no-latency > /dev/null 2> $OUT/c3.err
python3 -c "
import json; d=json.load(open('$OUT/c3.json')); r=d['roofline']; print('c3', d['value'], r['frac'], r.get('placement'), d['setup_s'])"

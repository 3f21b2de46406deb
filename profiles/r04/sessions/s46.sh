cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s46; mkdir -p $O
# slices of any length for the four-word build (RB_PHASE_N_SLICES: N equal slices instead of ten of 4 MiB): fewer windows = fewer predicated loads for the TA
( RB_PHASE_N_SLICES=7 timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "packed or merged or merge" ) > $O/pytest_n7.txt 2>&1; tail -n 2 $O/pytest_n7.txt | cut -c1-200
for rl in 0 360; do
 for NS in 10 9 8 7 6; do
  for T in 400 450 500 550 600 700; do
    RB_PHASE_N_SLICES=$NS timeout 300 python3 bench.py --workload readme --read-len $rl --phased 1,4096,$T,0 --steps 6 --warmup 2 --no-cpu-baseline --no-latency > $O/b.json 2> $O/b.err
    python3 - $O/b.json $rl $NS $T <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p=d["roofline"]["plan"][0]
    print("readme", sys.argv[2], "slices", sys.argv[3], "ticks", sys.argv[4], "%.2f M reads/s" % (d["value"]/1e6), "K1 %.3f ms" % d["roofline"]["avg_kernel_ms"], "mism", (d.get("parity") or {}).get("raw_max_mismatches"))
except Exception as ex:
    print("readme", sys.argv[2], sys.argv[3], "failed", ex)
PY
  done
 done
done

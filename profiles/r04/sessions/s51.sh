cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s51; mkdir -p $O
# equal-length slices for the one- and two-word builds too? (RB_PHASE_N_SLICES, any stride) -- window sweeps
( RB_PHASE_N_SLICES=3 timeout 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -q -k "packed or merged or merge or phase or range or short" ) > $O/pytest_n3.txt 2>&1; tail -n 2 $O/pytest_n3.txt | cut -c1-200
one() { # workload nslices ticks
    if [ $2 = rule ]; then unset RB_PHASE_N_SLICES; else export RB_PHASE_N_SLICES=$2; fi
    if [ $3 = rule ]; then PH=""; else PH="--phased 1,4096,$3,0"; fi
    timeout 300 python3 bench.py --workload $1 $PH --steps 6 --warmup 2 --no-cpu-baseline --no-latency > $O/b.json 2> $O/b.err
    python3 - $O/b.json $1 $2 $3 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p=d["roofline"]["plan"][0]
    print(sys.argv[2], "slices", sys.argv[3], "ticks", sys.argv[4], "%.2f M reads/s" % (d["value"]/1e6), "K1 %.3f ms" % d["roofline"]["avg_kernel_ms"], p.get("phase_slices"), p.get("phase_window_ticks"))
except Exception as ex:
    print(sys.argv[2], sys.argv[3], "failed", ex)
PY
}
one targets3 rule rule
for NS in 4 3; do for T in 900 1100 1300 1500 1800; do one targets3 $NS $T; done; done
one c1 rule rule
for NS in 4 3; do for T in 1200 1500 1800 2000; do one c1 $NS $T; done; done
one w1_64mib rule rule
for NS in 14 12 10; do for T in 400 500 600 750; do one w1_64mib $NS $T; done; done

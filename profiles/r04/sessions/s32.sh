cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s32; mkdir -p $O
# every other wave trails the clock by half a window (RB_TRAIL_HALF=1, exp) against the shipped form (base); slices of 2 and 4 MiB; window sweep
export RB_AMD_LIBRARY=$GRAFT_REPO_ROOT/readbouncer_amd/libreadbouncer_amd_exp.so; ( timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "packed or merged or merge or phased" ) > $O/pytest_exp.txt 2>&1; tail -n 2 $O/pytest_exp.txt | cut -c1-200
one() { # workload ticks lg lib
    if [ $4 = base ]; then unset RB_AMD_LIBRARY; else export RB_AMD_LIBRARY=$GRAFT_REPO_ROOT/readbouncer_amd/libreadbouncer_amd_exp.so; fi
    RB_PHASE_SLICE_LOG2=$3 timeout 300 python3 bench.py --workload $1 --phased 1,4096,$2,0 --steps 6 --warmup 2 --no-cpu-baseline --no-latency > $O/b.json 2> $O/b.err
    python3 - $O/b.json $1 $4 $2 $3 <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p=d["roofline"]["plan"][0]
    print(sys.argv[2], "slice 2^%s" % sys.argv[5], sys.argv[3], "ticks", sys.argv[4], "%.2f M reads/s" % (d["value"]/1e6), "K1 %.3f ms" % d["roofline"]["avg_kernel_ms"], p.get("phase_window_ticks"), p.get("phase_slices"))
except Exception as ex:
    print(sys.argv[2], sys.argv[3], "failed", ex)
PY
}
for LG in 21 22; do
  for T in 350 450 550 700 900 1100; do for lib in exp base; do one targets3 $T $LG $lib; done; done
  for T in 250 350 450 500 600; do for lib in exp base; do one readme $T $LG $lib; done; done
  for T in 500 700 900 1200 1500; do for lib in exp base; do one c1 $T $LG $lib; done; done
done

cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s31; mkdir -p $O
# next-slice prefetch (RB_PREFETCH_NEXT=1; exp7 = seven waves with a small spill, exp6 = six waves) against the shipped form, targets3,
# slices of 4 / 2 / 1 MiB, window sweep
export RB_AMD_LIBRARY=$GRAFT_REPO_ROOT/readbouncer_amd/libreadbouncer_amd_exp7.so; ( timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "packed or merged or merge" ) > $O/pytest_exp.txt 2>&1; tail -n 2 $O/pytest_exp.txt | cut -c1-200
for LG in 22 21 20; do
 for T in 250 350 450 550 700 900 1100; do
  for lib in exp7 exp6 base; do
    if [ $lib = base ]; then unset RB_AMD_LIBRARY; else export RB_AMD_LIBRARY=$GRAFT_REPO_ROOT/readbouncer_amd/libreadbouncer_amd_$lib.so; fi
    RB_PHASE_SLICE_LOG2=$LG timeout 300 python3 bench.py --workload targets3 --phased 1,4096,$T,0 --steps 6 --warmup 2 --no-cpu-baseline --no-latency > $O/b.json 2> $O/b.err
    python3 - $O/b.json $lib $T $LG <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p=d["roofline"]["plan"][0]
    print("targets3 slice 2^%s" % sys.argv[4], sys.argv[2], "ticks", sys.argv[3], "%.2f M reads/s" % (d["value"]/1e6), "K1 %.3f ms" % d["roofline"]["avg_kernel_ms"], p.get("phase_window_ticks"), p.get("phase_slices"))
except Exception as ex:
    print("targets3", sys.argv[2], "failed", ex)
PY
  done
 done
done

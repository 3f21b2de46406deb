cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s30; mkdir -p $O
# two-word one-lane build (targets3): a second k-mer per batch gathered into LDS (exp) against the shipped form (base), window sweep
export RB_AMD_LIBRARY=$GRAFT_REPO_ROOT/readbouncer_amd/libreadbouncer_amd_exp.so; ( timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "packed or merged or merge" ) > $O/pytest_exp.txt 2>&1; tail -n 2 $O/pytest_exp.txt | cut -c1-200
for T in 0 500 650 800 950 1100 1300; do
  for lib in exp base; do
    if [ $lib = exp ]; then export RB_AMD_LIBRARY=$GRAFT_REPO_ROOT/readbouncer_amd/libreadbouncer_amd_exp.so; else unset RB_AMD_LIBRARY; fi
    if [ $T = 0 ]; then PH=""; else PH="--phased 1,4096,$T,0"; fi
    timeout 300 python3 bench.py --workload targets3 $PH --steps 6 --warmup 2 --no-cpu-baseline --no-latency > $O/b.json 2> $O/b.err
    python3 - $O/b.json $lib $T <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p=d["roofline"]["plan"][0]
    print("targets3", sys.argv[2], "ticks", sys.argv[3], "%.2f M reads/s" % (d["value"]/1e6), "K1 %.3f ms" % d["roofline"]["avg_kernel_ms"], p.get("phase_window_ticks"), p.get("phase_slices"))
except Exception as ex:
    print("targets3", sys.argv[2], "failed", ex)
PY
  done
done

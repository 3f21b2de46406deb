cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s19; mkdir -p $O
# window sweep: clock-watching passes (RB_TIMED_PASS=1, exp) against the shipped form (base), README shape 250 bp and 360 bp
export RB_AMD_LIBRARY=$GRAFT_REPO_ROOT/readbouncer_amd/libreadbouncer_amd_exp.so; ( timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "packed or merged or merge" ) > $O/pytest_exp.txt 2>&1; tail -n 2 $O/pytest_exp.txt | cut -c1-200
for rl in 0 360; do
 for T in 150 200 250 300 350 400 450 500 650; do
  for lib in exp base; do
    if [ $lib = exp ]; then export RB_AMD_LIBRARY=$GRAFT_REPO_ROOT/readbouncer_amd/libreadbouncer_amd_exp.so; else unset RB_AMD_LIBRARY; fi
    timeout 300 python3 bench.py --workload readme --read-len $rl --phased 1,4096,$T,0 --steps 6 --warmup 2 --no-cpu-baseline --no-latency > $O/b.json 2> $O/b.err
    python3 - $O/b.json $rl $lib $T <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p=d["roofline"]["plan"][0]
    print("readme", sys.argv[2], sys.argv[3], "ticks", sys.argv[4], "%.2f M reads/s" % (d["value"]/1e6), "K1 %.3f ms" % d["roofline"]["avg_kernel_ms"], p.get("phase_window_ticks"), p.get("phase_slices"))
except Exception as ex:
    print("readme", sys.argv[2], sys.argv[3], "failed", ex)
PY
  done
 done
done

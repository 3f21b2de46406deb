cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s37; mkdir -p $O
# TIMING ONLY: the four-word build with its hashing replaced by a few integer instructions (wrong results) -- what hiding the hashing could gain
for T in 350 400 450 500 550; do
  for lib in exp base; do
    if [ $lib = base ]; then unset RB_AMD_LIBRARY; else export RB_AMD_LIBRARY=$GRAFT_REPO_ROOT/readbouncer_amd/libreadbouncer_amd_exp.so; fi
    timeout 300 python3 bench.py --workload readme --phased 1,4096,$T,0 --steps 6 --warmup 2 --no-cpu-baseline --no-latency > $O/b.json 2> $O/b.err
    python3 - $O/b.json $lib $T <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("readme", sys.argv[2], "ticks", sys.argv[3], "%.2f M reads/s" % (d["value"]/1e6), "K1 %.3f ms" % d["roofline"]["avg_kernel_ms"])
except Exception as ex:
    print("readme", sys.argv[2], "failed", ex)
PY
  done
done

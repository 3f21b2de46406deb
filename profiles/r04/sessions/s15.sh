cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s15; mkdir -p $O
# second k-mer of a batch gathered into LDS (RB_LDS_GATHER=1 build of rb_kernels.hip) against the shipped form: four-word one-lane builds
export RB_AMD_LIBRARY=$GRAFT_REPO_ROOT/readbouncer_amd/libreadbouncer_amd_exp.so
( timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "packed or merged or merge" ) > $O/pytest_exp.txt 2>&1; tail -n 4 $O/pytest_exp.txt | cut -c1-300
for rl in 0 360; do
  for lib in exp base exp base; do
    if [ $lib = exp ]; then export RB_AMD_LIBRARY=$GRAFT_REPO_ROOT/readbouncer_amd/libreadbouncer_amd_exp.so; else unset RB_AMD_LIBRARY; fi
    timeout 300 python3 bench.py --workload readme --read-len $rl --steps 10 --warmup 3 --no-cpu-baseline --no-latency > $O/bench_readme${rl}_$lib.json 2> $O/bench_readme${rl}_$lib.err
    python3 - $O/bench_readme${rl}_$lib.json $rl $lib <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("readme", sys.argv[2], sys.argv[3], "%.2f M reads/s" % (d["value"]/1e6), "K1 %.3f ms" % d["roofline"]["avg_kernel_ms"])
except Exception as ex:
    print("readme", sys.argv[2], sys.argv[3], "failed", ex)
PY
  done
done

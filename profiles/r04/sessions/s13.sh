cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s13; mkdir -p $O
# exec-masked gathers (RB_EXEC_MASKED=1 build of rb_kernels.hip) against the bounds-check form: four-word / three-word one-lane builds
export RB_AMD_LIBRARY=$GRAFT_REPO_ROOT/readbouncer_amd/libreadbouncer_amd_exp.so
( timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -k "packed or merged or merge" ) > $O/pytest_exp.txt 2>&1; tail -n 4 $O/pytest_exp.txt | cut -c1-300
for w in readme readme_360bp; do
  for lib in exp base exp base; do
    if [ $lib = exp ]; then export RB_AMD_LIBRARY=$GRAFT_REPO_ROOT/readbouncer_amd/libreadbouncer_amd_exp.so; else unset RB_AMD_LIBRARY; fi
    timeout 300 python3 bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-latency > $O/bench_${w}_$lib.json 2> $O/bench_${w}_$lib.err
    python3 - $O/bench_${w}_$lib.json $w $lib <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], sys.argv[3], "%.2f M reads/s" % (d["value"]/1e6), "K1 %.3f ms" % d["roofline"]["avg_kernel_ms"], "parity", (d.get("parity") or {}).get("mismatches"), (d.get("parity") or {}).get("raw_max_mismatches"))
PY
  done
done

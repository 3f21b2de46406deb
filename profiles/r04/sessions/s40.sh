cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s40; mkdir -p $O
( timeout 900 python3 -m pytest tests/test_bench_ranks.py -q -m gpu -k "request_roofline or bound" ) 2>&1 | tail -n 4
( timeout 900 python3 -m pytest tests/test_bench_ranks.py -q -m gpuperf ) 2>&1 | tail -n 4

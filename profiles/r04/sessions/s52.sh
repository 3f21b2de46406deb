cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s52; mkdir -p $O
( time timeout 1500 python3 -m pytest tests -m gpu -q ) > $O/pytest_gpu.txt 2>&1; tail -n 6 $O/pytest_gpu.txt | cut -c1-300
( time timeout 900 python3 -m pytest tests -m gpuperf -q ) > $O/pytest_gpuperf.txt 2>&1; tail -n 6 $O/pytest_gpuperf.txt | cut -c1-300
RB_FUZZ_SEEDS=600 timeout 1200 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -n 3

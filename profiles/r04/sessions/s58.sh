cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s58; mkdir -p $O
( time timeout 1500 python3 -m pytest tests -m gpu -q ) > $O/pytest_gpu.txt 2>&1; tail -n 6 $O/pytest_gpu.txt | cut -c1-300
( time timeout 900 python3 -m pytest tests -m gpuperf -q ) > $O/pytest_gpuperf.txt 2>&1; tail -n 6 $O/pytest_gpuperf.txt | cut -c1-300

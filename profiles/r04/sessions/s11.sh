cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04s11
( time timeout 1500 python3 -m pytest tests -m gpu -q ) > gpurun_out/r04s11/pytest_gpu.txt 2>&1; tail -n 6 gpurun_out/r04s11/pytest_gpu.txt | cut -c1-300
timeout 900 python3 profiles/calibrate_gain.py > gpurun_out/r04s11/calibrate_gain.txt 2>&1; cut -c1-260 gpurun_out/r04s11/calibrate_gain.txt

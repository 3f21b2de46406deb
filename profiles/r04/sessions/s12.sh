cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04s12
( time timeout 1500 python3 -m pytest tests -m gpu -q ) > gpurun_out/r04s12/pytest_gpu.txt 2>&1; tail -n 6 gpurun_out/r04s12/pytest_gpu.txt | cut -c1-300
timeout 900 python3 profiles/calibrate_gain.py > gpurun_out/r04s12/calibrate_gain.txt 2>&1; cut -c1-230 gpurun_out/r04s12/calibrate_gain.txt
timeout 900 python3 profiles/cli_readme250.py 64000000 - quick > gpurun_out/r04s12/cli_throughput_64M_reads.txt 2>&1; grep -E "defaults|classifiers" gpurun_out/r04s12/cli_throughput_64M_reads.txt | cut -c1-330

mkdir -p gpurun_out/r02
hipcc -O3 --offload-arch=gfx950 profiles/gather_probe.hip -o /tmp/gather_probe && timeout 400 /tmp/gather_probe phased_pf > gpurun_out/r02/gather_phased_pf.txt 2>&1; cat gpurun_out/r02/gather_phased_pf.txt
timeout 1500 python -m pytest tests/test_host_cli.py tests/test_gpu_parity.py -m gpu -q > gpurun_out/r02/pytest_gpu2.txt 2>&1; tail -25 gpurun_out/r02/pytest_gpu2.txt

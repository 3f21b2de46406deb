mkdir -p gpurun_out/r02
hipcc -O3 --offload-arch=gfx950 profiles/gather_probe.hip -o /tmp/gather_probe && timeout 300 /tmp/gather_probe phased > gpurun_out/r02/gather_phased.txt 2>&1; cat gpurun_out/r02/gather_phased.txt
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02/pytest_gpu.txt 2>&1; tail -25 gpurun_out/r02/pytest_gpu.txt

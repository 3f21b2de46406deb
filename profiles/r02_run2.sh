mkdir -p gpurun_out/r02
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r02/pytest_gpu.txt 2>&1; tail -15 gpurun_out/r02/pytest_gpu.txt
timeout 200 python bench.py --workload readme --steps 5 --warmup 1 --no-cpu-baseline --no-latency > gpurun_out/r02/readme_overlap.json 2> gpurun_out/r02/readme_overlap.err
timeout 200 python bench.py --workload readme --steps 5 --warmup 1 --no-cpu-baseline --no-latency --no-overlap > gpurun_out/r02/readme_serial.json 2> gpurun_out/r02/readme_serial.err
python3 -c "
import json
for n in ('overlap','serial'):
    try:
        d=json.load(open('gpurun_out/r02/readme_%s.json'%n)); print(n, round(d['value']), d['roofline']['avg_kernel_ms'])
    except Exception as e: print(n,'ERR',e)
"
timeout 600 bash profiles/run_gather_probe.sh r02 pmc_only

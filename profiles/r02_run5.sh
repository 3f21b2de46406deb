mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x > gpurun_out/r02/pytest_gpu3.txt 2>&1; tail -15 gpurun_out/r02/pytest_gpu3.txt
for ph in off 6,32,10 6,32,20 6,32,30 6,32,45 6,32,60; do
  timeout 200 python bench.py --workload readme --steps 5 --warmup 1 --no-cpu-baseline --no-latency --phased $ph > gpurun_out/r02/readme_ph_$ph.json 2> gpurun_out/r02/readme_ph_$ph.err
  python3 -c "
import json,sys
try:
    d=json.load(open('gpurun_out/r02/readme_ph_$ph.json')); print('readme phased $ph', round(d['value']), d['roofline']['avg_kernel_ms'], d['config']['decisions'])
except Exception as e: print('$ph','ERR',e)
"
done
for ph in off 6,32,20 6,32,30 6,32,45; do
  timeout 200 python bench.py --workload c1 --steps 5 --warmup 1 --no-cpu-baseline --no-latency --phased $ph > gpurun_out/r02/c1_ph_$ph.json 2> gpurun_out/r02/c1_ph_$ph.err
  python3 -c "
import json,sys
try:
    d=json.load(open('gpurun_out/r02/c1_ph_$ph.json')); print('c1 phased $ph', round(d['value']), d['roofline']['avg_kernel_ms'], d['config']['decisions'])
except Exception as e: print('$ph','ERR',e)
"
done

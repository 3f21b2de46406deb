mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x > gpurun_out/r02/pytest_gpu4.txt 2>&1; tail -8 gpurun_out/r02/pytest_gpu4.txt
for w in readme c1; do
for ph in off 0,0,30 6,32,15 6,32,20 6,32,30 6,32,40; do
  timeout 200 python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-latency --phased $ph > gpurun_out/r02/${w}_ph_$ph.json 2> gpurun_out/r02/${w}_ph_$ph.err
  python3 -c "
import json,sys
try:
    d=json.load(open('gpurun_out/r02/${w}_ph_$ph.json')); print('$w phased $ph', round(d['value']), d['roofline']['avg_kernel_ms'], d['config']['decisions'])
except Exception as e: print('$ph','ERR',e)
"
done
done

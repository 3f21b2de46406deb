mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x > gpurun_out/r02/pytest_gpu5.txt 2>&1; tail -5 gpurun_out/r02/pytest_gpu5.txt
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for ph in off 6,32,150,0 6,32,200,0 6,32,250,0 6,32,300,0 6,32,350,0 6,32,400,0; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02/stats2_readme_$ph -- python3 $R/bench.py --workload readme --steps 5 --warmup 1 --no-cpu-baseline --no-latency --phased $ph > $R/gpurun_out/r02/stats2_readme_$ph.log 2>&1
  f=$(find $R/gpurun_out/r02/stats2_readme_$ph -name "*kernel_stats.csv" | head -1)
  echo "== readme phased $ph"; grep -E "ibf_count" $f | sed -e 's/(rb::[^"]*"/"/' | cut -d, -f1-4 | head -6; grep -o '"value": [0-9.]*' $R/gpurun_out/r02/stats2_readme_$ph.log | head -1
done
cd $R
for ph in off 6,32,200,0 6,32,300,0 6,32,400,0 6,32,500,0 6,32,600,0; do
  timeout 200 python bench.py --workload c1 --steps 5 --warmup 1 --no-cpu-baseline --no-latency --phased $ph > gpurun_out/r02/c1b_ph_$ph.json 2> gpurun_out/r02/c1b_ph_$ph.err
  python3 -c "
import json,sys
try:
    d=json.load(open('gpurun_out/r02/c1b_ph_$ph.json')); print('c1 phased $ph', round(d['value']), d['roofline']['avg_kernel_ms'], d['config']['decisions'])
except Exception as e: print('$ph','ERR',e)
"
done

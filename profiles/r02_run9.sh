mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -x > gpurun_out/r02/pytest_gpu6.txt 2>&1; tail -5 gpurun_out/r02/pytest_gpu6.txt
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for ph in 6,32,250,0 6,32,350,0 6,32,450,0 6,32,350,4; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02/stats3_readme_$ph -- python3 $R/bench.py --workload readme --steps 5 --warmup 1 --no-cpu-baseline --no-latency --phased $ph > $R/gpurun_out/r02/stats3_readme_$ph.log 2>&1
  f=$(find $R/gpurun_out/r02/stats3_readme_$ph -name "*kernel_stats.csv" | head -1)
  echo "== readme phased $ph"; grep -E "ibf_count" $f | sed -e 's/(rb::[^"]*"/"/' | cut -d, -f1-4 | head -6; grep -o '"value": [0-9.]*' $R/gpurun_out/r02/stats3_readme_$ph.log | head -1
done
cd $R
timeout 200 python bench.py --workload readme --steps 10 --warmup 2 --cpu-seconds 5 --no-latency > gpurun_out/r02/readme_default.json 2> gpurun_out/r02/readme_default.err
python3 -c "
import json
d=json.load(open('gpurun_out/r02/readme_default.json')); print('readme default', round(d['value']), d['roofline']['avg_kernel_ms'], d.get('parity'), d['cpu_baseline']['value'])"

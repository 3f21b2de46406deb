mkdir -p gpurun_out/r02
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for ph in off 6,32,20 6,32,30 6,32,40 6,32,55; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02/stats_readme_$ph -- python3 $R/bench.py --workload readme --steps 5 --warmup 1 --no-cpu-baseline --no-latency --phased $ph > $R/gpurun_out/r02/stats_readme_$ph.log 2>&1
  f=$(find $R/gpurun_out/r02/stats_readme_$ph -name "*kernel_stats.csv" | head -1)
  echo "== readme phased $ph"; grep -E "ibf_count" $f | sed -e 's/(rb::IbfDev[^"]*"/"/' -e 's/(rb::FilterSet[^"]*"/"/' | cut -d, -f1-4 | head -6
done
cd $R
timeout 300 python bench.py --workload c5 --replay-seconds 2.0 > gpurun_out/r02/c5_cpp.json 2> gpurun_out/r02/c5_cpp.err; python3 -c "
import json
d=json.load(open('gpurun_out/r02/c5_cpp.json')); print('c5', d['value'], d['latency'], d['config']['dispatcher'], d['config']['micro_batch_reads'])"

mkdir -p gpurun_out/r02
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for ph in 6,32,400,0 6,32,450,0 6,32,500,0 6,32,550,0 6,32,600,0 6,32,700,0; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02/stats5_readme_$ph -- python3 $R/bench.py --workload readme --steps 5 --warmup 1 --no-cpu-baseline --no-latency --phased $ph > $R/gpurun_out/r02/stats5_readme_$ph.log 2>&1
  f=$(find $R/gpurun_out/r02/stats5_readme_$ph -name "*kernel_stats.csv" | head -1)
  echo "== readme phased $ph"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ibf_count" in r["Name"]:
        print(r["Name"].split("(")[0][-52:], r["Calls"], "avg ms %.3f" % (float(r["AverageNs"])/1e6))
PY
  grep -o '"value": [0-9.]*' $R/gpurun_out/r02/stats5_readme_$ph.log | head -1
done
cd $R
for ph in 6,32,400,0 6,32,500,0 6,32,600,0; do
  timeout 200 python bench.py --workload c1 --steps 5 --warmup 1 --no-cpu-baseline --no-latency --phased $ph > gpurun_out/r02/c1c_ph_$ph.json 2> gpurun_out/r02/c1c_ph_$ph.err
  python3 -c "
import json,sys
try:
    d=json.load(open('gpurun_out/r02/c1c_ph_$ph.json')); print('c1 phased $ph', round(d['value']), d['roofline']['avg_kernel_ms'])
except Exception as e: print('$ph','ERR',e)
"
done

#!/bin/bash
# round 3, GPU session 13: shipped values: full parity suite, narrow-filter numbers, SQ counters
set -u
O=gpurun_out/r03
mkdir -p $O
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
one() { local tag=$1; shift
  python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/z_$tag.json 2>> $O/tune.err
  python - <<PY
import json
d=json.load(open("$O/z_$tag.json"))
print("$tag", round(d["value"]/1e6,2), "M reads/s", round(d["roofline"]["avg_kernel_ms"],2), "ms")
PY
}
one readme250 --workload readme
one readme360 --workload readme --read-len 360
one c1 --workload c1
one t1_250 --workload mock_t1
one dep_250 --workload mock_deplete
one t1_360 --workload mock_t1 --read-len 360
one dep_360 --workload mock_deplete --read-len 360
one readme1500 --workload readme --read-len 1500 --reads 200000
bash profiles/collect_pmc.sh readme 1000000 $O/pmc_readme_new > /dev/null 2>&1
bash profiles/collect_pmc.sh readme 1000000 $O/pmc_readme360_new "--read-len 360" > /dev/null 2>&1
for d in pmc_readme_new pmc_readme360_new; do echo "== $d"; cat $O/$d/l2.summary.csv $O/$d/ea.summary.csv $O/$d/sq.summary.csv 2>/dev/null | cut -d, -f 8- | cut -c1-150 | grep -E "TCC_HIT|TCC_MISS|RDREQ_sum|WAIT_ANY|WAIT_INST_ANY|WAVE_CYCLES|INSTS_VALU" | head -40; done

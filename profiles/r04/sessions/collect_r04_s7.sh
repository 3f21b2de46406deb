#!/bin/bash
# Round 4, session 7: the driver's command on the tree with bit-packed merged tables; rocprofv3 stats + PMC passes of the narrow shapes
TAG=${1:-r04s7}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
T="timeout 1800"
( time $T python3 bench.py --steps 20 --warmup 5 ) > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -3 $OUT/bench_default.err
python3 - "$OUT/bench_default.json" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r=d["roofline"]; print("c3", round(d["value"]), round(r["frac"],4), r.get("frac_of_measured_read_peak"), d["parity"])
for k,v in d["other_configs"].items():
    rr=v.get("roofline") or {}
    print("  ", k, round(v.get("value",0)), rr.get("frac") and round(rr["frac"],4), rr.get("avg_kernel_ms"), rr.get("frac_of_measured_read_peak"), v.get("parity"), v.get("error"))
PY
# standalone legs again, after the GPU has been busy for minutes (session 6 ran them cold)
for w in readme targets3 deplete_target; do
  $T python3 bench.py --workload $w --steps 40 --warmup 10 --no-cpu-baseline --no-latency > $OUT/bench_$w.json 2> /dev/null
done
cd /tmp && export TMPDIR=/tmp
for w in readme targets3 deplete_target; do
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$w -- python3 $R/bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline --no-latency > $OUT/stats_$w.log 2>&1
  bash $R/profiles/collect_pmc.sh $w 1000000 $OUT/pmc_$w > /dev/null 2>&1
done
$T rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_readme360 -- python3 $R/bench.py --workload readme --read-len 360 --steps 10 --warmup 2 --no-cpu-baseline --no-latency > $OUT/stats_readme360.log 2>&1
bash $R/profiles/collect_pmc.sh readme 1000000 $OUT/pmc_readme360 "--read-len 360" > /dev/null 2>&1
for w in readme targets3 deplete_target readme360; do f=$(find $OUT/stats_$w -name "*kernel_stats.csv" | head -1); echo "== $w"; python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "ibf_count" in r["Name"]:
        print("  ", r["Name"].split("(")[0][-70:], r["Calls"], "avg ms %.4f" % (float(r["AverageNs"])/1e6))
PY
done
for f in $OUT/bench_readme.json $OUT/bench_targets3.json $OUT/bench_deplete_target.json; do python3 - "$f" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r=d["roofline"]; print(sys.argv[1].split("/")[-1], round(d["value"]), round(r["frac"],4), round(r["avg_kernel_ms"],3))
PY
done

#!/bin/bash
# r05 session 3: unit counters (TA / TCP / TCC / LDS) of the compacted-gather build g8c128 on the two-word and the four-word shape, and of the
# shipped build in the same session for the LDS / TA columns
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s03
mkdir -p $OUT
cd $R
lds_pass() { # tag workload
  # (added in round 6, ADVICE r5: as this session RAN, the variable was not set here, so rocprofv3 profiled bench.py's supervisor and the
  # measuring process was its child; the "lds" rows kept in negative/pmc_units_compact_g8c128_*.txt carry a note to that effect)
  export RB_BENCH_NO_SUPERVISOR=1
  cd /tmp && export TMPDIR=/tmp
  timeout -k 5 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$OUT/$1/lds" -- python3 "$R/bench.py" --workload $2 --reads 1000000 --steps 2 --warmup 1 --no-cpu-baseline --no-latency > "$OUT/$1/lds.log" 2>&1
  f=$(find "$OUT/$1/lds" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "ibf_count_max" in r.get("Kernel_Name",""):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()):
    print("lds %-44s dispatches %d mean %.6g" % (k, len(v), sum(v)/len(v)))
PY
  cd $R
}
for w in targets3 readme; do
  echo "== $w, compacted build g8c128"
  RB_AMD_LIBRARY=$R/readbouncer_amd/exp/libreadbouncer_amd_g8c128.so bash profiles/collect_pmc_units.sh $w 1000000 $OUT/compact_$w | tee $OUT/pmc_units_compact_g8c128_$w.txt
  RB_AMD_LIBRARY=$R/readbouncer_amd/exp/libreadbouncer_amd_g8c128.so lds_pass compact_$w $w | tee -a $OUT/pmc_units_compact_g8c128_$w.txt
  echo "== $w, shipped build"
  bash profiles/collect_pmc_units.sh $w 1000000 $OUT/shipped_$w | tee $OUT/pmc_units_shipped_$w.txt
done
find $OUT -name "*.csv" -size +200k -delete; find $OUT -name "*.db" -delete

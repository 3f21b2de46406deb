#!/bin/bash
# Which unit of the memory path a count kernel keeps busy: separate rocprofv3 --pmc passes over TA / TCP / TCC / TD / the L1 TLB
# (no trace domains beside --kernel-trace).  Usage: collect_pmc_units.sh <workload> <reads> <outdir> [extra bench flags]
set -u
W=$1; N=$2; OUT=$3; EXTRA=${4:-}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export RB_BENCH_NO_SUPERVISOR=1  # the profiled process is the one that measures (bench.py would otherwise run rank 0 as a child of a supervisor)
ARGS="--workload $W --reads $N --steps 2 --warmup 1 --no-cpu-baseline --no-latency $EXTRA"
pass() { # tag counters...
  local tag=$1; shift
  timeout -k 5 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$tag" -- python3 "$R/bench.py" $ARGS > "$OUT/$tag.log" 2>&1
  local f=$(find "$OUT/$tag" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$tag" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "ibf_count_max" in r.get("Kernel_Name",""):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in sorted(acc.items()):
    print("%s %-44s dispatches %d mean %.6g" % (sys.argv[2], k, len(v), sum(v)/len(v)))
PY
}
# (two counters of one block per pass: a larger request is refused -- "exceeds the capabilities of the hardware to collect" -- and the
# profiler then sits in its abort handler until it is killed)
pass ta TA_TA_BUSY_sum GRBM_GUI_ACTIVE
pass ta2 TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
pass tcc TCC_BUSY_sum TCC_CYCLE_sum
pass tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum

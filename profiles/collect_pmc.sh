#!/bin/bash
# Collects the HBM-traffic counters of the count kernel in separate rocprofv3 --pmc passes
# (MI355X_MICROARCH.md "HBM": FETCH_SIZE in its own pass; KiB units; on gfx950 a wide coalesced read
# stream is reported at exactly half its bytes).  Usage: collect_pmc.sh <workload> <reads> <outdir> [extra bench flags]
set -u
W=$1; N=$2; OUT=$3; EXTRA=${4:-}
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$OUT"
echo "$N" > "$OUT/reads.txt"
cd /tmp && export TMPDIR=/tmp
export RB_BENCH_NO_SUPERVISOR=1  # the profiled process is the one that measures (bench.py would otherwise run rank 0 as a child of a supervisor)
ARGS="--workload $W --reads $N --steps 2 --warmup 1 --no-cpu-baseline --no-latency $EXTRA"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 "$R/bench.py" $ARGS > "$OUT/fetch.log" 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d "$OUT/l2" -- python3 "$R/bench.py" $ARGS > "$OUT/l2.log" 2>&1
timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d "$OUT/ea" -- python3 "$R/bench.py" $ARGS > "$OUT/ea.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d "$OUT/sq" -- python3 "$R/bench.py" $ARGS > "$OUT/sq.log" 2>&1
for d in fetch l2 ea sq; do
  f=$(find "$OUT/$d" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && grep -E "ibf_count_max|Counter_Name" "$f" | cut -d, -f1-20 | head -12 > "$OUT/$d.summary.csv"
done
tail -2 "$OUT"/*.log | tail -12

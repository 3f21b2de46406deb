#!/bin/bash
# r06: unit and SQ counters of the two-word phased kernels, one build per call -- the shipped one-read build or the multi-read build
# (RB_MULTI_READS / RB_PHASE_XCD_SKEW through the library's tuning environment).  Separate rocprofv3 --pmc passes, two counters of a block each.
# Usage: collect_pmc_multi.sh <tag> <workload> <multi_reads> <xcd_skew> <outdir>
set -u
TAG=$1; W=$2; MR=$3; SK=$4; OUT=$5
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export RB_BENCH_NO_SUPERVISOR=1 RB_TUNING_ENV=1 RB_MULTI_READS=$MR RB_PHASE_XCD_SKEW=$SK
ARGS="--workload $W --reads 1000000 --steps 2 --warmup 1 --no-cpu-baseline --no-latency"
pass() {
  local tag=$1; shift
  timeout -k 5 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$TAG.$tag" -- python3 "$R/bench.py" $ARGS > "$OUT/$TAG.$tag.log" 2>&1
  local f=$(find "$OUT/$TAG.$tag" -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$TAG $tag" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(list); dur=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "ibf_count_max" in r.get("Kernel_Name",""):
        acc[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        dur[(r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
for k,v in sorted(acc.items()):
    print("%s %-50s %-34s dispatches %d mean %.6g  kernel %.3f ms" % (sys.argv[2], k[0], k[1], len(v), sum(v)/len(v), sum(dur[k])/len(dur[k])))
PY
  rm -rf "$OUT/$TAG.$tag"
}
pass sq1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU
pass sq2 SQ_BUSY_CYCLES SQ_WAVE_CYCLES
pass sq3 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pass sq4 SQ_INSTS_SALU SQ_INSTS_LDS
pass sq5 SQ_WAVES SQ_INSTS_VMEM_RD
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
pass tcc TCC_HIT_sum TCC_MISS_sum
pass ea TCC_EA0_RDREQ_sum
pass ta TA_TA_BUSY_sum GRBM_GUI_ACTIVE
pass ta2 TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_READ_WAVEFRONTS_sum

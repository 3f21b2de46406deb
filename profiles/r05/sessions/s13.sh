#!/bin/bash
# r05 session 13: what differs between a slow and a fast allocation of the same table?  L1-TLB counters of K1 on the slowest- and the
# fastest-probing copy (profiles/placement_trial.py launches K1 five times on each, slow / fast / slow / fast), placement trials off so that the
# script's own copies are plain allocations.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s13
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "tlb2 TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_PERMISSION_MISS_sum" "tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "ea TCC_EA0_RDREQ_sum TCC_EA0_RD_UNCACHED_32B_sum"; do
  set -- $pass; tag=$1; shift
  RB_PLACEMENT_TRIES_OFF=1 timeout -k 5 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$tag -- python3 $R/profiles/placement_trial.py c3np2 5 > $OUT/$tag.log 2>&1
  grep -E "allocation|probe" $OUT/$tag.log | cut -c1-120
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$tag" <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "ibf_count_max_kernel" in r.get("Kernel_Name","")]
by=collections.defaultdict(list)
for r in rows: by[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for name,v in sorted(by.items()):
    v.sort()
    vals=[x for _,x in v]
    # 4 groups of 5 launches: slow, fast, slow, fast
    g=[vals[i*5:(i+1)*5] for i in range(len(vals)//5)]
    print(sys.argv[2], name, "dispatches", len(vals), "| per group means:", ["%.4g" % (sum(x)/len(x)) for x in g if x])
PY
done
find $OUT -name "*.db" -delete; find $OUT -name "*kernel_trace.csv" -delete

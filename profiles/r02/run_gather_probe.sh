#!/bin/bash
# profiles/gather_probe.hip on the GPU box: the rate table, then one rocprofv3 --pmc pass per interesting case
# (fabric request sizes, L2 hits).  Usage: bash profiles/run_gather_probe.sh <tag>
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$TAG/gather_probe
mkdir -p $OUT
hipcc -O3 --offload-arch=gfx950 $R/profiles/gather_probe.hip -o /tmp/gather_probe || exit 1
if [ "${2:-all}" = "all" ]; then
  timeout 300 /tmp/gather_probe all > $OUT/rates.txt 2>&1
  cat $OUT/rates.txt
fi
cd /tmp && export TMPDIR=/tmp
i=0
for c in "full 20 8 0 0" "full 20 16 0 0" "xcd 20 8 0 0" "full 400 8 0 0"; do
  i=$((i+1))
  echo "== $c" >> $OUT/pmc.txt
  # at most two TCC counters per pass (six in one pass: "exceeds the capabilities of the hardware", and rocprofv3 then hangs)
  for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=$(echo $set | tr ' ' '_')
    timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/pmc_${i}_$tag -- /tmp/gather_probe one $c > $OUT/pmc_${i}_$tag.log 2>&1
    f=$(find $OUT/pmc_${i}_$tag -name "*counter_collection.csv" | head -1)
    [ -n "$f" ] && python3 $R/profiles/pmc_kernel.py "$f" gather >> $OUT/pmc.txt 2>&1
  done
done
cat $OUT/pmc.txt

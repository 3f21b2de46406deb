#!/bin/bash
# r06 session 37: session 33's exact sequence (c4 first, then the narrow shapes; the old harness) twice more, now with the reporter in place
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06l2
mkdir -p $OUT
cd $R
for i in 1 2; do
  ( time RB_SOAK_RACY=1 timeout 1500 python3 profiles/soak_determinism.py ) > $OUT/soak_determinism_s37_$i.txt 2>&1
  echo "run $i exit $?"; grep -v amdgpu.ids $OUT/soak_determinism_s37_$i.txt | cut -c1-300
done
echo done

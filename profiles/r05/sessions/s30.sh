#!/bin/bash
# r05 session 30: does the way the host waits inside hipStreamSynchronize explain the 4 us between "results are in host memory" and "the wait returns"?
# the probe with the device's schedule flag (spin / yield / block) and with the runtime's active-wait switches
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05x
mkdir -p $OUT
cd $R
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 profiles/graph_launch_probe.hip -o /tmp/graph_launch_probe || exit 1
for how in auto spin yield block; do
  echo "== schedule flag $how"; timeout 120 /tmp/graph_launch_probe $how 2>&1 | grep -E "^#|launched" | grep -E "^#|2 kernels x  1.0|3 kernels x  8.0"
done > $OUT/sync_wait_modes.txt 2>&1
for envs in "ROC_ACTIVE_WAIT_TIMEOUT=1000" "HIP_FORCE_ACTIVE_WAIT=1" "GPU_ENABLE_WAIT_FOR_FENCE=1" "HSA_ENABLE_INTERRUPT=0"; do
  echo "== $envs"; env $envs timeout 120 /tmp/graph_launch_probe 2>&1 | grep -E "launched" | grep -E "2 kernels x  1.0|3 kernels x  8.0"
done >> $OUT/sync_wait_modes.txt 2>&1
cat $OUT/sync_wait_modes.txt

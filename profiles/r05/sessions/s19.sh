#!/bin/bash
# r05 session 19 (run twice: the second time with the completion-word variant in the probe): would a hipGraph help the micro-batch chain?  two or three short dependent kernels + a synchronise, launched vs replayed
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05m
mkdir -p $OUT
cd $R
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 profiles/graph_launch_probe.hip -o /tmp/graph_launch_probe || exit 1
timeout 300 /tmp/graph_launch_probe > $OUT/graph_launch_probe.txt 2>&1
cat $OUT/graph_launch_probe.txt

#!/bin/bash
# r05 session 23: IBF::load_filter at scale: a 2 GiB and the 8 GiB filter from a file in memory into HBM
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05q
mkdir -p $OUT
cd $R
free -g | head -2; df -h /dev/shm | tail -1; nproc
timeout 600 python3 profiles/load_throughput.py c2 > $OUT/load_c2.txt 2>&1; grep -v amdgpu.ids $OUT/load_c2.txt
AVAIL=$(free -g | awk '/Mem:/{print $7}')
if [ "$AVAIL" -gt 64 ]; then timeout 900 python3 profiles/load_throughput.py c3 > $OUT/load_c3.txt 2>&1; grep -v amdgpu.ids $OUT/load_c3.txt; fi
ls /dev/shm | head

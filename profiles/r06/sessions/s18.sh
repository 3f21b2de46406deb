#!/bin/bash
# r06 session 18: the round's evidence on the final tree (collect_r06.sh), and IBF::load_filter at scale with the reader gang of rb_io.h
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash profiles/r06/collect_r06.sh r06z
timeout 900 python3 profiles/load_throughput.py c3 > gpurun_out/r06z/load_throughput_c3.txt 2>&1
tail -12 gpurun_out/r06z/load_throughput_c3.txt | cut -c1-250

#!/bin/bash
# r06 session 24: the equal cut extended to the one-word LDS-offset builds -- the guard over one-word points the rule changed -- and then the
# round's evidence once more on the final tree (collect_r06.sh: counters first, then the bench lines that replay them)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p $R/gpurun_out/r06y
timeout 1200 python3 profiles/phase_rule_check.py --points 1:200:13,1:250:13,1:200:16,1:250:16,1:300:16,1:250:10,2:250:18.9 > $R/gpurun_out/r06y/guard_one_word_equal.txt 2>&1
echo "guard (one-word, equal slices) exit $?"; grep -v "^      " $R/gpurun_out/r06y/guard_one_word_equal.txt | cut -c1-260
bash profiles/r06/collect_r06.sh r06y
timeout 900 python3 profiles/load_throughput.py c3 > gpurun_out/r06y/load_throughput_c3.txt 2>&1
tail -12 gpurun_out/r06y/load_throughput_c3.txt | cut -c1-250
echo done

#!/bin/bash
# r06 session 15: whole -m gpu suite on the tree with the four-word LDS-offset builds; guard of the refitted two-word windows; the README shape
# (four-word packed table) at 250 / 360 bp, register builds against LDS-offset builds (one round of six tiles or rounds of three at 360 bp)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06o
mkdir -p $OUT
cd $R
timeout 3000 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
tail -15 $OUT/pytest_gpu.txt
timeout 1500 python3 profiles/phase_rule_check.py --reads 1000000 --points 2:250:19,2:360:19,2:200:13,2:300:13,2:250:8,2:360:8,2:250:28,2:360:28,2:200:24,2:300:30,2:150:19,2:250:11,2:360:11 --factors 0.6,0.7,0.8,0.9,1.0,1.1,1.2,1.35,1.5 > $OUT/phase_rule_check_two_word.txt 2>&1
echo "guard exit $?" >> $OUT/phase_rule_check_two_word.txt
grep -E "^2-word|outside|guard exit" $OUT/phase_rule_check_two_word.txt | cut -c1-260
export RB_TUNING_ENV=1
timeout 900 python3 profiles/multi_reads_sweep.py --workloads readme,readme360 --rpw 0,1 --skew 2 --factors 0.6,0.7,0.8,0.9,1.0,1.1,1.2,1.35,1.5,1.75,2.0 2>&1 | grep -v amdgpu.ids | tee $OUT/wide_sweep.txt
echo "== RB_MULTI_WIDE_SIX=0"
RB_MULTI_WIDE_SIX=0 timeout 900 python3 profiles/multi_reads_sweep.py --workloads readme360 --rpw 1 --skew 2 --factors 0.8,0.9,1.0,1.1,1.2 2>&1 | grep -v amdgpu.ids | tee -a $OUT/wide_sweep.txt

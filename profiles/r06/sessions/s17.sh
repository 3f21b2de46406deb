#!/bin/bash
# r06 session 17: the register builds with 32-bit k-mer values from triples (one-word tables of any size, tables beyond the LDS-offset builds'
# range): whole -m gpu suite + fuzz, the one-word shapes before / after (RB_MULTI_ONE_WORD=0 keeps them on the register builds), guard of the refitted wide windows
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06q
mkdir -p $OUT
cd $R
timeout 3000 python3 -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
tail -6 $OUT/pytest_gpu.txt
export RB_TUNING_ENV=1
timeout 900 python3 profiles/multi_reads_sweep.py --workloads c1,c1_360,w1_64mib --rpw 0 --skew 2 --factors 0.8,0.9,1.0,1.1,1.2 2>&1 | grep -v amdgpu.ids | tee $OUT/one_word_triples.txt
timeout 2400 python3 profiles/phase_rule_check.py --reads 1000000 --points 4:250:37.7,4:360:37.7,4:250:24,4:360:24,4:200:36,4:300:24,3:200:13,3:250:30,3:360:30,4:250:12,4:360:12,4:250:46,4:360:46 --factors 0.6,0.7,0.8,0.9,1.0,1.1,1.2,1.35,1.5 > $OUT/phase_rule_check_wide.txt 2>&1
echo "guard exit $?" >> $OUT/phase_rule_check_wide.txt
grep -E "^[0-9]-word|outside|guard exit" $OUT/phase_rule_check_wide.txt | cut -c1-290

#!/bin/bash
# r06 session 16: parity of the one-word LDS-offset builds; one-word shapes, register builds against LDS-offset builds; guard over wide and one-word points
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06p
mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "several_reads" > $OUT/pytest_new.txt 2>&1
tail -3 $OUT/pytest_new.txt
export RB_TUNING_ENV=1
timeout 900 python3 profiles/multi_reads_sweep.py --workloads c1,c1_360 --rpw 0,1 --skew 2 --factors 0.6,0.7,0.8,0.9,1.0,1.1,1.2,1.35,1.5 2>&1 | grep -v amdgpu.ids | tee $OUT/one_word_sweep.txt
timeout 2400 python3 profiles/phase_rule_check.py --reads 1000000 --points 4:250:37.7,4:360:37.7,4:250:24,4:360:24,4:200:36,4:300:24,3:200:13,3:250:30,3:360:30,4:250:12,4:360:12,1:250:10,1:360:10,1:200:13,1:300:13,1:250:6,1:360:6 --factors 0.6,0.7,0.8,0.9,1.0,1.1,1.2,1.35,1.5,1.75 > $OUT/phase_rule_check_wide_one.txt 2>&1
echo "guard exit $?" >> $OUT/phase_rule_check_wide_one.txt
cut -c1-330 $OUT/phase_rule_check_wide_one.txt

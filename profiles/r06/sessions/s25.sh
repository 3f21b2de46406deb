#!/bin/bash
# r06 session 25: one-word blocks with 22-bit block numbers (the third one's top two bits in a per-lane spill word): one-word tables of 16-32 MiB
# -- the 64-bin filter of the reference's own test data among them -- move from the register builds to the LDS-offset builds and their equal cut.
# Parity (the LDS-offset test with 2.47 M, 2^21 and 2^22 - 2 blocks; the fuzz; the full-size narrow shapes), the config-1 geometry before /
# after (reads-per-wave 0 = register build), the guard over one-word points of 16-31 MiB.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06w
mkdir -p $OUT
cd $R
( time timeout 1500 python3 -m pytest tests -m gpu -x -q -k "several_reads or random_geometry or narrow or measurement_aids" ) > $OUT/pytest_22bit.txt 2>&1
tail -5 $OUT/pytest_22bit.txt
export RB_TUNING_ENV=1
timeout 900 python3 profiles/multi_reads_sweep.py --workloads c1,c1_360 --rpw 0,1 --skew 2 --factors 0.8,0.86,0.92,0.96,1.0,1.04,1.08,1.15,1.25 2>&1 | grep -v amdgpu.ids | tee $OUT/c1_22bit.txt | cut -c1-330
# two reads per wave once more, under the equal cut (negative result 13 was measured with 4 MiB slices)
timeout 600 python3 profiles/multi_reads_sweep.py --workloads deplete_target --rpw 2 --skew 2 --factors 0.9,1.0,1.1,1.2,1.3,1.45,1.6,1.8 2>&1 | grep -v amdgpu.ids | tee $OUT/r2_equal_cut.txt | cut -c1-330
unset RB_TUNING_ENV
timeout 1500 python3 profiles/phase_rule_check.py --points 1:250:19.8,1:360:19.8,1:200:24,1:250:26,1:300:28,1:360:31,1:250:31 > $OUT/guard_one_word_22bit.txt 2>&1
echo "guard (one-word, 16-31 MiB) exit $?"; grep -v "^      " $OUT/guard_one_word_22bit.txt | cut -c1-260
echo done

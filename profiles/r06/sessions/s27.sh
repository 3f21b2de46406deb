#!/bin/bash
# r06 session 27: v_lshl_add_u32 in the unpack of the packed block numbers (10 -> 7 VALU operations per slot and window; the window loop is
# VALU-bound at 78-82 % of the SIMDs' cycles): parity of the LDS-offset builds, the full-range test with the new 16-32 MiB one-word cases,
# K1 of every LDS-offset shape at the rule's window
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06q
mkdir -p $OUT
cd $R
( time timeout 1500 python3 -m pytest tests -m gpu -x -q -k "several_reads or random_geometry or narrow or phased_form_over or measurement_aids" ) > $OUT/pytest_lshl_add.txt 2>&1
tail -5 $OUT/pytest_lshl_add.txt
export RB_TUNING_ENV=1
timeout 900 python3 profiles/multi_reads_sweep.py --workloads deplete_target,targets3,deplete_target360,targets3_360,readme,readme360,c1,c1_360,mock_deplete,mock_deplete360 --rpw 1 --skew 2 --factors 0.92,0.96,1.0,1.04,1.08 2>&1 | grep -v amdgpu.ids | tee $OUT/lshl_add_all_shapes.txt | cut -c1-260
echo done

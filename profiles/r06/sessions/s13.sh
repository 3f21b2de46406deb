#!/bin/bash
# r06 session 13: parity of the templated multi builds (four and six tiles); the six-tile build on the 360 bp two-word shapes; XCD time skew on
# every narrow shape (which shapes gain from it?)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06m
mkdir -p $OUT
cd $R
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "several_reads" > $OUT/pytest_new.txt 2>&1
tail -3 $OUT/pytest_new.txt
timeout 1200 python3 profiles/multi_reads_sweep.py --workloads deplete_target360,targets3_360,mock_deplete,mock_deplete360 --rpw 0,1 --skew 0,2 --factors 0.7,0.8,0.9,1.0,1.1,1.2,1.35,1.5 2>&1 | grep -v amdgpu.ids | tee $OUT/six_tiles_sweep.txt
timeout 1200 python3 profiles/multi_reads_sweep.py --workloads readme,readme360,c1,c1_360,w1_64mib --rpw 0 --skew 0,2 --factors 0.8,0.9,1.0,1.1,1.2 2>&1 | grep -v amdgpu.ids | tee $OUT/skew_all_shapes.txt

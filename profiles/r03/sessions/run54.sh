#!/bin/bash
# round 3, GPU session 54: the phased builds compiled for more waves per SIMD (amdgpu_waves_per_eu minimum): one-word 250 bp 73 -> 56
# registers (8 waves), one-word 360 bp 93 -> 76 (6 waves), two-word 250 bp 86 -> 72 (7 waves) -- all without scratch.  A/B with window sweeps.
set -u
O=gpurun_out/r03
mkdir -p $O
T="325,400,500,600,700,850,1000,1200,1500"
variant() { local tag=$1; shift
  touch readbouncer_amd/csrc/rb_kernels.hip
  make -C readbouncer_amd/csrc -j4 KFLAGS="$*" > $O/build_$tag.log 2>&1 || { echo "build $tag failed"; tail -3 $O/build_$tag.log; return; }
  timeout 900 python profiles/r03/slice_size_sweep.py 1 250 8,10.5,20,32,64 21,22 $T > $O/occ_${tag}_w1_250.txt 2>&1
  timeout 900 python profiles/r03/slice_size_sweep.py 1 360 10.5,20,32,64 21,22 $T > $O/occ_${tag}_w1_360.txt 2>&1
  timeout 900 python profiles/r03/slice_size_sweep.py 2 250 10.5,19,32,64 21,22 $T > $O/occ_${tag}_w2_250.txt 2>&1
}
variant base
variant more -DRB_WAVES_0_1=7 -DRB_WAVES_0_3=6 -DRB_WAVES_1_1=6
ls $O | grep occ_

#!/bin/bash
# round 3, GPU session 58: the two-word 360 bp build at five waves per SIMD with 40 bytes of scratch (gather batch of one k-mer) against four waves
set -u
O=gpurun_out/r03
mkdir -p $O
T="325,400,500,600,700,850,1000,1200,1500"
variant() { local tag=$1; shift
  touch readbouncer_amd/csrc/rb_kernels.hip
  make -C readbouncer_amd/csrc -j4 KFLAGS="$*" > $O/build_$tag.log 2>&1 || { echo "build $tag failed"; tail -3 $O/build_$tag.log; return; }
  timeout 900 python profiles/r03/slice_size_sweep.py 2 360 8,10.5,19,32,48 21,22 $T > $O/occ13_${tag}.txt 2>&1
}
variant base
variant five -DRB_GATHER_KB3=1 -DRB_WAVES_1_3=5

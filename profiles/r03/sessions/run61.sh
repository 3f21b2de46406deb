#!/bin/bash
# round 3, GPU session 61: one-word blocks, reads of 385-512 k-mers: rounds of three tiles at eight waves per SIMD (63 registers) against two
# rounds of four tiles at four waves (111 registers)
set -u
O=gpurun_out/r03
mkdir -p $O
T="250,325,400,500,600,700,850,1000,1200,1500"
variant() { local tag=$1; shift
  touch readbouncer_amd/csrc/rb_kernels.hip
  make -C readbouncer_amd/csrc -j4 KFLAGS="$*" > $O/build_$tag.log 2>&1 || { echo "build $tag failed"; tail -3 $O/build_$tag.log; return; }
  timeout 900 python profiles/r03/slice_size_sweep.py 1 430,500 3,6,8,10.5,16,20,32,48,64 21,22 $T > $O/t3_${tag}.txt 2>&1
}
variant base
variant three -DRB_TILES_ROUNDS=3 -DRB_WAVES_0_2=7

#!/bin/bash
# round 3, GPU session 59: three-word blocks, 257-512 k-mers: a build without the fourth column at five waves per SIMD (94 registers) against the
# four-word build (114 registers, four waves) that has served them so far
set -u
O=gpurun_out/r03
mkdir -p $O
T="325,400,500,600,700,850,1000"
variant() { local tag=$1; shift
  touch readbouncer_amd/csrc/rb_kernels.hip
  make -C readbouncer_amd/csrc -j4 KFLAGS="$*" > $O/build_$tag.log 2>&1 || { echo "build $tag failed"; tail -3 $O/build_$tag.log; return; }
  python bench.py --workload targets3 --read-len 360 --steps 5 --warmup 2 --no-cpu-baseline --no-latency | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$tag targets3 360', round(d['value']/1e6,2), round(d['roofline']['avg_kernel_ms'],2), d['parity'])"
  timeout 900 python profiles/r03/slice_size_sweep.py 3 360,500 6,9,12,18,24,30,36 21,22 $T > $O/w3r_${tag}.txt 2>&1
}
variant base
variant five -DRB_WIDE3_ROUNDS=1 -DRB_WAVES_2_2_NW3=5

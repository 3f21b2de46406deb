#!/bin/bash
# builds the experimental libraries of the compacted-gather experiment (CPU, hipcc cross-compiles): readbouncer_amd/exp/libreadbouncer_amd_<tag>.so
# usage: bash profiles/r05/build_compact_variants.sh "g6c128:-DRB_COMPACT_G=6 -DRB_COMPACT_CAP=128 -DRB_COMPACT_G4=6" ...
set -e
R=$(cd $(dirname $0)/../.. && pwd)
C=$R/readbouncer_amd/csrc
mkdir -p $R/readbouncer_amd/exp
make -C $C -j8 > /dev/null
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  ( cd $C && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -ffp-contract=off --offload-arch=gfx950 -DRB_COMPACT=1 $flags -c rb_kernels.hip -o /tmp/rb_kernels_$tag.o \
    && /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 rb_host.o rb_live.o rb_pool.o /tmp/rb_kernels_$tag.o rb_engine.o rb_probe.o -o $R/readbouncer_amd/exp/libreadbouncer_amd_$tag.so && echo built $tag ) &
done
wait

#!/bin/bash
# r05 session 2: compacted-gather experiment -- parity of one experimental build, then the window sweep of the shipped library and four variants
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/s02
mkdir -p $OUT
cd $R
( RB_AMD_LIBRARY=$R/readbouncer_amd/exp/libreadbouncer_amd_g6c128.so timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "merged or packed or phased or narrow or geometr" ) > $OUT/parity_g6c128.txt 2>&1
tail -n 3 $OUT/parity_g6c128.txt | cut -c1-300
timeout 600 python3 profiles/compact_gather_sweep.py > $OUT/sweep_shipped.txt 2>&1
cat $OUT/sweep_shipped.txt | cut -c1-400
for tag in g6c128 g4c64 g8c128 g3c64; do
  RB_AMD_LIBRARY=$R/readbouncer_amd/exp/libreadbouncer_amd_$tag.so timeout 600 python3 profiles/compact_gather_sweep.py > $OUT/sweep_$tag.txt 2>&1
  cat $OUT/sweep_$tag.txt | cut -c1-400
done

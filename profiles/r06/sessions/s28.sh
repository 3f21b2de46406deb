#!/bin/bash
# r06 session 28: the -m gpu suite and smoke() once more on the final tree (three full-range cases were added after session 26's collection)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06p
mkdir -p $OUT
cd $R
( time timeout 1500 python3 -m pytest tests -m gpu -q ) > $OUT/pytest_gpu.txt 2>&1
tail -4 $OUT/pytest_gpu.txt
( time timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.txt 2>&1
tail -2 $OUT/smoke.txt
echo done

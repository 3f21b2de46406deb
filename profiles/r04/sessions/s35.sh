cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s35; mkdir -p $O
for i in 1 2 3; do ( timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "share_the_merged" ) 2>&1 | tail -n 3; done

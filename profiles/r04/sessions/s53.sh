cd $GRAFT_REPO_ROOT
( timeout 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "equal_length" ) 2>&1 | tail -n 12 | cut -c1-300

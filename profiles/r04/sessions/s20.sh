cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s20; mkdir -p $O
( timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "share_the_merged or merge or packed or thread_safe" ) > $O/pytest_merge.txt 2>&1; tail -n 15 $O/pytest_merge.txt | cut -c1-300
timeout 600 python3 profiles/engines_on_one_gpu.py --shapes readme,readme_unmerged > $O/engines_on_one_gpu.txt 2>&1; cat $O/engines_on_one_gpu.txt | cut -c1-150
timeout 600 python3 profiles/engines_on_one_gpu.py --shapes readme --forms device --k 1,2,4,6 >> $O/engines_on_one_gpu.txt 2>&1; tail -n 4 $O/engines_on_one_gpu.txt | cut -c1-150

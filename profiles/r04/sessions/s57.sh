cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s57; mkdir -p $O
timeout 900 python3 profiles/equal_slices_check.py --bins 192 --points 24:250,32:250,37.73:250,41.5:250,46:250,37.73:360,41.5:360,46:360,37.73:200,37.73:300,37.73:430,46:430 > $O/equal_slices_check_three_word.txt 2>&1; echo "exit $?" >> $O/equal_slices_check_three_word.txt; cut -c1-330 $O/equal_slices_check_three_word.txt
( timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -q -k "packed or merged or merge or phase or range or equal" ) 2>&1 | tail -n 3

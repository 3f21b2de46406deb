cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s47; mkdir -p $O
timeout 900 python3 profiles/equal_slices_check.py > $O/equal_slices_check.txt 2>&1; echo "exit $?" >> $O/equal_slices_check.txt; cut -c1-330 $O/equal_slices_check.txt
( timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_properties.py -m gpu -q -x -k "packed or merged or merge or measurement_aids or phase or range" ) > $O/pytest.txt 2>&1; tail -n 8 $O/pytest.txt | cut -c1-300

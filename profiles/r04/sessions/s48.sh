cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s48; mkdir -p $O
timeout 1200 python3 profiles/equal_slices_check.py --points 24:250,28:250,32:250,34:250,37.73:250,38.5:250,41.5:250,44:250,46:250,28:360,34:360,37.73:360,41.5:360,46:360,37.73:200,41.5:200,37.73:300,37.73:430,46:430 > $O/equal_slices_check.txt 2>&1; echo "exit $?" >> $O/equal_slices_check.txt; cut -c1-330 $O/equal_slices_check.txt

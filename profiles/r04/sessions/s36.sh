cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s36; mkdir -p $O
timeout 300 python3 profiles/soak_shared_merged.py 20 6 > $O/soak_shared_merged.txt 2>&1; tail -n 5 $O/soak_shared_merged.txt | cut -c1-400

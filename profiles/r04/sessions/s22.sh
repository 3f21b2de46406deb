cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s22; mkdir -p $O
nproc
timeout 900 python3 profiles/cli_readme250.py 64000000 - classifiers > $O/cli_classifiers_64M.txt 2>&1; grep -E "classifiers \(round" $O/cli_classifiers_64M.txt | cut -c1-250
timeout 900 python3 profiles/cli_readme250.py 16000000 - classifiers > $O/cli_classifiers_16M.txt 2>&1; grep -E "classifiers \(round" $O/cli_classifiers_16M.txt | cut -c1-250

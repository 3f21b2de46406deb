cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s55; mkdir -p $O
timeout 900 python3 profiles/cli_readme250.py 64000000 - quick > $O/cli_throughput_64M_reads.txt 2>&1; grep -E "defaults|classifiers" $O/cli_throughput_64M_reads.txt | cut -c1-260
timeout 900 python3 profiles/cli_readme250.py 16000000 - quick > $O/cli_throughput_16M_reads.txt 2>&1; grep -E "defaults|classifiers" $O/cli_throughput_16M_reads.txt | cut -c1-260

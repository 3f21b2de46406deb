cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s21; mkdir -p $O
timeout 900 python3 profiles/cli_readme250.py 16000000 > $O/cli_throughput.txt 2>&1; grep -E "defaults|serial|identical|parsers . class" $O/cli_throughput.txt | cut -c1-330
timeout 900 python3 profiles/cli_readme250.py 64000000 - quick > $O/cli_throughput_64M_reads.txt 2>&1; grep -E "defaults|classifiers" $O/cli_throughput_64M_reads.txt | cut -c1-330

cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s17; mkdir -p $O
timeout 600 python3 profiles/line_rate_probe.py > $O/line_rate_probe.txt 2>&1; cat $O/line_rate_probe.txt

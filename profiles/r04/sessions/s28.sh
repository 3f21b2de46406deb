cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s28; mkdir -p $O
timeout 600 python3 profiles/line_rate_probe.py 3,4,4.5,5,5.5,6,7,8,10,14,20 > $O/line_rate_probe_small.txt 2>&1; cat $O/line_rate_probe_small.txt

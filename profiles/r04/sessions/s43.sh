cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04s43; mkdir -p $O
timeout 700 bash profiles/collect_pmc_units.sh targets3 1000000 $O/targets3 > $O/targets3.txt 2>&1; cat $O/targets3.txt | cut -c1-160
grep -l "exceeds the capabilities" $O/targets3/*.log 2>/dev/null

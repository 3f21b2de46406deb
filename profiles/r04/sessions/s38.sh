cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s38; mkdir -p $O
timeout 2400 python3 profiles/phase_rule_check.py > $O/phase_rule_check_end_of_round.txt 2>&1; echo "exit code $?" >> $O/phase_rule_check_end_of_round.txt; grep -E "rule vs best|outside|exit code" $O/phase_rule_check_end_of_round.txt | cut -c1-40,150-330

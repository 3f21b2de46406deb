cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s25; mkdir -p $O
timeout 1500 python3 profiles/phase_rule_check.py --points 4:250:24,4:250:32,4:250:37.73,4:200:36,4:360:24,4:360:32,4:360:37.73,4:430:37.73 --factors 0.6,0.7,0.8,0.9,1.0,1.1,1.25 > $O/phase_rule_check_four_word.txt 2>&1; cut -c1-260 $O/phase_rule_check_four_word.txt

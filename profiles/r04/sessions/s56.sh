cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s56; mkdir -p $O
# three-word builds (192 bins): the rule leaves them at 4 MiB slices; what would N equal slices give?  ("rule" column = N forced by the env, "4 MiB slices" = as shipped)
for NS in 8 9; do echo "RB_PHASE_N_SLICES=$NS"; RB_PHASE_N_SLICES=$NS timeout 600 python3 profiles/equal_slices_check.py --bins 192 --points 37.73:250,41.5:250,37.73:360,41.5:360 2>&1 | grep -v amdgpu.ids | cut -c1-330; done > $O/equal_slices_three_word.txt 2>&1; cat $O/equal_slices_three_word.txt

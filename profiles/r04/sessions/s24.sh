cd $GRAFT_REPO_ROOT; O=gpurun_out/r04s24; mkdir -p $O
for mib in 0 8 4 2 1; do
  echo "host slice MiB $mib"; timeout 300 python3 profiles/engines_on_one_gpu.py --shapes readme,c4 --forms host --k 1,2 --host-slice-mib $mib 2>&1 | grep -v amdgpu.ids | cut -c1-110
done > $O/host_slices.txt 2>&1; cat $O/host_slices.txt

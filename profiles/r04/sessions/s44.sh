cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04s44; mkdir -p $O
for w in readme w1_64mib c3; do
  n=1000000; [ $w = c3 ] && n=2000000
  timeout 700 bash profiles/collect_pmc_units.sh $w $n $O/$w > $O/$w.txt 2>&1; echo "== $w"; cat $O/$w.txt | cut -c1-120
done

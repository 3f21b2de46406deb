cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04s42; mkdir -p $O
for w in targets3 readme c3; do
  n=1000000; [ $w = c3 ] && n=2000000
  bash profiles/collect_pmc_units.sh $w $n $O/$w > $O/$w.txt 2>&1; cat $O/$w.txt | cut -c1-160
done

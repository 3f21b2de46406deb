#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03
bash $R/profiles/collect_pmc.sh readme 1000000 $O/pmc_readme_new > /dev/null 2>&1
bash $R/profiles/collect_pmc.sh readme 1000000 $O/pmc_readme360_new "--read-len 360" > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv,glob,os
for d in ("pmc_readme_new","pmc_readme360_new"):
    print("==",d)
    for p in ("l2","ea","sq","fetch"):
        hits=glob.glob(os.path.join("gpurun_out/r03",d,p,"**","*counter_collection.csv"),recursive=True)
        if not hits: continue
        tot={}
        for r in csv.DictReader(open(hits[0])):
            if "ibf_count_max" not in r["Kernel_Name"]: continue
            a=tot.setdefault(r["Counter_Name"],[0,0.0]); a[0]+=1; a[1]+=float(r["Counter_Value"])
        for k,(n,v) in tot.items(): print("  %-24s dispatches %3d  sum/3 steps %.4g" % (k,n,v/3))
PY

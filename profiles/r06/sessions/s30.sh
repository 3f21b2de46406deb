#!/bin/bash
# r06 session 30: the counters of the wide legs once more on the final tree (their kernels did not change this round, their traffic.json entries
# were round 5's): c3 and c3np2 at 10 M reads per launch, GRCh38-F100k at 2 M, c2, c4 and the one-word 64 MiB table at 1 M
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06n
mkdir -p $OUT
cd $R
bash profiles/collect_pmc.sh c3 10000000 $OUT/pmc_c3 > $OUT/pmc_c3.log 2>&1
bash profiles/collect_pmc.sh c3np2 10000000 $OUT/pmc_c3np2 > $OUT/pmc_c3np2.log 2>&1
bash profiles/collect_pmc.sh grch38_f100k 2000000 $OUT/pmc_grch38_f100k > $OUT/pmc_grch38_f100k.log 2>&1
for w in c2 c4 w1_64mib; do bash profiles/collect_pmc.sh $w 1000000 $OUT/pmc_$w > $OUT/pmc_$w.log 2>&1; done
find $OUT -name "*.db" -delete; find $OUT -path "*pmc_*" -name "*kernel_trace.csv" -delete; find $OUT -path "*pmc_*" -name "*agent_info.csv" -delete
cd $R
mkdir -p $OUT/r06
RB_EVIDENCE_DATE=$(date +%F) python3 profiles/summarize.py $OUT $OUT/r06 2>&1 | tail -8
du -sm $OUT | cut -f1 | xargs echo "MiB:"
echo done

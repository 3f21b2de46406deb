OUT=$(pwd)/gpurun_out/r02a
bash profiles/collect_pmc.sh readme 1000000 $OUT/pmc_readme_r01 "--phased off --serial-table-mib 0" > /dev/null 2>&1
bash profiles/collect_pmc.sh c1 1000000 $OUT/pmc_c1_r01 "--phased off --serial-table-mib 0" > /dev/null 2>&1
ls $OUT/pmc_readme_r01 $OUT/pmc_c1_r01

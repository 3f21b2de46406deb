cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out/r04s49; mkdir -p $O; R=$GRAFT_REPO_ROOT
bash profiles/collect_pmc.sh readme 1000000 $O/pmc_readme > /dev/null 2>&1
bash profiles/collect_pmc.sh readme 1000000 $O/pmc_readme360 "--read-len 360" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_readme -- python3 $R/bench.py --workload readme --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/stats_readme.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_readme360 -- python3 $R/bench.py --workload readme --read-len 360 --steps 5 --warmup 2 --no-cpu-baseline --no-latency > $O/stats_readme360.log 2>&1
cd $R
ls $O/pmc_readme $O/pmc_readme360 | head -20
( time timeout 1500 python3 -m pytest tests -m gpu -q ) > $O/pytest_gpu.txt 2>&1; tail -n 6 $O/pytest_gpu.txt | cut -c1-300
( time timeout 900 python3 -m pytest tests -m gpuperf -q ) > $O/pytest_gpuperf.txt 2>&1; tail -n 6 $O/pytest_gpuperf.txt | cut -c1-300

#!/bin/bash
# A/B: default loads vs non-temporal table gathers (rebuilds the library on the GPU box)
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { python3 $R/bench.py --workload $1 --steps $2 --warmup 1 --reads $3 --no-cpu-baseline --no-latency 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$4', '$1', round(d['value']), round(d['roofline']['achieved']), round(d['roofline']['avg_kernel_ms'],2))"; }
run c2 10 1000000 default; run c3 3 2000000 default; run c2 10 1000000 default
make -C $R/readbouncer_amd/csrc -B KFLAGS=-DRB_NT_LOADS=1 > /dev/null 2>&1
run c2 10 1000000 nt; run c3 3 2000000 nt; run c2 10 1000000 nt
make -C $R/readbouncer_amd/csrc -B > /dev/null 2>&1

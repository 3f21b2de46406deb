#!/bin/bash
# Deep parity of the round-5 tree: long oracle legs on every bench workload -- raw maxima AND decisions, with the threshold-adjacent
# read stratum -- incl. config 3 at the reference's own sizing, the reference-default GRCh38 filter (W = 485) and the bit-packed merged
# tables at 250 / 360 / 600 / 1 500 bp; a 1000-seed x 2 N-rule fuzz soak; 200 k micro-batches through the latency kernel.
# (bench.py prints a bounded line since this round: the full result of each leg is its RB_BENCH_DETAIL sidecar.)
OUT=gpurun_out/r05deep; mkdir -p $OUT
run() { # tag secs args...
  local tag=$1 secs=$2; shift 2
  RB_BENCH_DETAIL=$OUT/$tag.detail.json timeout 1500 python3 bench.py "$@" --steps 3 --warmup 1 --cpu-seconds $secs --no-latency > $OUT/$tag.json 2> $OUT/$tag.err
  python3 -c "
import json
d=json.load(open('$OUT/$tag.detail.json')); p=d['parity']
print('$tag', round(d['value']), {k:p[k] for k in ('checked_reads','decision_mismatches','raw_max_mismatches','near_threshold_reads')}, d['config']['decisions'], 'cpu', round(d['cpu_baseline']['value']), [x.get('kernel','?')[14:-7]+(':'+x.get('phase_shape_name','') if x.get('phased') else '') for x in d['roofline']['plan']][:2])"
}
run readme250 30 --workload readme
run readme360 30 --workload readme --read-len 360
run readme600 20 --workload readme --read-len 600 --reads 500000
run readme1500 15 --workload readme --read-len 1500 --reads 200000
run targets3_250 20 --workload targets3
run targets3_360 20 --workload targets3 --read-len 360
run deplete_target_250 20 --workload deplete_target
run c1 30 --workload c1
run w1_64mib_250 20 --workload w1_64mib
run c2 30 --workload c2
run c4 45 --workload c4
run c3 45 --workload c3 --reads 2000000
run c3np2 45 --workload c3np2 --reads 2000000
run grch38_f100k 45 --workload grch38_f100k
RB_FUZZ_SEEDS=1000 timeout 2400 python3 -m pytest tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -3
timeout 600 python3 profiles/soak_split.py 2>&1 | tail -3

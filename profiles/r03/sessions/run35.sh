#!/bin/bash
# round 3, GPU session 35: CLI pipeline knobs on the README benchmark (2 M reads of 1 kbp, 4 GB FASTQ)
set -u
O=gpurun_out/r03
mkdir -p $O
nproc
for a in "--ingest-threads 4" "--ingest-threads 8" "--ingest-threads 12" "--ingest-threads 8 --classify-threads 3" "--ingest-threads 8 --segment-mb 128" "--ingest-threads 8 --classify-threads 4 --segment-mb 128"; do
  echo "== $a"
  RB_CLI_ARGS="$a" timeout 600 python profiles/cli_readme.py 2000000 2>&1 | grep "chunk_length" | sed 's/.*THROUGHPUT/THROUGHPUT/' | cut -c1-140
done

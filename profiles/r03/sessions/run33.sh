#!/bin/bash
# round 3, GPU session 33: reads per call from which the phased form pays on the larger tables of its new range
set -u
O=gpurun_out/r03
mkdir -p $O
timeout 600 python profiles/r03/phased_batch_size_large_tables.py 250 > $O/phased_batch_large.txt 2>&1
timeout 600 python profiles/r03/phased_batch_size_large_tables.py 360 >> $O/phased_batch_large.txt 2>&1
cat $O/phased_batch_large.txt

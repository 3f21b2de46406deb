#!/bin/bash
set -u
O=gpurun_out/r03
mkdir -p $O
hipcc -O3 --offload-arch=gfx950 profiles/gather_probe.hip -o /tmp/gather_probe || exit 1
timeout 600 /tmp/gather_probe phased_slack > $O/gather_phased_slack.txt 2>&1
cat $O/gather_phased_slack.txt

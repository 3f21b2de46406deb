#!/usr/bin/env python3
"""Means of the counters of the throughput count kernel in a profiles/collect_pmc.sh output directory."""
import csv, glob, sys
d = sys.argv[1]
for p in ("fetch", "l2", "ea", "sq"):
    hits = glob.glob("%s/%s/**/*counter_collection.csv" % (d, p), recursive=True)
    if not hits:
        continue
    acc = {}
    for r in csv.DictReader(open(hits[0])):
        if "ibf_count_max_kernel" not in r["Kernel_Name"]:
            continue
        a = acc.setdefault(r["Counter_Name"], [0, 0.0, 0.0])
        a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for k, (n, v, ms) in acc.items():
        print(p, k, n, "%.5g" % (v / n), "%.3f ms" % (ms / n))

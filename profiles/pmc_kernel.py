#!/usr/bin/env python3
"""Per-kernel means of the counters in ONE rocprofv3 counter_collection.csv: pmc_kernel.py <csv> <kernel substring>"""
import csv, sys
acc = {}
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] not in r["Kernel_Name"]:
        continue
    a = acc.setdefault((r["Kernel_Name"].split("(")[0][:60], r["Counter_Name"]), [0, 0.0, 0.0])
    a[0] += 1; a[1] += float(r["Counter_Value"]); a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
for (k, c), (n, v, ms) in sorted(acc.items()):
    print("%-60s %-28s n=%d mean=%.5g  %.3f ms" % (k, c, n, v / n, ms / n))

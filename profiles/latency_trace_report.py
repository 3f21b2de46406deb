#!/usr/bin/env python3
"""Per-call timeline from a rocprofv3 kernel trace of profiles/latency_trace.py: durations and gaps of the last 200 calls."""
import csv, glob, sys
import numpy as np
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ev = [(r["Kernel_Name"].split("(")[0][-40:], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# group into calls: a call ends with decide_kernel
calls, cur = [], []
for e in ev:
    cur.append(e)
    if "decide_kernel" in e[0]:
        calls.append(cur); cur = []
calls = [c for c in calls[-200:] if len(c) == len(calls[-1])]
names = [e[0] for e in calls[-1]]
dur = np.array([[e[2] - e[1] for e in c] for c in calls]) / 1e3
gap = np.array([[c[i + 1][1] - c[i][2] for i in range(len(c) - 1)] for c in calls]) / 1e3
span = np.array([c[-1][2] - c[0][1] for c in calls]) / 1e3
for i, nme in enumerate(names):
    print("%-42s dur p50 %.1f us%s" % (nme, np.median(dur[:, i]), "" if i == len(names) - 1 else "   gap to next p50 %.1f us" % np.median(gap[:, i])))
print("first kernel start -> last kernel end p50 %.1f us over %d calls" % (np.median(span), len(calls)))

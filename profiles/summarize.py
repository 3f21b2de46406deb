#!/usr/bin/env python3
"""Turns the scratch output of `profiles/collect_all.sh <tag>` (gpurun_out/<tag>) into the tracked evidence:
profiles/<round>/{bench_*.json, *_kernel_stats.csv, pmc_summary.csv, hbm_peak_raw.txt, latency_*.txt} and
profiles/traffic.json (HBM bytes per launch of the count kernel from the separate --pmc passes, corrected as
MI355X_MICROARCH.md prescribes: FETCH_SIZE is in KiB and gfx950 reports 128-byte requests at 64 bytes).
Usage: python3 profiles/summarize.py gpurun_out/r01e profiles/r01"""
import csv
import glob
import re
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
os.makedirs(dst, exist_ok=True)

for f in glob.glob(os.path.join(src, "bench_*.json")) + glob.glob(os.path.join(src, "latency_*.txt")):
    shutil.copy(f, dst)
if os.path.exists(os.path.join(src, "hbm_peak.txt")):
    shutil.copy(os.path.join(src, "hbm_peak.txt"), os.path.join(dst, "hbm_peak_raw.txt"))
for w in ("c2", "c3", "c3np2", "grch38_f100k", "c4", "c5", "readme", "readme360", "readme_phased", "readme360_phased", "c1", "w1_64mib", "w1_64mib_plain", "targets3", "targets3_apart", "deplete_target", "deplete_target_apart", "default"):
    hits = glob.glob(os.path.join(src, "stats_" + w, "**", "*kernel_stats.csv"), recursive=True)
    if hits:
        shutil.copy(hits[0], os.path.join(dst, w + "_kernel_stats.csv"))


# workloads whose step launches ONE kernel name several times and no other count kernel (the step count cannot be read off the
# dispatch counts then): dispatches per step
SAME_KERNEL_PER_STEP = {"targets3_apart": 3, "deplete_target_apart": 2}


# workloads of which only ONE build of the count kernel is the subject (the leg also launches the full-count build for comparison)
ONLY_KERNEL = {"c3_early": re.compile(r"ibf_count_max_kernel<[^>]*, true, true>")}  # <..., PH, EARLY = true>


def counter_means(path, per_step=1, only=None):
    """{counter: (dispatches, value per step, kernel ms per step)} of the count kernels (every form of K1).  A step launches
    one count kernel per filter -- different template instantiations, or the same one several times (the three one-word
    targets of the README shape): the per-step figure is the sum over all dispatches divided by the number of steps, and
    the number of steps is the smallest dispatch count of any kernel name."""
    tot, per_kernel = {}, {}
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            if "ibf_count_max" not in r["Kernel_Name"] or (only is not None and not only.search(r["Kernel_Name"])):
                continue
            a = tot.setdefault(r["Counter_Name"], [0, 0.0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            k = (r["Counter_Name"], r["Kernel_Name"])
            per_kernel[k] = per_kernel.get(k, 0) + 1
    out = {}
    for name, (n, v, ms) in tot.items():
        steps = min(c for (cn, _k), c in per_kernel.items() if cn == name) // per_step
        out[name] = (n, v / steps, ms / steps)
    return out


from readbouncer_amd import synth  # noqa: E402

rows, traffic = [], {}
MOCK = ["mock_deplete", "mock_t1", "mock_t2", "mock_t3"]
FILTERS = {"c2": ["c2"], "c3": ["c3"], "c3_early": ["c3"], "c3np2": ["c3np2"], "c4": ["c3", "zymo"], "grch38_f100k": ["grch38_f100k"], "c1": ["c1"],
           "readme": MOCK, "c1_r01": ["c1"], "readme_r01": MOCK, "readme_skew": MOCK, "readme360": MOCK, "readme360_six0": MOCK,
           "readme_phased": MOCK, "readme360_phased": MOCK, "w1_64mib": ["w1_64mib"], "w1_64mib_plain": ["w1_64mib"],
           "targets3": ["mock_t1", "mock_t2", "mock_t3"], "targets3_apart": ["mock_t1", "mock_t2", "mock_t3"],
           "deplete_target": ["mock_t3", "mock_t1"], "deplete_target_apart": ["mock_t3", "mock_t1"]}
READ_LEN = {"readme": 250, "readme_r01": 250, "readme_skew": 250, "readme_phased": 250, "w1_64mib": 250, "w1_64mib_plain": 250,
            "targets3": 250, "targets3_apart": 250, "deplete_target": 250, "deplete_target_apart": 250}
when = os.environ.get("RB_EVIDENCE_DATE", "")
for w in ("c2", "c3", "c3_early", "c3np2", "c4", "grch38_f100k", "c1", "readme", "c1_r01", "readme_r01", "readme_skew", "readme360", "readme360_six0", "readme_phased", "readme360_phased",
          "w1_64mib", "w1_64mib_plain", "targets3", "targets3_apart", "deplete_target", "deplete_target_apart"):
    d = os.path.join(src, "pmc_" + w)
    if not os.path.isdir(d):
        continue
    m = {}
    for p in ("fetch", "l2", "ea", "sq"):
        hits = glob.glob(os.path.join(d, p, "**", "*counter_collection.csv"), recursive=True)
        if not hits:
            continue
        for name, (n, v, ms) in counter_means(hits[0], SAME_KERNEL_PER_STEP.get(w, 1), ONLY_KERNEL.get(w)).items():
            rows.append((w, p, name, n, v, ms))
            m[name] = v
    if "FETCH_SIZE" not in m:
        continue
    reads = 500_000 if w == "grch38_f100k" else 1_000_000
    if os.path.exists(os.path.join(d, "reads.txt")):  # written by collect_pmc.sh: reads per launch of that pass
        reads = int(open(os.path.join(d, "reads.txt")).read().split()[0])
    wls = [synth.WORKLOADS[k] for k in FILTERS[w]]
    alg = synth.algorithmic_bytes_per_read(READ_LEN.get(w, 360), [(x["n_bins"], x["k"], x["h"]) for x in wls])
    hbm = m["FETCH_SIZE"] * 1024 * 2
    traffic[w] = {
        "reads_per_launch": reads,
        "FETCH_SIZE_KiB": m["FETCH_SIZE"],
        "hbm_bytes_per_launch": hbm,
        "hbm_bytes_per_read": hbm / reads,
        "TCC_EA0_RDREQ": m.get("TCC_EA0_RDREQ_sum"),
        "rdreq_x128B": m.get("TCC_EA0_RDREQ_sum", 0) * 128,
        "algorithmic_bytes_per_read": alg,
        "traffic_over_algorithmic": hbm / reads / alg,
        "l2_hit_rate": m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]) if "TCC_HIT_sum" in m else None,
        "valu_insts_per_read": m.get("SQ_INSTS_VALU", 0) / reads,
        "wait_any_frac_of_wave_cycles": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else None,
        "source": "rocprofv3 --pmc, separate passes (profiles/collect_pmc.sh), %s, %s" % (os.path.basename(dst.rstrip("/")), when),
        "note": "separate rocprofv3 --pmc passes (profiles/collect_pmc.sh); FETCH_SIZE x1024 x2 per MI355X_MICROARCH.md "
                "HBM section; cross-checked by TCC_EA0_RDREQ x 128 B",
    }
with open(os.path.join(dst, "pmc_summary.csv"), "w") as fh:
    fh.write("workload,pass,counter,dispatches,mean_value,mean_kernel_ms\n")
    for r in rows:
        fh.write("%s,%s,%s,%d,%g,%.3f\n" % r)
if traffic:
    tpath = os.path.join(root, "profiles", "traffic.json")
    merged = {}
    if os.path.exists(tpath):  # workloads not collected this time keep their earlier (dated) entry
        merged = json.load(open(tpath))
    merged.update(traffic)
    with open(tpath, "w") as fh:
        json.dump(merged, fh, indent=1)
for w, t in traffic.items():
    print(w, "traffic/algorithmic %.4f" % t["traffic_over_algorithmic"], "rdreq x128 / fetch %.4f" % (t["rdreq_x128B"] / t["hbm_bytes_per_launch"]))

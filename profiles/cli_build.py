#!/usr/bin/env python3
"""usage = "build" of the host CLI, end to end, at genome scale (SURVEY f.1's reason to exist; src/IBF/IBFBuild.cpp:421-521,
src/main/ibfbuild.hpp:21-59, src/main/main.cpp:286-344): a synthetic multi-record FASTA of GRCh38's shape -- 24 records, 3.1 Gbp, N runs
at both ends of every record (the last base of a record is an N: cutOutNNNs' end-of-sequence case, IBFBuild.cpp:121-125), a long N run
and a few short ones inside, soft-masked stretches -- goes through

    readbouncer_amd_cli --config build.toml     (parse -> cutOutNNNs -> sizing -> filter in HBM -> fragmenter -> insert kernel -> download -> .ibf)

at (i) fragment_size 380 000 (8 2xx bins: config 3's width) and (ii) the reference's default 100 000 (~31 000 bins).  Reported per run:
the process wall, the CLI's own BUILD_PHASES split, the size of the .ibf.  Checked per run: the bins of two whole records -- the smallest
one and the one with the longest inner N run -- against the ORACLE builder (oracle cutOutNNNs + fragmenter + insertKmer into a small
filter of the same block count: a k-mer's block number depends on noOfBlocks, k and h only, IBFBuild.cpp:404-413), column by column,
bit for bit, which also pins the global bin numbering across records (IBFBuild.cpp:165-204).

  python3 profiles/cli_build.py [--gbp 3.1] [--workdir /dev/shm/rb_cli_build] [--fragments 380000,100000]
"""
import argparse
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from readbouncer_amd import capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gbp", type=float, default=3.1)
ap.add_argument("--workdir", default="")
ap.add_argument("--fragments", default="380000,100000")
ap.add_argument("--keep", action="store_true")
args = ap.parse_args()

# GRCh38's chromosome lengths (Mbp), scaled to --gbp
CHROM_MBP = [248.9, 242.2, 198.3, 190.2, 181.5, 170.8, 159.3, 145.1, 138.4, 133.8, 135.1, 133.3, 114.4, 107.0, 102.0, 90.3, 83.3, 80.4,
             58.6, 64.4, 46.7, 50.8, 156.0, 57.2]
scale = args.gbp * 1e3 / sum(CHROM_MBP)
work = args.workdir
if not work:
    need = args.gbp * 1e9 * 1.02 + 9e9
    st = os.statvfs("/dev/shm")
    work = "/dev/shm/rb_cli_build" if st.f_bavail * st.f_frsize > need else "/tmp/rb_cli_build"
os.makedirs(work, exist_ok=True)
fasta = os.path.join(work, "genome.fasta")
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
LINE = 60

t0 = time.time()
rng = np.random.default_rng(38)
KEEP = (0, 20, 23)  # records checked against the oracle builder: the first, the smallest, the last
records = []        # (name, kept sequence or None, length, cleaned length)
with open(fasta, "wb") as fh:
    for ci, mbp in enumerate(CHROM_MBP):
        n = int(mbp * 1e6 * scale)
        seq = ACGT[rng.integers(0, 4, size=n, dtype=np.uint8)]
        runs = []
        lead = min(10_000, n // 50)
        trail = min(10_000, n // 50) if ci % 2 == 0 else 0          # even records END with N; odd ones with a base (the piece that runs to
        runs += [(0, lead), (n - trail, trail)]                     # the end loses its last base, IBFBuild.cpp:121-125)
        inner = min(int(3e6 * scale) + 1000, n // 10)               # centromere
        runs.append((n // 3 + ci * 1013 % 5000, inner))
        for _ in range(6):                                          # short runs: 1 .. 50 N
            runs.append((int(rng.integers(lead + 100, n - trail - 200)), int(rng.integers(1, 51))))
        for s, l in runs:
            seq[s:s + l] = ord("N")
        for _ in range(20):                                         # soft-masked repeats (lower case; an 'n' is NOT cut out: find("N"))
            s = int(rng.integers(0, n - 5000))
            seq[s:s + 3000] |= 0x20
        fh.write((">chr%d synthetic %d bp\n" % (ci + 1, n)).encode())
        whole = (n // LINE) * LINE
        body = np.empty((n // LINE, LINE + 1), dtype=np.uint8)
        body[:, :LINE] = seq[:whole].reshape(-1, LINE)
        body[:, LINE] = ord("\n")
        body.tofile(fh)
        if whole < n:
            fh.write(seq[whole:].tobytes() + b"\n")
        clen = int((seq != ord("N")).sum()) - (1 if seq[-1] != ord("N") else 0)
        records.append(("chr%d" % (ci + 1), seq.copy() if ci in KEEP else None, n, clen))
        del seq, body
for name, seq, n, clen in records:
    if seq is not None:
        assert len(po.cut_out_nnns(seq.tobytes().decode())) == clen, name  # the closed form above = the oracle's cutOutNNNs
print("setup %.1f s: %s, %.2f GB, 24 records, %.3f Gbp (%.3f Gbp after cutOutNNNs)"
      % (time.time() - t0, fasta, os.path.getsize(fasta) / 1e9, sum(r[2] for r in records) / 1e9, sum(r[3] for r in records) / 1e9), flush=True)

cli = os.path.join(ROOT, "readbouncer_amd", "readbouncer_amd_cli")
K, H = 13, 3


def check_record(ibf_path, info, rec_index, fragment, first_bin):
    """the bins of record rec_index (all of them up to 300, else both ends and every eighth in between) against the oracle builder"""
    name, seq, n, _clen = records[rec_index]
    clean = po.cut_out_nnns(seq.tobytes().decode())
    n_frag = len(clean) // fragment + 1
    W_small = (n_frag + 63) // 64
    n_blocks = info["n_blocks"]
    o = po.OracleIBF(n_frag, H, K, n_blocks * W_small * 64)
    assert o.n_blocks == n_blocks
    nxt = o.add_sequence(po.encode(clean), fragment, 0)
    assert nxt == n_frag, (nxt, n_frag)
    small = o.words()[:n_blocks * W_small].reshape(n_blocks, W_small)
    # the same bins of the file: words are block-major, W per block, behind the 8-byte header; the word columns that hold this record's
    # bins are read in ONE pass over the file
    W = info["bin_width"]
    mm = np.memmap(ibf_path, dtype=np.uint64, mode="r", offset=8, shape=(n_blocks * W,))
    c0, c1 = first_bin >> 6, (first_bin + n_frag - 1) >> 6
    sub = np.ascontiguousarray(mm.reshape(n_blocks, W)[:, c0:c1 + 1])
    del mm
    which = list(range(n_frag)) if n_frag <= 300 else sorted(set(list(range(100)) + list(range(n_frag - 100, n_frag)) + list(range(0, n_frag, 8))))
    bad, set_bits = 0, 0
    for j in which:
        b = first_bin + j
        col_big = (sub[:, (b >> 6) - c0] >> np.uint64(b & 63)) & np.uint64(1)
        col_small = (small[:, j >> 6] >> np.uint64(j & 63)) & np.uint64(1)
        set_bits += int(col_small.sum())
        if not np.array_equal(col_big, col_small):
            bad += 1
    assert set_bits > 0
    return n_frag, len(which), bad


for fragment in [int(x) for x in args.fragments.split(",")]:
    out_dir = os.path.join(work, "out_%d" % fragment)
    os.makedirs(out_dir, exist_ok=True)
    cfg = os.path.join(work, "build_%d.toml" % fragment)
    open(cfg, "w").write('usage = "build"\noutput_directory = "%s"\nlog_directory = "%s/logs"\n[IBF]\nkmer_size = %d\nfragment_size = %d\nthreads = 8\n'
                         'target_files = ["%s"]\n' % (out_dir, out_dir, K, fragment, fasta))
    t = time.time()
    p = subprocess.run([cli, "--config", cfg], capture_output=True, text=True)
    wall = time.time() - t
    phases = [l for l in p.stdout.splitlines() if l.startswith("BUILD_PHASES")]
    if p.returncode != 0 or not phases:
        print("fragment_size %d: CLI failed (%d)\n%s\n%s" % (fragment, p.returncode, p.stdout[-2000:], p.stderr[-2000:]))
        continue
    ibf = os.path.join(out_dir, "genome.ibf")
    kv = dict(x.split("=", 1) for x in phases[0].split()[1:])
    print("fragment_size %7d: process wall %6.2f s | %s | .ibf %.2f GB | %s"
          % (fragment, wall, "  ".join("%s %s" % (k, v) for k, v in kv.items() if k != "file"), os.path.getsize(ibf) / 1e9,
             " ".join(l.strip() for l in p.stderr.splitlines() if "IBF-build" in l or "bins were written" in l)), flush=True)
    # geometry of the file: the reference's sizing rule for the bin count the CLI reports (checked against the file's length)
    n_bins = int(kv["bins"])
    n_bits = capi.calculate_filter_size_bits(fragment, K, H, 0.01, n_bins)
    W = (n_bins + 63) // 64
    info = {"n_bins": n_bins, "bin_width": W, "n_blocks": n_bits // (64 * W)}
    assert os.path.getsize(ibf) == 8 + ((n_bits + 256 + 63) // 64) * 8, (os.path.getsize(ibf), n_bits)
    # global bin numbering: a record's first bin = the bins of all records before it (IBFBuild.cpp:165-204)
    first_bin, at = [], 0
    for name, seq, n, clen in records:
        first_bin.append(at)
        at += clen // fragment + 1
    print("   bins: %d expected from the records' cleaned lengths, %d reported by the CLI%s" % (at, n_bins, "" if at == n_bins else "   MISMATCH"), flush=True)
    for i in KEEP:
        t = time.time()
        n_frag, checked, bad = check_record(ibf, info, i, fragment, first_bin[i])
        print("   %-6s: %5d bins from bin %5d; %d of them against the oracle builder (cutOutNNNs + fragmenter + insertKmer): %d differ  (%.1f s)"
              % (records[i][0], n_frag, first_bin[i], checked, bad, time.time() - t), flush=True)
if not args.keep:
    import shutil
    shutil.rmtree(work, ignore_errors=True)

#!/usr/bin/env python3
"""Host CLI end to end on the narrow-filter workload it serves: the README benchmark shape (README.md:254-262:
1 deplete + 3 target filters, 122 / 43 / 29 / 49 bins at fragment_size 100000) loaded from .ibf files, N reads of 250 bp as
FASTQ in the page cache (/dev/shm), usage = "classify" with chunk_length 250, max_chunks 1 -> per-target FASTA + unclassified.fasta.
Reports, per knob setting: the CLI's own THROUGHPUT line (reads / wall of classify_reads, its wait/format breakdown) and the
wall time of the whole process (HIP start-up and filter loading included); the parser alone (--parse-stats); and checks that
the outputs of the parallel mapped-output run equal those of a one-thread run with positional writes, byte for byte.

  python3 profiles/cli_readme250.py [n_reads=16000000] [workdir=/dev/shm/rb_cli250]
"""
import hashlib
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from readbouncer_amd import capi, synth  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
L = 250
need = n_reads * (2 * L + 17) * 1.7  # FASTQ + outputs
work = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else "/dev/shm/rb_cli250"
if len(sys.argv) <= 2 or sys.argv[2] == "-":
    st = os.statvfs("/dev/shm")
    if st.f_bavail * st.f_frsize < need:  # a container with a small /dev/shm: the file system of /tmp (page cache after the first pass)
        work = "/tmp/rb_cli250"
os.makedirs(work, exist_ok=True)
t0 = time.time()
paths, refs = [], []
for i, key in enumerate(("mock_deplete", "mock_t1", "mock_t2", "mock_t3")):
    d, r = synth.build_device_filter(0, synth.WORKLOADS[key], fill_seed=11 + i, plant_seed=110 + i, n_segments=512)
    paths.append(os.path.join(work, key + ".ibf"))
    d.download().store(paths[-1])
    d.free()
    refs.append(r)
ref = np.concatenate(refs)
base_n = min(n_reads, 200_000)
buf, _, _ = synth.make_reads(5, base_n, L, ref)
# fixed-width records, filled with numpy: "@rRRR_IIIIII\n" + seq + "\n+\n" + qual + "\n"
rec_len = 13 + L + 3 + L + 1
block = np.empty((base_n, rec_len), dtype=np.uint8)
block[:, 0] = ord("@")
block[:, 1] = ord("r")
block[:, 5] = ord("_")
idx = np.arange(base_n)
for k in range(6):
    block[:, 6 + k] = ord("0") + (idx // 10 ** (5 - k)) % 10
block[:, 12] = ord("\n")
block[:, 13:13 + L] = buf.reshape(base_n, L)
block[:, 13 + L:13 + L + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
block[:, 13 + L + 3:13 + 2 * L + 3] = ord("I")
block[:, -1] = ord("\n")
fq = os.path.join(work, "reads.fastq")
with open(fq, "wb") as fh:
    rep = 0
    left = n_reads
    while left > 0:
        for k in range(3):
            block[:, 2 + k] = ord("0") + (rep // 10 ** (2 - k)) % 10
        m = min(left, base_n)
        block[:m].tofile(fh)
        left -= m
        rep += 1
print("setup %.1fs: 4 filters %.0f MB, FASTQ %.2f GB in %s (%d reads of %d bp, %d bytes per record)"
      % (time.time() - t0, sum(os.path.getsize(p) for p in paths) / 1e6, os.path.getsize(fq) / 1e9, work, n_reads, L, rec_len), flush=True)
cli = os.path.join(ROOT, "readbouncer_amd", "readbouncer_amd_cli")
for thr in (1, 4, 6, 8, 12):
    p = subprocess.run([cli, "--ingest-threads", str(thr), "--no-digest", "--parse-stats", fq], capture_output=True, text=True)
    print("parser alone, %2d threads: %s" % (thr, p.stdout.strip()), flush=True)
cfg = os.path.join(work, "c.toml")
out_dir = os.path.join(work, "out")
open(cfg, "w").write('usage = "classify"\noutput_directory = "%s"\nlog_directory = "%s/logs"\n[IBF]\n'
                     'deplete_files = ["%s"]\ntarget_files = ["%s", "%s", "%s"]\nread_files = ["%s"]\nchunk_length = %d\nmax_chunks = 1\n'
                     % (out_dir, out_dir, paths[0], paths[1], paths[2], paths[3], fq, L))


def digest_outputs():
    h = {}
    for name in sorted(os.listdir(out_dir)):
        f = os.path.join(out_dir, name)
        if os.path.isfile(f) and name.endswith(".fasta"):
            m = hashlib.md5()
            with open(f, "rb") as fh:
                for chunk in iter(lambda: fh.read(1 << 24), b""):
                    m.update(chunk)
            h[name] = (os.path.getsize(f), m.hexdigest())
    return h


def run(args, tag):
    subprocess.run(["rm", "-rf", out_dir])  # truncating last run's GB-sized outputs would be timed otherwise
    a = time.time()
    p = subprocess.run([cli, "--config", cfg] + args, capture_output=True, text=True)
    wall = time.time() - a
    lines = [l for l in p.stdout.splitlines() if l.startswith(("RESULT", "THROUGHPUT", "PHASES", "Real time", "CPU time"))]
    print("%-44s process wall %.2f s = %.2f M reads/s | %s %s" % (tag, wall, n_reads / wall / 1e6, " | ".join(lines), p.stderr.strip().replace("\n", " ")[-80:]), flush=True)


# the floor under the output side: how fast bytes enter the page cache of this directory (unclassified.fasta is ONE file)
probe = "/tmp/rb_pcw_probe"  # (/dev/shm is mounted noexec on the GPU boxes)
if subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(ROOT, "profiles", "pagecache_write_probe.cpp"), "-o", probe]).returncode == 0:
    print(subprocess.run([probe, work, "2700000000"], capture_output=True, text=True).stdout.strip(), flush=True)
if len(sys.argv) > 3 and sys.argv[3] == "classifiers":  # how many classifier threads: alternating, three rounds
    for i in range(3):
        for c in (4, 5, 6):
            run(["--classify-threads", str(c)], "%d classifiers (round %d)" % (c, i + 1))
    subprocess.run(["rm", "-rf", work])
    sys.exit(0)
if len(sys.argv) > 3 and sys.argv[3] == "quick":  # a long file, the defaults only: how the fixed costs of a run amortise
    for i in range(4):
        run([], "defaults (6 parsers, 4 classifiers, 32 MB segments)")
    run(["--classify-threads", "4"], "4 classifiers")
    run(["--ingest-threads", "1", "--classify-threads", "1", "--batch-reads", "1000000"], "serial: 1 parser 1 classifier")
    subprocess.run(["rm", "-rf", work])
    sys.exit(0)
run([], "warm-up (defaults)")
run([], "defaults (6 parsers, 4 classifiers, 32 MB segments)")
run([], "defaults")
run([], "defaults")
for ingest, cls, seg, batch in ((6, 4, 32, 65536), (8, 6, 32, 65536), (4, 6, 32, 65536), (6, 6, 16, 32768), (6, 6, 64, 65536), (6, 8, 32, 65536), (6, 2, 32, 65536)):
    run(["--ingest-threads", str(ingest), "--classify-threads", str(cls), "--segment-mb", str(seg), "--batch-reads", str(batch)],
        "parsers %d classifiers %d segment %d MB batch %d" % (ingest, cls, seg, batch))
run([], "defaults (outputs digested)")
par = digest_outputs()
run(["--mmap-output"], "defaults --mmap-output")
mm = digest_outputs()
run(["--ingest-threads", "1", "--classify-threads", "1", "--batch-reads", "1000000"], "serial: 1 parser 1 classifier")
ser = digest_outputs()
print("outputs byte-identical, mapped-output run against the serial run:", mm == ser and len(mm) == 4)
print("outputs:", {k: v[0] for k, v in par.items()})
print("outputs byte-identical, parallel run (positional writes) against the serial run:", par == ser and len(par) == 4)
subprocess.run(["rm", "-rf", work])

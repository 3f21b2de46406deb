#!/usr/bin/env python3
"""End-to-end throughput of the host CLI (usage="classify"): FASTQ on disk -> mmap ingest -> GPU -> FASTA outputs.
Builds a config-2-like filter file with the product library, writes N synthetic FASTQ reads, runs the CLI."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from readbouncer_amd import capi, synth  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/rb_cli_bench"
os.makedirs(work, exist_ok=True)
w = synth.WORKLOADS["c2"]
t0 = time.time()
d, ref = synth.build_device_filter(0, w, fill_seed=2, plant_seed=20)
ibf = os.path.join(work, "chr1_like.ibf")
d.download().store(ibf)
buf, offs, lens = synth.make_reads(5, min(n_reads, 200_000), 500, ref)
fq = os.path.join(work, "reads.fastq")
qual = b"I" * 500
with open(fq, "wb") as fh:
    base = buf.reshape(-1, 500)
    for rep in range((n_reads + len(base) - 1) // len(base)):
        rows = base[: min(len(base), n_reads - rep * len(base))]
        fh.write(b"".join(b"@r%d_%d\n%s\n+\n%s\n" % (rep, i, r.tobytes(), qual) for i, r in enumerate(rows)))
print("setup %.1fs: ibf %.2f GB, fastq %.2f GB" % (time.time() - t0, os.path.getsize(ibf) / 1e9, os.path.getsize(fq) / 1e9))
cfg = os.path.join(work, "c.toml")
open(cfg, "w").write('usage = "classify"\noutput_directory = "%s/out"\nlog_directory = "%s/out/logs"\n[IBF]\n'
                     'deplete_files = ["%s"]\nread_files = ["%s"]\nchunk_length = 360\nmax_chunks = 1\n' % (work, work, ibf, fq))
cli = os.path.join(ROOT, "readbouncer_amd", "readbouncer_amd_cli")
print(subprocess.run([cli, "--parse-stats", fq], capture_output=True, text=True).stdout.strip())
for batch, threads, cthreads in ((65536, 4, 2), (65536, 4, 2), (65536, 4, 1), (65536, 4, 3), (65536, 6, 2), (262144, 4, 2), (65536, 1, 1)):
    subprocess.run(["rm", "-rf", os.path.join(work, "out")])  # truncating last run's GB-sized outputs would be timed otherwise
    p = subprocess.run([cli, "--config", cfg, "--batch-reads", str(batch), "--ingest-threads", str(threads),
                        "--classify-threads", str(cthreads)], capture_output=True, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith(("RESULT", "THROUGHPUT"))]
    print("batch", batch, "parsers", threads, "classifiers", cthreads, " | ".join(lines), p.stderr.strip()[-100:])

#!/usr/bin/env python3
"""The reference's README benchmark (README.md:254-262: `classify` of real reads against 1 deplete + 3 target filters,
250 bp chunks, ~506 reads/s there on unstated hardware) as an END-TO-END run of this CLI: FASTQ on disk -> mmap ingest
-> chunk loop on the GPU -> per-target FASTA + unclassified.fasta.  Filters of the README shape (122 / 43 / 29 / 49 bins
at fragment_size 100000, 10-20 MB each) are written as .ibf files first; reads are 1 kbp with planted positives."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from readbouncer_amd import capi, synth  # noqa: E402

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
work = sys.argv[2] if len(sys.argv) > 2 else "/tmp/rb_cli_readme"
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
L = 1000
buf, offs, lens = synth.make_reads(5, min(n_reads, 100_000), L, ref)
fq = os.path.join(work, "reads.fastq")
qual = b"I" * L
with open(fq, "wb") as fh:
    base = buf.reshape(-1, L)
    for rep in range((n_reads + len(base) - 1) // len(base)):
        rows = base[: min(len(base), n_reads - rep * len(base))]
        fh.write(b"".join(b"@r%d_%d\n%s\n+\n%s\n" % (rep, i, r.tobytes(), qual) for i, r in enumerate(rows)))
print("setup %.1fs: 4 filters %.0f MB, fastq %.2f GB (%d reads of %d bp)" % (time.time() - t0, sum(os.path.getsize(p) for p in paths) / 1e6,
                                                                              os.path.getsize(fq) / 1e9, n_reads, L))
cli = os.path.join(ROOT, "readbouncer_amd", "readbouncer_amd_cli")
for chunk, max_chunks in ((250, 1), (250, 5)):  # RB_CLI_ARGS="--ingest-threads 8 ..." passes pipeline knobs through
    cfg = os.path.join(work, "c.toml")
    open(cfg, "w").write('usage = "classify"\noutput_directory = "%s/out"\nlog_directory = "%s/out/logs"\n[IBF]\n'
                         'deplete_files = ["%s"]\ntarget_files = ["%s", "%s", "%s"]\nread_files = ["%s"]\nchunk_length = %d\nmax_chunks = %d\n'
                         % (work, work, paths[0], paths[1], paths[2], paths[3], fq, chunk, max_chunks))
    for rep in range(2):
        subprocess.run(["rm", "-rf", os.path.join(work, "out")])
        a = time.time()
        p = subprocess.run([cli, "--config", cfg] + os.environ.get("RB_CLI_ARGS", "").split(), capture_output=True, text=True)
        wall = time.time() - a
        lines = [l for l in p.stdout.splitlines() if l.startswith(("RESULT", "THROUGHPUT"))]
        print("chunk_length %d max_chunks %d: process wall %.2f s (%.2f M reads/s incl. loading 4 filters) | %s %s"
              % (chunk, max_chunks, wall, n_reads / wall / 1e6, " | ".join(lines), p.stderr.strip()[-120:]))

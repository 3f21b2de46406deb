#!/usr/bin/env python3
"""SURVEY 8f.1 measurement: GPU IBF build (fragmenter + insert kernel + store) at config-2 and config-3 scale,
next to the CPU oracle builder on a sample, with a bit-exact comparison of the sampled bins."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from readbouncer_amd import capi  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def run(label, n_bins, frag, genome_len, k=13):
    rng = np.random.default_rng(1)
    bits = capi.calculate_filter_size_bits(frag, k, 3, 0.01, n_bins)
    t0 = time.perf_counter()
    genome = ACGT[rng.integers(0, 4, size=genome_len, dtype=np.uint8)]
    t_gen = time.perf_counter() - t0
    d = capi.DeviceIBF.create(0, n_bins, 3, k, bits)
    t0 = time.perf_counter()
    nxt = d.add_sequence(genome.tobytes(), frag, 0)
    t_build = time.perf_counter() - t0
    # CPU oracle on the first 4 fragments only (same bins), compare those bins' bits column by column
    sample_bins = 4
    o = po.OracleIBF(n_bins, 3, k, bits)
    t0 = time.perf_counter()
    o.add_sequence(po.encode(genome[: sample_bins * frag].tobytes()), frag, 0)
    t_cpu = time.perf_counter() - t0
    cpu_rate = sample_bins * frag / t_cpu
    host = d.download()
    W = host.info["bin_width"]
    g = host.words()[: host.info["n_blocks"] * W].reshape(-1, W)[:, 0]
    c = o.words()[: o.n_blocks * W].reshape(-1, W)[:, 0]
    mask = np.uint64((1 << (sample_bins - 1)) - 1)  # bins 0..2 are complete in both (bin 3's fragment is cut in the sample)
    same = bool(np.array_equal(g & mask, c & mask))
    print("%s: %d bins, %.2f GB filter, %.1f Mbp genome -> %d bins used | GPU build %.2f s = %.0f Mbp/s (incl. H2D of the "
          "genome) | CPU oracle %.2f Mbp/s single thread | sampled bins bit-identical: %s | (genome synthesis %.1f s)"
          % (label, n_bins, bits / 8e9, genome_len / 1e6, nxt, t_build, genome_len / 1e6 / t_build, cpu_rate / 1e6, same, t_gen))
    return same


ok = run("config2-scale", 1024, 243000, 248_000_000)
ok &= run("config3-scale", 8192, 378000, 3_090_000_000) if len(sys.argv) > 1 and sys.argv[1] == "full" else True
sys.exit(0 if ok else 1)

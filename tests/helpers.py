"""Shared test helpers (test infrastructure: may use the oracle)."""
import os

import numpy as np

from oracle import pyoracle as po


def read_fasta(path):
    """Minimal FASTA reader: [(id, sequence)], multi-line records, blank lines ignored."""
    recs, name, parts = [], None, []
    with open(path) as fh:
        for line in fh:
            line = line.rstrip("\r\n")
            if not line:
                continue
            if line[0] == ">":
                if name is not None:
                    recs.append((name, "".join(parts)))
                name, parts = line[1:], []
            else:
                parts.append(line)
    if name is not None:
        recs.append((name, "".join(parts)))
    return recs


def read_fastq(path):
    recs = []
    with open(path) as fh:
        lines = [l.rstrip("\r\n") for l in fh]
    for i in range(0, len(lines) - 3, 4):
        recs.append((lines[i][1:], lines[i + 1]))
    return recs


def build_filter_like_reference(seqs, k=13, fragment_length=100000, h=3, max_fp=0.01, overlap=1500):
    """IBF::create_filter (IBFBuild.cpp:421-521) through the oracle.
    seqs: list of raw sequence strings of ONE reference file."""
    cleaned = []
    n_bins = 0
    for s in seqs:
        if len(s) < k:  # invalidSeqs, IBFBuild.cpp:70-74
            continue
        c = po.cut_out_nnns(s)
        cleaned.append(c)
        n_bins += len(c) // fragment_length + 1  # IBFBuild.cpp:90
    bits = po.calculate_filter_size_bits(fragment_length, k, h, max_fp, n_bins)
    f = po.OracleIBF(n_bins, h, k, bits)
    binid = 0
    for c in cleaned:
        binid = f.add_sequence(po.encode(c), fragment_length, binid, overlap)
    return f


def random_dna(rng, n, with_n=0.0):
    arr = rng.integers(0, 4, size=n)
    s = np.array(list("ACGT"))[arr]
    if with_n > 0:
        mask = rng.random(n) < with_n
        s[mask] = "N"
    return "".join(s)


def mutate(rng, s, rate):
    """i.i.d. substitutions at the given rate."""
    a = np.frombuffer(s.encode(), dtype=np.uint8).copy()
    m = rng.random(len(a)) < rate
    a[m] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(m.sum()))]
    return a.tobytes().decode()


def pack_reads(reads):
    """list[str] -> (uint8 concat, uint64 offsets, uint32 lens)"""
    lens = np.array([len(r) for r in reads], dtype=np.uint32)
    offs = np.zeros(len(reads), dtype=np.uint64)
    if len(reads):
        offs[1:] = np.cumsum(lens[:-1], dtype=np.uint64)
    buf = np.frombuffer("".join(reads).encode(), dtype=np.uint8).copy() if len(reads) else np.zeros(0, np.uint8)
    if buf.size == 0:
        buf = np.zeros(1, np.uint8)
    return buf, offs, lens

"""Seeded differential fuzzing of the HIP path against the oracle over random filter geometries (bins, blocks,
k, h), read lengths and alphabets -- the corner cases nobody thought of listing."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import pyoracle as po
from readbouncer_amd import capi
from tests import helpers as H


# RB_FUZZ_SEEDS=<n> widens the sweep for an offline soak (profiles/r01/deep_parity.txt)
# n_rule: what the reverse strand holds for an N of the read -- 3 (T; ModComplementDna over a Dna5String, the default)
# or 4 (N).  Both candidates of that recalled SeqAn fact are fuzzed, kernels and oracle under the same rule.
@pytest.mark.parametrize("n_rule", [3, 4])
@pytest.mark.parametrize("seed", range(int(os.environ.get("RB_FUZZ_SEEDS", "24"))))
def test_random_geometry(seed, n_rule):
    prev = po.set_revcomp_of_n(n_rule)
    try:
        _random_geometry(seed, n_rule)
    finally:
        po.set_revcomp_of_n(prev)


def _random_geometry(seed, n_rule):
    rng = np.random.default_rng(1000 + seed)
    n_bins = int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129, int(rng.integers(1, 9000)), int(rng.integers(1, 700))]))
    k = int(rng.choice([3, 5, 8, 11, 13, 15, 19, 27, 28, 32, int(rng.integers(3, 33))]))
    h = int(rng.choice([1, 2, 3, 3, 3, 4, 5, 8]))
    W = (n_bins + 63) // 64
    n_blocks = int(rng.choice([1, 2, 3, 64, 1000, 1024, int(rng.integers(1, 4000))]))
    if W * n_blocks > 4_000_000:
        n_blocks = max(1, 4_000_000 // W)
    n_bits = n_blocks * W * 64 + int(rng.integers(0, 64 * W))
    d = capi.DeviceIBF.create(0, n_bins, h, k, n_bits)
    if rng.random() < 0.5:
        d.fill_synth(int(rng.integers(1, 1 << 30)))
    ref = H.random_dna(rng, 5000, with_n=0.01)
    frag = int(rng.integers(max(k, 20), 2000))
    nfrag_bins = min(n_bins, 5000 // frag + 2)
    starts = (np.arange(nfrag_bins) * frag).astype(np.uint64)
    ends = np.minimum(starts + np.uint64(frag + k - 1), np.uint64(5000)).astype(np.uint64)
    ok = starts < 5000
    d.insert(ref, starts[ok], ends[ok], rng.integers(0, n_bins, size=int(ok.sum())).astype(np.uint64))
    host = d.download()
    o = po.OracleIBF.wrap(n_bins, h, k, n_bits, host.words())
    reads = []
    for i in range(60):
        L = int(rng.choice([0, 1, k - 1, k, k + 1, 63 + k, 64 + k, 65 + k, int(rng.integers(1, 700)), int(rng.integers(1, 1300))]))
        L = max(0, L)
        if i % 3 == 0:
            r = H.random_dna(rng, L, with_n=0.05)
        else:
            s = int(rng.integers(0, max(1, 5000 - L)))
            r = H.mutate(rng, ref[s:s + L], float(rng.choice([0.0, 0.05, 0.2])))
        if i % 7 == 0:
            r = r.lower()
        reads.append(r)
    buf, offs, lens = H.pack_reads(reads)
    exp_max = po.batch_raw_max(o, buf, offs, lens, 4)
    r_err = float(rng.choice([0.1, 0.05, 0.14]))
    exp_dec, exp_st = po.batch_check_unblock([o], [], buf, offs, lens, r=r_err, n_threads=4)
    eng = capi.Engine(0, [d], [])
    eng.set_revcomp_of_n(n_rule)
    for split in (2048, 0):
        eng.set_split_threshold(split)
        mc, _, dec, st = eng.classify(buf, offs, lens, error_rate=r_err)
        assert np.array_equal(mc[:, 0], exp_max), (n_bins, k, h, n_blocks, split)
        assert np.array_equal(dec, exp_dec) and np.array_equal(st, exp_st), (n_bins, k, h, n_blocks, split)
    # the opt-in early-decision mode on the throughput form (a call that asks for decisions only): same decisions, whatever the geometry
    # (builds without an early twin -- h != 3 -- count on as before)
    eng.set_early_decision(1)
    dec, st = eng.decide(buf, offs, lens, error_rate=r_err)
    eng.set_early_decision(0)
    assert np.array_equal(dec, exp_dec) and np.array_equal(st, exp_st), (n_bins, k, h, n_blocks, "early")
    # throughput form with clock-phased gathers (forced; serves blocks of one and two words with three hash functions)
    eng.set_phased(0, 1 << 40, int(rng.choice([0, 100, 1500])), int(rng.choice([1, 30])), 1)
    eng.set_phase_slices(1, int(rng.choice([2, 5, 8, 32])))  # as many slices as that, however small the table
    mc, _, dec, st = eng.classify(buf, offs, lens, error_rate=r_err)
    assert np.array_equal(mc[:, 0], exp_max), (n_bins, k, h, n_blocks, "phased")
    assert np.array_equal(dec, exp_dec) and np.array_equal(st, exp_st), (n_bins, k, h, n_blocks, "phased")
    eng.set_phased(0, 0, 0, 0, 0)
    # latency form with several workgroups per read (wide filters), on the batch and on a few reads of it (more parts)
    eng.set_split_threshold(2048)
    eng.set_split_parts(int(rng.choice([2, 3, 8, 16])), int(rng.choice([1, 2, 4, 8])))
    for n_sub in (len(reads), 7, 2):
        mc = eng.classify(buf, offs[:n_sub], lens[:n_sub], error_rate=r_err)[0]
        assert np.array_equal(mc[:, 0], exp_max[:n_sub]), (n_bins, k, h, n_blocks, "parts", n_sub)

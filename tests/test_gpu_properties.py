"""Size-independent properties at BASELINE scale, where the CPU oracle is too slow to check every read: strand symmetry,
batch-partition invariance, order invariance, kernel-form invariance -- on config 2 (0.41 GB filter, 10^5 reads) and on
config 4 (8 GiB deplete + 600-bin target filter, 10^6 reads, full check_unblock).  A sample of each batch is still
compared with the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import pyoracle as po
from readbouncer_amd import capi, synth


@pytest.fixture(scope="module")
def c2():
    w = synth.WORKLOADS["c2"]
    d, ref = synth.build_device_filter(0, w, fill_seed=2, plant_seed=20)
    buf, offs, lens = synth.make_reads(77, 100_000, 360, ref)
    eng = capi.Engine(0, [d], [])
    base = eng.classify(buf, offs, lens)
    return d, eng, buf, offs, lens, base


def test_strand_symmetry(c2):
    # max over bins of max(fwd, rev) is invariant under reverse-complementing the read (IBFClassify.cpp:149-162)
    d, eng, buf, offs, lens, base = c2
    comp = np.zeros(256, dtype=np.uint8)
    comp[np.frombuffer(b"ACGT", dtype=np.uint8)] = np.frombuffer(b"TGCA", dtype=np.uint8)
    rc = comp[buf.reshape(-1, 360)[:, ::-1]].reshape(-1).copy()
    got = eng.classify(rc, offs, lens)
    assert np.array_equal(got[0], base[0]) and np.array_equal(got[2], base[2])
    assert base[2].sum() > 40_000  # about half of the reads are planted positives


def test_partition_and_order_invariance(c2):
    d, eng, buf, offs, lens, base = c2
    n = len(lens)
    rng = np.random.default_rng(5)
    perm = rng.permutation(n)
    got = eng.classify(buf, offs[perm], lens[perm])  # same buffer, shuffled work order
    assert np.array_equal(got[0], base[0][perm]) and np.array_equal(got[2], base[2][perm])
    pos = 0
    for size in (1, 7, 63, 64, 65, 2047, 2048, 2049, 30000):  # ragged batches across the latency/throughput switch
        sl = slice(pos, pos + size)
        g = eng.classify(buf, offs[sl], lens[sl])
        assert np.array_equal(g[0], base[0][sl]) and np.array_equal(g[2], base[2][sl]), size
        pos += size


def test_kernel_forms_and_load_policies_agree(c2):
    d, eng, buf, offs, lens, base = c2
    sl = slice(0, 1500)
    for split, nt in ((0, 512 << 20), (2048, 512 << 20), (0, 0), (2048, 0)):
        eng.set_split_threshold(split)
        eng.set_nt_threshold(nt)
        g = eng.classify(buf, offs[sl], lens[sl])
        assert np.array_equal(g[0], base[0][sl]) and np.array_equal(g[2], base[2][sl]), (split, nt)
    eng.set_split_threshold(2048)
    eng.set_nt_threshold(512 << 20)


def test_sample_against_oracle(c2):
    d, eng, buf, offs, lens, base = c2
    host = d.download()
    o = po.OracleIBF.wrap(host.info["n_bins"], 3, 13, host.info["n_bits"], host.words())
    idx = np.random.default_rng(9).choice(len(lens), size=3000, replace=False)
    exp_max = po.batch_raw_max(o, buf, offs[idx], lens[idx], 8)
    exp_dec, _ = po.batch_check_unblock([o], [], buf, offs[idx], lens[idx], n_threads=8)
    assert np.array_equal(base[0][idx, 0], exp_max) and np.array_equal(base[2][idx], exp_dec)


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs 3/4 at full filter size: 8 GiB GRCh38-scale deplete filter (8192 bins) + 600-bin target filter,
# 10^6 reads through the whole check_unblock decision.
@pytest.fixture(scope="module")
def c4():
    dep, ref_d = synth.build_device_filter(0, synth.WORKLOADS["c3"], fill_seed=4, plant_seed=40)
    tgt, ref_t = synth.build_device_filter(0, synth.WORKLOADS["zymo"], fill_seed=6, plant_seed=60)
    buf, offs, lens = synth.make_reads(78, 1_000_000, 360, np.concatenate([ref_d, ref_t]))
    eng = capi.Engine(0, [dep], [tgt])
    base = eng.classify(buf, offs, lens)
    return dep, tgt, eng, buf, offs, lens, base


def test_c4_strand_symmetry_and_decision_mix(c4):
    dep, tgt, eng, buf, offs, lens, base = c4
    comp = np.zeros(256, dtype=np.uint8)
    comp[np.frombuffer(b"ACGT", dtype=np.uint8)] = np.frombuffer(b"TGCA", dtype=np.uint8)
    rc = comp[buf.reshape(-1, 360)[:, ::-1]].reshape(-1).copy()
    got = eng.classify(rc, offs, lens)
    assert np.array_equal(got[0], base[0]) and np.array_equal(got[1], base[1]) and np.array_equal(got[2], base[2])
    counts = np.bincount(base[2], minlength=3)
    assert counts.min() > 100_000  # wait / unblock / stop_receiving all occur (reads from both references + random ones)
    assert (base[3] == capi.RB_OK).all()


def test_c4_partition_forms_and_host_paths(c4):
    """the same reads as one 10^6 batch (throughput kernels, sliced PCIe copies), as micro-batches (latency kernels: one
    mixed-geometry launch, several workgroups per read on the 8 GiB filter) and with those features switched off"""
    dep, tgt, eng, buf, offs, lens, base = c4
    pos = 0
    for size in (1, 5, 14, 64, 200, 2048, 2049, 70_000):
        sl = slice(pos, pos + size)
        g = eng.classify(buf, offs[sl], lens[sl])
        assert np.array_equal(g[0], base[0][sl]) and np.array_equal(g[2], base[2][sl]) and np.array_equal(g[1], base[1][sl]), size
        pos += size
    sl = slice(500_000, 500_300)
    for parts, split in ((1, 2048), (16, 2048), (8, 0)):
        eng.set_split_parts(parts, 4)
        eng.set_split_threshold(split)
        g = eng.classify(buf, offs[sl], lens[sl])
        assert np.array_equal(g[0], base[0][sl]) and np.array_equal(g[2], base[2][sl]), (parts, split)
    eng.set_split_parts(8, 4)
    eng.set_split_threshold(2048)
    eng.set_host_slice_bytes(0)  # the whole 360 MB batch in one copy
    g = eng.classify(buf, offs, lens)
    eng.set_host_slice_bytes(32 << 20)
    assert np.array_equal(g[0], base[0]) and np.array_equal(g[2], base[2])


def test_c4_sample_against_oracle(c4):
    dep, tgt, eng, buf, offs, lens, base = c4
    views, keep = [], []
    for d in (dep, tgt):
        host = d.download()
        keep.append(host)
        views.append(po.OracleIBF.wrap(host.info["n_bins"], 3, 13, host.info["n_bits"], host.words()))
    idx = np.random.default_rng(10).choice(len(lens), size=2000, replace=False)
    exp_dec, exp_st = po.batch_check_unblock(views[:1], views[1:], buf, offs[idx], lens[idx], n_threads=8)
    assert np.array_equal(base[2][idx], exp_dec) and np.array_equal(base[3][idx], exp_st)
    for f, o in enumerate(views):
        assert np.array_equal(base[0][idx, f], po.batch_raw_max(o, buf, offs[idx], lens[idx], 8))


@pytest.fixture(scope="module")
def narrow():
    """the README benchmark shape (README.md:254-262): one two-word deplete filter (122 bins, 20 MB) and three one-word
    target filters (43 / 29 / 49 bins, 10 MB each) at their real sizes -- the geometry the phased kernels are planned for"""
    deplete, target, refs = [], [], []
    for i, key in enumerate(("mock_deplete", "mock_t1", "mock_t2", "mock_t3")):
        f, r = synth.build_device_filter(0, synth.WORKLOADS[key], fill_seed=11 + i, plant_seed=110 + i, n_segments=512)
        (deplete if i == 0 else target).append(f)
        refs.append(r)
    ref = np.concatenate(refs)
    eng = capi.Engine(0, deplete, target)
    return deplete, target, ref, eng


def test_narrow_filters_kernel_forms_agree_at_full_size(narrow):
    """Plain gathers, clock-phased gathers at several window lengths, both-strands tiles (250 bp) and per-strand tiles
    (360 bp, ragged lengths), filters overlapped or taking turns: the same maxima and decisions for 10^5 reads each;
    strand symmetry on top; a sample against the oracle."""
    deplete, target, ref, eng = narrow
    rng = np.random.default_rng(8)
    for read_len in (250, 360):
        buf, offs, lens = synth.make_reads(90 + read_len, 100_000, read_len, ref)
        eng.set_phased(0, 0, 0, 0, 0)           # plain kernels, one-word filters on the round-1 tiles
        eng.set_serial_table_bytes(0)
        base = eng.classify(buf, offs, lens)
        assert len(set(base[2].tolist())) == 3
        eng.set_serial_table_bytes(64 << 20)
        for args in ((6 << 20, 32 << 20, 0, 0, 4096), (6 << 20, 32 << 20, 120, 0, 1024), (6 << 20, 32 << 20, 1500, 0, 1024),
                     (0, 0, 450, 0, 1024)):     # the last one: short-read tiles without windows
            eng.set_phased(*args)
            got = eng.classify(buf, offs, lens)
            assert np.array_equal(got[0], base[0]) and np.array_equal(got[2], base[2]) and np.array_equal(got[1], base[1]), args
        # ragged lengths around the 256-k-mer switch, in one batch (both paths of one launch)
        eng.set_phased(6 << 20, 32 << 20, 0, 0, 1024)
        m = 40_000
        rl = rng.integers(200, 330, size=m).astype(np.uint32)
        ro = (np.arange(m, dtype=np.uint64) * np.uint64(read_len))
        rl = np.minimum(rl, read_len).astype(np.uint32)
        g1 = eng.classify(buf, ro, rl)
        eng.set_phased(0, 0, 0, 0, 0)
        g0 = eng.classify(buf, ro, rl)
        assert np.array_equal(g1[0], g0[0]) and np.array_equal(g1[2], g0[2])
        eng.set_phased(6 << 20, 32 << 20, 0, 0, 4096)
        # strand symmetry through the phased kernels
        comp = np.zeros(256, dtype=np.uint8)
        comp[np.frombuffer(b"ACGT", dtype=np.uint8)] = np.frombuffer(b"TGCA", dtype=np.uint8)
        rc = comp[buf.reshape(-1, read_len)[:, ::-1]].reshape(-1).copy()
        got = eng.classify(rc, offs, lens)
        assert np.array_equal(got[0], base[0]) and np.array_equal(got[2], base[2])
        # oracle sample
        keep = [f.download() for f in deplete + target]
        views = [po.OracleIBF.wrap(h.info["n_bins"], 3, 13, h.info["n_bits"], h.words()) for h in keep]
        n = 3000
        exp_dec, exp_st = po.batch_check_unblock(views[:1], views[1:], buf, offs[:n], lens[:n], n_threads=8)
        assert np.array_equal(base[2][:n], exp_dec) and np.array_equal(base[3][:n], exp_st)

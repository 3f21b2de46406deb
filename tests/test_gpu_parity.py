"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs.  Bit-exact (integer/bit work).  Run on the MI355X box with `-m gpu`."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import pyoracle as po
from readbouncer_amd import capi
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

READ_354 = "AAAAAAAACCCCCCCCCGAGAGAGGAGAGAGGAGAGAGAGAGCCCCAAAAGAGAGGAGA" * 6


def oracle_view(dibf):
    host = dibf.download()
    i = host.info
    return po.OracleIBF.wrap(i["n_bins"], i["n_hash"], i["kmer_size"], i["n_bits"], host.words()), host


def make_reads(rng, ref, n, lo=40, hi=500, err=0.08, n_frac=0.1):
    reads = []
    for i in range(n):
        L = int(rng.integers(lo, hi))
        kind = i % 4
        if kind == 0 or len(ref) <= L:
            r = H.random_dna(rng, L)
        else:
            s = int(rng.integers(0, len(ref) - L))
            r = H.mutate(rng, ref[s:s + L], err if kind != 3 else 0.0)
        if rng.random() < n_frac and L > 5:
            p = int(rng.integers(0, L - 1))
            r = r[:p] + "N" + r[p + 1:]
        if kind == 2 and rng.random() < 0.5:  # reverse-complement positives
            r = "".join("ACGTN"[x] for x in po.revcomp(po.encode(r)))
        reads.append(r)
    return reads


GEOMETRIES = [
    # (n_bins, n_blocks, k, h)
    (40, 4099, 13, 3),      # W=1 (B<64), odd block count
    (64, 4096, 13, 3),      # W=1, power-of-two block count
    (100, 2053, 13, 3),     # W=2, padding bits in the last word
    (130, 1021, 15, 3),     # W=3 -> 4 lanes per block, one idle
    (600, 997, 13, 3),      # W=10 (mock-community target), 16 lanes per block, 6 idle
    (1024, 1024, 13, 3),    # W=16: config 2 geometry
    (1030, 509, 13, 3),     # W=17 -> 32 lanes per block
    (4096, 257, 13, 3),     # W=64: one whole wave per block
    (8192, 128, 13, 3),     # W=128: config 3 geometry, 16 B per lane
    (8300, 131, 13, 3),     # W=130: two column slices, 16 B per lane, ragged tail
    (4200, 211, 13, 3),     # W=66 even
    (4300, 199, 13, 3),     # W=68
    (4160 + 1, 173, 13, 3), # W=66 (4161 bins -> 66 words)
    (5000, 163, 13, 3),     # W=79 odd: 8-byte path with two slices
    (700, 811, 20, 3),      # k=20 (u64 k-mer values)
    (300, 1531, 31, 3),     # k=31: base-5 value wraps mod 2^64
    (500, 1201, 13, 2),     # h=2: generic hash path
    (500, 1201, 13, 4),     # h=4
    (64, 1, 13, 3),         # a single block
]


@pytest.mark.parametrize("n_bins,n_blocks,k,h", GEOMETRIES)
def test_raw_max_matches_oracle(n_bins, n_blocks, k, h):
    rng = np.random.default_rng(n_bins * 7 + n_blocks + k + h)
    W = (n_bins + 63) // 64
    n_bits = n_blocks * W * 64 + int(rng.integers(0, 64 * W))  # ragged tail bits that belong to no block
    d = capi.DeviceIBF.create(0, n_bins, h, k, n_bits)
    assert d.info["n_blocks"] == n_blocks and d.info["bin_width"] == W
    d.fill_synth(1234 + n_bins)
    ref = H.random_dna(rng, 6000)
    nb = min(n_bins, 12)
    starts = np.arange(nb, dtype=np.uint64) * 500
    d.insert(ref, starts, starts + 500, rng.choice(n_bins, size=nb, replace=False).astype(np.uint64))
    o, _keep = oracle_view(d)
    reads = make_reads(rng, ref, 96, lo=max(5, k - 3), hi=460)
    reads += ["", "A" * (k - 1), "A" * k, "N" * 50, "ACGT" * 90, ref[100:100 + k], ref[0:360]]
    buf, offs, lens = H.pack_reads(reads)
    eng = capi.Engine(0, [d], [])
    maxcount, best, decision, status = eng.classify(buf, offs, lens)  # micro-batch: latency form of K1
    expect = po.batch_raw_max(o, buf, offs, lens, 4)
    assert np.array_equal(maxcount[:, 0], expect)
    assert expect.max() > 100  # the planted reads really hit
    eng.set_split_threshold(0)  # same batch through the throughput form (one wave per read)
    mc2, _, dec2, st2 = eng.classify(buf, offs, lens)
    assert np.array_equal(mc2[:, 0], expect) and np.array_equal(dec2, decision) and np.array_equal(st2, status)
    eng.set_nt_threshold(0)  # non-temporal gathers, both forms
    assert np.array_equal(eng.classify(buf, offs, lens)[0][:, 0], expect)
    eng.set_split_threshold(2048)
    assert np.array_equal(eng.classify(buf, offs, lens)[0][:, 0], expect)
    eng.set_nt_threshold(512 << 20)
    # throughput form with clock-phased gathers (planned for one- and two-word tables of 6-128 MiB only; forced here on every
    # geometry -- wider blocks keep the plain kernel): windows from far shorter to far longer than a wave needs, the table cut
    # into as many as 8 / 32 / 3 slices (the built-in rule would give a test-sized table one slice)
    eng.set_split_threshold(0)
    for base_ticks, max_slices in ((1, 8), (300, 32), (2000, 3)):
        eng.set_phased(0, 1 << 40, base_ticks, 3, 1)
        eng.set_phase_slices(1, max_slices)
        mc3, _, dec3, st3 = eng.classify(buf, offs, lens)
        assert np.array_equal(mc3[:, 0], expect) and np.array_equal(dec3, decision) and np.array_equal(st3, status), base_ticks
    eng.set_phased()  # the built-in rules
    eng.set_phase_slices()
    eng.set_split_threshold(2048)
    # latency form with several workgroups per read (wide filters only; a no-op setting for the narrow ones):
    # workgroups per read x shares per 64-k-mer tile, twice each (the arrival counters must come back to zero)
    for parts, shares in ((1, 1), (2, 1), (8, 2), (8, 8), (3, 4), (16, 8), (8, 4)):
        eng.set_split_parts(parts, shares)
        for n_sub in (len(reads), 5, 5):
            sub_mc = eng.classify(buf, offs[:n_sub], lens[:n_sub])[0][:, 0]
            assert np.array_equal(sub_mc, expect[:n_sub]), (parts, shares, n_sub)
    # deplete-only decision + status against the oracle's check_unblock
    exp_dec, exp_st = po.batch_check_unblock([o], [], buf, offs, lens, n_threads=4)
    assert np.array_equal(decision, exp_dec)
    assert np.array_equal(status, exp_st)
    assert (status == capi.RB_ERR_SHORT_READ).sum() >= 2


@pytest.mark.parametrize("n_bins,n_blocks", [(1024, 2048), (8192, 300), (40, 5003)])
def test_long_reads_use_wide_counters(n_bins, n_blocks):
    # > 1023 k-mers per read -> 16 counter planes (uint16_t semantics of the reference)
    rng = np.random.default_rng(5)
    d = capi.DeviceIBF.create(0, n_bins, 3, 13, ((n_bins + 63) // 64) * 64 * n_blocks)
    ref = H.random_dna(rng, 30000)
    d.add_sequence(ref, 1000 if n_bins >= 1024 else 30000 // n_bins + 1)
    o, _k = oracle_view(d)
    reads = [ref[100:100 + 1500], ref[5000:5000 + 2300], H.random_dna(rng, 1800), ref[900:1100] * 8, ref[:4000]]
    buf, offs, lens = H.pack_reads(reads)
    eng = capi.Engine(0, [d], [])
    maxcount, _, decision, status = eng.classify(buf, offs, lens)
    assert np.array_equal(maxcount[:, 0], po.batch_raw_max(o, buf, offs, lens, 2))
    assert maxcount[:, 0].max() > 1023
    eng.set_split_threshold(0)
    assert np.array_equal(eng.classify(buf, offs, lens)[0], maxcount)
    eng.set_split_threshold(2048)
    for parts, shares in ((1, 1), (8, 8), (5, 2)):
        eng.set_split_parts(parts, shares)
        assert np.array_equal(eng.classify(buf, offs, lens)[0], maxcount), (parts, shares)
    exp_dec, exp_st = po.batch_check_unblock([o], [], buf, offs, lens)
    assert np.array_equal(decision, exp_dec) and np.array_equal(status, exp_st)


def test_fused_wide_filters_with_split_parts():
    """Four filters of one kernel geometry (40 word columns) share ONE latency launch; with several workgroups per
    read the workspace and the arrival counters are indexed by (filter, read)."""
    rng = np.random.default_rng(77)
    ref = H.random_dna(rng, 40000)
    filters, oracles, keep = [], [], []
    for i in range(4):
        d = capi.DeviceIBF.create(0, 2500 + 13 * i, 3, 13, 40 * 64 * (30011 + 7 * i))
        d.add_sequence(ref[i * 10000:(i + 1) * 10000], 500)
        o, h = oracle_view(d)
        filters.append(d); oracles.append(o); keep.append(h)
    reads = make_reads(rng, ref, 40, lo=10, hi=700)
    buf, offs, lens = H.pack_reads(reads)
    eng = capi.Engine(0, filters[:2], filters[2:])
    exp = np.stack([po.batch_raw_max(o, buf, offs, lens, 4) for o in oracles], axis=1)
    edec, est = po.batch_check_unblock(oracles[:2], oracles[2:], buf, offs, lens, n_threads=4)
    for parts, shares in ((8, 4), (1, 1), (4, 8), (8, 4)):
        eng.set_split_parts(parts, shares)
        for n_sub in (len(reads), 3, len(reads)):
            mc, _, dec, st = eng.classify(buf, offs[:n_sub], lens[:n_sub])
            assert np.array_equal(mc, exp[:n_sub]), (parts, shares, n_sub)
            assert np.array_equal(dec, edec[:n_sub]) and np.array_equal(st, est[:n_sub])
    assert len(set(edec.tolist())) >= 2


def test_fill_kernel_matches_oracle_definition():
    d = capi.DeviceIBF.create(0, 100, 3, 13, 128 * 3001 + 77)
    d.fill_synth(99)
    host = d.download()  # keep the image alive while its words are viewed
    got = host.words().copy()
    o = po.OracleIBF(100, 3, 13, 128 * 3001 + 77)
    o.fill_synth(99)
    assert np.array_equal(got, o.words())


@pytest.mark.parametrize("n_bins,k,frag", [(70, 13, 1000), (1024, 13, 977), (9, 15, 100000), (200, 20, 333)])
def test_insert_kernel_builds_identical_filter(n_bins, k, frag):
    # K4 + the reference fragmenter against the oracle's create_filter restatement: same bits
    rng = np.random.default_rng(n_bins + k)
    seqs = [H.random_dna(rng, int(L), with_n=0.001) for L in (frag * 3 + 17, frag, 5 * k, frag * 2 - 1)]
    bits = capi.calculate_filter_size_bits(frag, k, 3, 0.01, n_bins)
    assert bits == po.calculate_filter_size_bits(frag, k, 3, 0.01, n_bins)
    d = capi.DeviceIBF.create(0, n_bins, 3, k, bits)
    o = po.OracleIBF(n_bins, 3, k, bits)
    b_gpu = b_cpu = 0
    for s in seqs:
        c = capi.cut_out_nnns(s)
        assert c == po.cut_out_nnns(s)
        b_gpu = d.add_sequence(c, frag, b_gpu)
        b_cpu = o.add_sequence(po.encode(c), frag, b_cpu)
    assert b_gpu == b_cpu <= n_bins
    host = d.download()
    nw = o.n_bits // 64
    assert np.array_equal(host.words()[:nw], o.words()[:nw])
    assert int(host.words()[:nw].astype(np.uint64).sum()) != 0


def test_reference_kat_through_the_gpu(refdata):
    # 282 / 182 / index 0 / (282,182): src/test/libIBFTests/read.hpp:221-251
    filt = []
    for name in ("libIBFTests_test.fasta", "libIBFTests_test1.fasta"):
        recs = [s for _, s in H.read_fasta(os.path.join(refdata, name)) if len(s) >= 13]
        cleaned = [capi.cut_out_nnns(s) for s in recs]
        n_bins = sum(len(c) // 100000 + 1 for c in cleaned)
        bits = capi.calculate_filter_size_bits(100000, 13, 3, 0.01, n_bins)
        d = capi.DeviceIBF.create(0, n_bins, 3, 13, bits)
        b = 0
        for c in cleaned:
            b = d.add_sequence(c, 100000, b)
        filt.append(d)
    buf, offs, lens = H.pack_reads([READ_354, "AAAAAAACCCCCCCCCGAGAGAGGAGAGAGGAGAG"])
    # classify(vector<IBFMeta>) over both filters as targets
    eng = capi.Engine(0, [], filt)
    maxcount, best, decision, status = eng.classify(buf, offs, lens, mode=capi.RB_MODE_CLASSIFY_CHUNK)
    assert maxcount[0].tolist() == [282, 182] and best[0] == 0 and decision[0] == 1
    assert maxcount[1, 0] == 23 and best[1] == -1 and decision[1] == 0  # threshold -7 wraps: no match
    # pair overload: v1 = {test.ibf}, v2 = {test1.ibf}; both hit -> check_unblock re-tests and waits
    eng2 = capi.Engine(0, [filt[0]], [filt[1]])
    mc, _, dec, st = eng2.classify(buf, offs, lens)
    assert mc[0].tolist() == [282, 182] and dec[0] == 0 and st[0] == 0


@pytest.mark.parametrize("nd,nt", [(1, 1), (2, 3), (1, 0), (0, 2), (3, 0)])
def test_decisions_match_oracle(nd, nt):
    rng = np.random.default_rng(100 * nd + nt)
    ref = H.random_dna(rng, 40000)
    filters, views, keep = [], [], []
    geos = [(300, 13), (64, 13), (1024, 13), (200, 15), (90, 13)]
    for i in range(nd + nt):
        n_bins, k = geos[i % len(geos)]
        W = (n_bins + 63) // 64
        d = capi.DeviceIBF.create(0, n_bins, 3, k, W * 64 * 3001)
        # overlapping reference windows so that some reads hit deplete AND target filters
        lo = (i * 6000) % 30000
        d.add_sequence(ref[lo:lo + 12000], 1000)
        o, kp = oracle_view(d)
        filters.append(d); views.append(o); keep.append(kp)
    dep, tgt = filters[:nd], filters[nd:]
    odep, otgt = views[:nd], views[nd:]
    reads = make_reads(rng, ref, 400, lo=10, hi=450, err=0.12) + ["", "ACGTACGTACGT", "ACGTACGTACGTA", "ACGTACGTACGTAC" ]
    buf, offs, lens = H.pack_reads(reads)
    eng = capi.Engine(0, dep, tgt)
    for r in (0.1, 0.05, 0.15):
        maxcount, best, decision, status = eng.classify(buf, offs, lens, error_rate=r)
        exp_dec, exp_st = po.batch_check_unblock(odep, otgt, buf, offs, lens, r=r, n_threads=4)
        assert np.array_equal(decision, exp_dec), "check_unblock decisions differ at r=%g" % r
        assert np.array_equal(status, exp_st)
        # the same through the throughput form with the opt-in early-decision mode (decisions only: no raw maxima asked for)
        eng.set_split_threshold(0)
        eng.set_early_decision(1)
        dec_e, st_e = eng.decide(buf, offs, lens, error_rate=r)
        eng.set_early_decision(0)
        eng.set_split_threshold(2048)
        assert np.array_equal(dec_e, exp_dec) and np.array_equal(st_e, exp_st), "early-decision mode differs at r=%g" % r
    assert len(set(exp_dec.tolist())) >= 2
    # offline chunk semantics (classify.hpp): classified flag, credited target, failed status
    _, best, classified, status = eng.classify(buf, offs, lens, mode=capi.RB_MODE_CLASSIFY_CHUNK)
    for i, rd in enumerate(reads):
        if len(rd) == 0:
            continue
        res = po.classify_read_chunks(odep, otgt, rd, len(rd), 1)
        assert res["status"] == status[i], (i, len(rd))
        assert res["classified"] == bool(classified[i]), (i, len(rd))
        if res["classified"] and nt:
            assert res["best_target"] == best[i]


@pytest.mark.parametrize("fragment_length", [1500, 150, 31])
def test_ibf_file_roundtrip_through_hbm(tmp_path, fragment_length):
    # 14 bins (one word per block, file layout kept), 134 bins (3 words -> padded to 4 in HBM), 646 bins (11 -> 16)
    rng = np.random.default_rng(3)
    ref = H.random_dna(rng, 20000)
    o = H.build_filter_like_reference([ref], k=13, fragment_length=fragment_length)
    p = tmp_path / "ref.ibf"
    o.store(str(p))  # written by the oracle's restatement of seqan::store
    assert capi.is_ibf_file(str(p))
    d = capi.DeviceIBF.open(0, str(p))  # load_filter straight into HBM
    W = d.info["bin_width"]
    assert d.device_stride() == {1: 1, 3: 4, 11: 16}[W]
    assert (d.info["n_bins"], d.info["n_hash"], d.info["kmer_size"], d.info["n_bits"]) == (o.n_bins, 3, 13, o.n_bits)
    reads = make_reads(rng, ref, 64)
    buf, offs, lens = H.pack_reads(reads)
    eng = capi.Engine(0, [d], [])
    maxcount, _, _, _ = eng.classify(buf, offs, lens)
    assert np.array_equal(maxcount[:, 0], po.batch_raw_max(o, buf, offs, lens, 2))
    # and back: product store -> oracle load
    q = tmp_path / "back.ibf"
    d.download().store(str(q))
    with open(p, "rb") as a, open(q, "rb") as b:
        assert a.read() == b.read()


def test_device_pointer_api_and_column_shards():
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(8)
    d = capi.DeviceIBF.create(0, 8192, 3, 13, 8192 * 512)
    d.fill_synth(3)
    ref = H.random_dna(rng, 8000)
    d.add_sequence(ref, 100)
    o, _k = oracle_view(d)
    reads = make_reads(rng, ref, 300, lo=100, hi=400)
    buf, offs, lens = H.pack_reads(reads)
    dev = torch.device("cuda:0")
    t_seq = torch.from_numpy(buf).to(dev)
    t_off = torch.from_numpy(offs.view(np.int64)).to(dev)
    t_len = torch.from_numpy(lens.view(np.int32)).to(dev)
    n = len(reads)
    t_max = torch.zeros((n, 1), dtype=torch.int16, device=dev)
    t_dec = torch.zeros(n, dtype=torch.uint8, device=dev)
    t_st = torch.zeros(n, dtype=torch.uint8, device=dev)
    eng = capi.Engine(0, [d], [])
    torch.cuda.synchronize()  # inputs/outputs were produced on torch's default stream
    side = torch.cuda.Stream()  # a real (non-null) stream: the call is then fully asynchronous
    stream = side.cuda_stream
    assert stream != 0
    eng.classify_device(t_seq.data_ptr(), t_off.data_ptr(), t_len.data_ptr(), n, int(lens.max()),
                        d_maxcount=t_max.data_ptr(), d_decision=t_dec.data_ptr(), d_status=t_st.data_ptr(),
                        stream=stream)
    torch.cuda.synchronize()
    expect = po.batch_raw_max(o, buf, offs, lens, 4)
    assert np.array_equal(t_max.cpu().numpy().view(np.uint16)[:, 0], expect)
    exp_dec, exp_st = po.batch_check_unblock([o], [], buf, offs, lens, n_threads=4)
    assert np.array_equal(t_dec.cpu().numpy(), exp_dec)
    # an understated max_len is reported per read, not silently miscounted
    t_st2 = torch.zeros(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    eng.classify_device(t_seq.data_ptr(), t_off.data_ptr(), t_len.data_ptr(), n, 200, d_maxcount=t_max.data_ptr(),
                        d_decision=t_dec.data_ptr(), d_status=t_st2.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    too_long = lens > 200
    assert (t_st2.cpu().numpy()[too_long] == capi.RB_ERR_INVALID_ARG).all() and (t_st2.cpu().numpy()[~too_long] == exp_st[~too_long]).all()
    # lengths that are not lengths at all (a buffer the caller has not finished writing: 2^31 - 1, 10^9): K1 never looks past the
    # declared max_len of an item, the decision kernel reports the read, its neighbours are classified as ever
    wild = lens.copy()
    wild[[3, 77, n - 5]] = [2**31 - 1, 10**9, 4 * 10**9]  # (not the last reads: a clamped item still spans max_len bytes of the buffer)
    t_wild = torch.from_numpy(wild.view(np.int32)).to(dev)
    t_st3 = torch.zeros(n, dtype=torch.uint8, device=dev)
    t_max3 = torch.zeros((n, 1), dtype=torch.int16, device=dev)
    torch.cuda.synchronize()
    for thr in (2048, 0):  # latency and throughput forms
        eng.set_split_threshold(thr)
        eng.classify_device(t_seq.data_ptr(), t_off.data_ptr(), t_wild.data_ptr(), n, int(lens.max()), d_maxcount=t_max3.data_ptr(),
                            d_decision=t_dec.data_ptr(), d_status=t_st3.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        sane = wild == lens
        assert (t_st3.cpu().numpy()[~sane] == capi.RB_ERR_INVALID_ARG).all() and (t_st3.cpu().numpy()[sane] == exp_st[sane]).all()
        assert np.array_equal(t_max3.cpu().numpy().view(np.uint16)[sane, 0], expect[sane])
    eng.set_split_threshold(2048)
    # bin-sharded layout: per-rank partial maxima, element-wise max == unsharded result
    for world in (2, 3, 8):
        acc = np.zeros(n, dtype=np.uint16)
        for rank in range(world):
            eng.set_column_shard(rank, world)
            t_part = torch.zeros((n, 1), dtype=torch.int16, device=dev)
            torch.cuda.synchronize()
            eng.classify_device(t_seq.data_ptr(), t_off.data_ptr(), t_len.data_ptr(), n, int(lens.max()),
                                d_maxcount=t_part.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            acc = np.maximum(acc, t_part.cpu().numpy().view(np.uint16)[:, 0])
        assert np.array_equal(acc, expect)
        t_all = torch.from_numpy(acc.view(np.int16).reshape(n, 1)).to(dev)
        t_dec2 = torch.zeros(n, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        eng.decide_device(t_all.data_ptr(), t_len.data_ptr(), n, int(lens.max()), d_decision=t_dec2.data_ptr(),
                          stream=stream)
        torch.cuda.synchronize()
        assert np.array_equal(t_dec2.cpu().numpy(), exp_dec)
    eng.set_column_shard(0, 1)
    # the same with a narrow filter (5 word columns) through the phased throughput kernel: every rank gathers its columns
    d5 = capi.DeviceIBF.create(0, 300, 3, 13, 320 * 30011)
    d5.add_sequence(ref, 100)
    o5, _k5 = oracle_view(d5)
    exp5 = po.batch_raw_max(o5, buf, offs, lens, 4)
    eng5 = capi.Engine(0, [d5], [])
    eng5.set_split_threshold(0)
    eng5.set_phased(0, 1 << 40, 200, 0, 1)
    eng5.set_phase_slices(1, 8)
    for world in (1, 2, 3):
        acc = np.zeros(n, dtype=np.uint16)
        for rank in range(world):
            eng5.set_column_shard(rank, world)
            t_part = torch.zeros((n, 1), dtype=torch.int16, device=dev)
            torch.cuda.synchronize()
            eng5.classify_device(t_seq.data_ptr(), t_off.data_ptr(), t_len.data_ptr(), n, int(lens.max()),
                                 d_maxcount=t_part.data_ptr(), stream=stream)
            torch.cuda.synchronize()
            acc = np.maximum(acc, t_part.cpu().numpy().view(np.uint16)[:, 0])
        assert np.array_equal(acc, exp5), world


def test_empty_batch_and_null_filters():
    d = capi.DeviceIBF.create(0, 64, 3, 13, 64 * 100)
    eng = capi.Engine(0, [d], [])
    mc, best, dec, st = eng.classify(np.zeros(1, np.uint8), np.zeros(0, np.uint64), np.zeros(0, np.uint32))
    assert mc.shape == (0, 1)
    with pytest.raises(capi.RBError) as ei:
        capi.Engine(0, [], [])
    assert ei.value.status == capi.RB_ERR_NULL_FILTER


def test_host_api_micro_and_large_paths():
    """rb_classify_batch: the pinned micro-batch path (compaction of arbitrary, even overlapping, offsets) and the
    large-batch path (spanned range copied as is, base shifted by the lowest offset)."""
    rng = np.random.default_rng(21)
    d = capi.DeviceIBF.create(0, 60, 3, 13, 64 * 300007)
    ref = H.random_dna(rng, 50000)
    d.add_sequence(ref, 1000)
    o, _k = oracle_view(d)
    eng = capi.Engine(0, [d], [])
    pool = np.frombuffer((H.random_dna(rng, 1000) + ref + H.random_dna(rng, 50000)).encode(), dtype=np.uint8).copy()
    # micro: shuffled, overlapping windows, first byte used is far from 0
    n = 500
    offs = rng.integers(900, len(pool) - 400, size=n).astype(np.uint64)
    lens = rng.integers(0, 400, size=n).astype(np.uint32)
    mc, _, dec, st = eng.classify(pool, offs, lens)
    assert np.array_equal(mc[:, 0], po.batch_raw_max(o, pool, offs, lens, 4))
    edec, est = po.batch_check_unblock([o], [], pool, offs, lens, n_threads=4)
    assert np.array_equal(dec, edec) and np.array_equal(st, est)
    # pointer-array form (one buffer per read)
    reads = [bytes(pool[int(o):int(o) + int(l)]) for o, l in zip(offs[:200], lens[:200])]
    mc2, _, dec2, st2 = eng.classify_reads(reads)
    assert np.array_equal(mc2, mc[:200]) and np.array_equal(dec2, dec[:200]) and np.array_equal(st2, st[:200])
    # large: > 8 MB of read bytes
    n = 32000
    offs = rng.integers(1000, len(pool) - 300, size=n).astype(np.uint64)
    lens = np.full(n, 300, dtype=np.uint32)
    mc, _, dec, st = eng.classify(pool, offs, lens)
    assert np.array_equal(mc[:, 0], po.batch_raw_max(o, pool, offs, lens, 8))
    edec, est = po.batch_check_unblock([o], [], pool, offs, lens, n_threads=8)
    assert np.array_equal(dec, edec) and np.array_equal(st, est)
    assert 0 < dec.sum() < n
    # the same batch in many PCIe slices (shuffled, aliasing offsets: slices re-copy shared bytes), ragged slice size,
    # tiny slices, and no slicing at all
    lens_r = rng.integers(0, 600, size=n).astype(np.uint32)
    offs = rng.integers(1000, len(pool) - 600, size=n).astype(np.uint64)
    exp_mc = po.batch_raw_max(o, pool, offs, lens_r, 8)
    edec, est = po.batch_check_unblock([o], [], pool, offs, lens_r, n_threads=8)
    for slice_bytes in (1 << 20, 777_777, 4096, 0):  # 4096: a dozen reads per slice, thousands of slices
        eng.set_host_slice_bytes(slice_bytes)
        mc, _, dec, st = eng.classify(pool, offs, lens_r)
        assert np.array_equal(mc[:, 0], exp_mc), slice_bytes
        assert np.array_equal(dec, edec) and np.array_equal(st, est), slice_bytes
    eng.set_host_slice_bytes(32 << 20)


def test_more_ranks_than_columns():
    # bin-sharded layout with a one-word filter on two ranks: the second rank owns no column -> zero partials
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(4)
    d = capi.DeviceIBF.create(0, 50, 3, 13, 64 * 50021)
    ref = H.random_dna(rng, 5000)
    d.add_sequence(ref, 100)
    o, _k = oracle_view(d)
    reads = make_reads(rng, ref, 64, lo=100, hi=300)
    buf, offs, lens = H.pack_reads(reads)
    eng = capi.Engine(0, [d], [])
    expect = po.batch_raw_max(o, buf, offs, lens, 2)
    eng.set_column_shard(0, 2)
    p0 = eng.classify(buf, offs, lens)[0][:, 0]
    eng.set_column_shard(1, 2)
    p1 = eng.classify(buf, offs, lens)[0][:, 0]
    assert np.array_equal(p0, expect) and not p1.any()


def test_uint16_wraparound_semantics():
    """Reads with more than 65535 k-mers: the reference's uint16_t counters and its uint16_t readlen both wrap
    (IBFClassify.cpp:149-159); the 16-plane bit-sliced counters and the threshold table wrap identically."""
    rng = np.random.default_rng(31)
    d = capi.DeviceIBF.create(0, 64, 3, 13, 64 * 1000003)
    unit = H.random_dna(rng, 997)
    d.add_sequence(unit * 3, 100000)
    o, _k = oracle_view(d)
    reads = [unit * 70, unit * 66 + "ACGT", H.random_dna(rng, 66000), (unit * 67)[:65535 + 12], (unit * 67)[:65536 + 12]]
    buf, offs, lens = H.pack_reads(reads)
    assert lens.max() > 65535
    eng = capi.Engine(0, [d], [])
    for thr in (2048, 0):
        eng.set_split_threshold(thr)
        mc, _, dec, st = eng.classify(buf, offs, lens)
        assert np.array_equal(mc[:, 0], po.batch_raw_max(o, buf, offs, lens, 4))
        edec, est = po.batch_check_unblock([o], [], buf, offs, lens, n_threads=4)
        assert np.array_equal(dec, edec) and np.array_equal(st, est)


def test_pool_shards_reads_across_engines():
    """rb_pool: single-process multi-GPU form.  On a one-GPU box the device list repeats device 0, which still
    exercises replication, contiguous slicing, the worker threads and the merge of outputs."""
    rng = np.random.default_rng(77)
    ref = H.random_dna(rng, 40000)
    images, views = [], []
    for n_bins, lo in ((300, 0), (64, 20000)):
        d = capi.DeviceIBF.create(0, n_bins, 3, 13, ((n_bins + 63) // 64) * 64 * 100003)
        d.add_sequence(ref[lo:lo + 20000], 1000)
        h = d.download()
        images.append(h)
        views.append(po.OracleIBF.wrap(h.info["n_bins"], 3, 13, h.info["n_bits"], h.words()))
    reads = make_reads(rng, ref, 3001, lo=5, hi=420)
    buf, offs, lens = H.pack_reads(reads)
    exp_dec, exp_st = po.batch_check_unblock(views[:1], views[1:], buf, offs, lens, n_threads=8)
    exp_max = np.stack([po.batch_raw_max(v, buf, offs, lens, 8) for v in views], axis=1)
    for devices in ([0], [0, 0], [0, 0, 0], [0] * 8):
        pool = capi.Pool(devices, images[:1], images[1:])
        assert pool.size() == len(devices)
        for min_split in (1, 500, 4096):
            pool.set_min_split(min_split)
            mc, best, dec, st = pool.classify(buf, offs, lens)
            assert np.array_equal(mc, exp_max) and np.array_equal(dec, exp_dec) and np.array_equal(st, exp_st)
        pool.destroy()
    assert len(set(exp_dec.tolist())) == 3
    # the same pool from .ibf FILES: streamed into device 0 once, replicated device to device (xGMI between peers; a
    # same-device copy on this box), no host image -- same outputs
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        paths = []
        for i, h in enumerate(images):
            paths.append(os.path.join(tmp, "f%d.ibf" % i))
            h.store(paths[-1])
        for devices in ([0], [0, 0, 0]):
            pool = capi.Pool.from_files(devices, paths[:1], paths[1:])
            assert pool.size() == len(devices) and pool.replication_seconds >= 0.0
            pool.set_min_split(500)
            mc, best, dec, st = pool.classify(buf, offs, lens)
            assert np.array_equal(mc, exp_max) and np.array_equal(dec, exp_dec) and np.array_equal(st, exp_st)
            pool.destroy()
        # the fall-back ladder of rb_pool_create_from_files, walked by an injected failure (the copies themselves cannot
        # fail on this box): a clone refused at its start, or lost at its end, makes that device stream the file itself --
        # the pool comes up the same and classifies the same (code that never ran before an 8-GPU node).
        # The injection exists in the TESTING build of the library only (-DRB_TESTING, libreadbouncer_amd_testing.so): a child
        # process loads that build; the product library ignores the variable (checked last).
        import subprocess
        np.save(os.path.join(tmp, "buf.npy"), buf), np.save(os.path.join(tmp, "offs.npy"), offs), np.save(os.path.join(tmp, "lens.npy"), lens)
        np.save(os.path.join(tmp, "mc.npy"), exp_max), np.save(os.path.join(tmp, "dec.npy"), exp_dec)
        child = ("import os, sys, numpy as np\n"
                 "sys.path.insert(0, %r)\n"
                 "from readbouncer_amd import capi\n"
                 "t = %r\n"
                 "L = lambda n: np.load(os.path.join(t, n + '.npy'))\n"
                 "pool = capi.Pool.from_files([0, 0, 0], [os.path.join(t, 'f0.ibf')], [os.path.join(t, 'f1.ibf')])\n"
                 "assert pool.size() == 3\n"
                 "pool.set_min_split(500)\n"
                 "mc, best, dec, st = pool.classify(L('buf'), L('offs'), L('lens'))\n"
                 "assert np.array_equal(mc, L('mc')) and np.array_equal(dec, L('dec'))\n"
                 "pool.destroy()\n"
                 "print('ladder ok', os.environ.get('RB_POOL_TEST_FAIL_CLONE'))\n") % (ROOT, tmp)
        testing_lib = os.path.join(ROOT, "readbouncer_amd", "libreadbouncer_amd_testing.so")
        assert os.path.exists(testing_lib), "build() makes the testing build next to the product library"
        for where in ("start", "finish"):
            env = dict(os.environ, RB_POOL_TEST_FAIL_CLONE=where, RB_AMD_LIBRARY=testing_lib)
            r = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=600)
            assert r.returncode == 0 and "ladder ok " + where in r.stdout, (where, r.stdout[-500:], r.stderr[-1500:])
        # replicas of different GPUs are allocated (placement trial included) from a thread per GPU: on one GPU the testing build starts
        # every worker's replica from a thread of its own instead
        env = dict(os.environ, RB_POOL_TEST_THREAD_PER_WORKER="1", RB_AMD_LIBRARY=testing_lib)
        env.pop("RB_POOL_TEST_FAIL_CLONE", None)
        r = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and "ladder ok None" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
        os.environ["RB_POOL_TEST_FAIL_CLONE"] = "start"  # the product library has no such switch: nothing changes
        try:
            pool = capi.Pool.from_files([0, 0, 0], paths[:1], paths[1:])
        finally:
            del os.environ["RB_POOL_TEST_FAIL_CLONE"]
        assert pool.size() == 3
        pool.destroy()
        with pytest.raises(capi.RBError):
            capi.Pool.from_files([0, 0], [os.path.join(tmp, "missing.ibf")], [])
    # a clone is a bit-identical replica
    d = capi.DeviceIBF.upload(0, images[0])
    c = d.clone_to(0)
    assert np.array_equal(c.download().words(), images[0].words())
    cmp_ = d.compare(c)
    assert cmp_["new_bits"] == 0 and cmp_["file_bits"] == cmp_["rebuilt_bits"] > 0


def test_alphabet_conversion_on_device():
    """(seqan::Dna5String) conversion inside K1: lower case, U/u as T, IUPAC codes and any other byte as N (ordinal 4),
    which is hashed like a fifth letter."""
    rng = np.random.default_rng(12)
    d = capi.DeviceIBF.create(0, 200, 3, 13, 256 * 50021)
    ref = H.random_dna(rng, 20000)
    d.add_sequence(ref, 100)
    o, _k = oracle_view(d)
    base = [ref[i:i + 300] for i in range(0, 6000, 300)]
    reads = []
    for i, r in enumerate(base):
        if i % 4 == 0:
            r = r.lower()
        elif i % 4 == 1:
            r = r.replace("T", "U")
        elif i % 4 == 2:
            r = "".join(c if rng.random() > 0.03 else "RYKMSWBDHVN-*x"[int(rng.integers(0, 14))] for c in r)
        else:
            r = "".join(c.lower() if rng.random() < 0.5 else c for c in r).replace("t", "u")
        reads.append(r)
    buf, offs, lens = H.pack_reads(reads)
    raw = np.frombuffer(bytes(range(256)) * 2, dtype=np.uint8).copy()  # every byte value, incl. NUL and > 127
    buf = np.concatenate([buf, raw])
    offs = np.append(offs, np.uint64(len(buf) - len(raw)))
    lens = np.append(lens, np.uint32(len(raw)))
    eng = capi.Engine(0, [d], [])
    mc, _, dec, st = eng.classify(buf, offs, lens)
    assert np.array_equal(mc[:, 0], po.batch_raw_max(o, buf, offs, lens, 2))
    edec, est = po.batch_check_unblock([o], [], buf, offs, lens, n_threads=2)
    assert np.array_equal(dec, edec) and np.array_equal(st, est)
    pb, po_, pl = H.pack_reads(base)
    plain = po.batch_raw_max(o, pb, po_, pl, 2)
    for i in range(len(base)):
        if i % 4 == 2:
            assert mc[i, 0] <= plain[i]
        else:
            assert mc[i, 0] == plain[i]  # case and U do not cost a single k-mer
    assert plain.min() > 50


@pytest.mark.parametrize("old_bins,new_bins", [(40, 60), (60, 70), (64, 65), (100, 1000), (130, 130)])
def test_resize_bins_and_update(old_bins, new_bins):
    """resizeBins + adding sequences to the new bins (IBF::update_filter, IBFBuild.cpp:223-321): identical to the
    oracle's restatement, old bins keep their counts."""
    rng = np.random.default_rng(old_bins * 1000 + new_bins)
    ref = H.random_dna(rng, 6000)
    add = H.random_dna(rng, 3000)
    W = (old_bins + 63) // 64
    d = capi.DeviceIBF.create(0, old_bins, 3, 13, W * 64 * 5003 + 40)
    o = po.OracleIBF(old_bins, 3, 13, W * 64 * 5003 + 40)
    frag = 6000 // min(old_bins, 30) + 1
    d.add_sequence(ref, frag)
    o.add_sequence(po.encode(ref), frag)
    d2, o2 = d.resize_bins(new_bins), o.resize_bins(new_bins)
    assert (d2.info["n_bins"], d2.info["n_blocks"], d2.info["n_bits"]) == (o2.n_bins, o2.n_blocks, o2.n_bits) == \
           (new_bins, 5003, 5003 * 64 * ((new_bins + 63) // 64))
    first_new = old_bins
    if new_bins > old_bins:
        frag2 = 3000 // (new_bins - old_bins) + 14
        b1 = d2.add_sequence(add, frag2, first_new)
        b2 = o2.add_sequence(po.encode(add), frag2, first_new)
        assert b1 == b2
    host = d2.download()
    nw = o2.n_bits // 64
    assert np.array_equal(host.words()[:nw], o2.words()[:nw])
    reads = [ref[100:400], add[50:350], ref[3000:3300], H.random_dna(rng, 300)]
    buf, offs, lens = H.pack_reads(reads)
    mc = capi.Engine(0, [d2], []).classify(buf, offs, lens)[0][:, 0]
    assert np.array_equal(mc, po.batch_raw_max(o2, buf, offs, lens))
    with pytest.raises(capi.RBError):
        d2.resize_bins(new_bins - 1)


@pytest.mark.parametrize("form", ["latency", "throughput", "phased"])
def test_packed_reads_and_on_gpu_chunking(form):
    """2-bit + N-bitmap reads and chunk selection on the device (classify.hpp:262-271) give exactly what the
    ASCII path gives on the sliced strings; read-id indirection covers the 'still unclassified' subset of a chunk loop.
    Through every form of K1: the latency kernels (micro-batch), the plain throughput kernels, and the phased kernels with
    their both-strands tiles (forced: the filters here are far smaller than the ones they are planned for)."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(19)
    ref = H.random_dna(rng, 30000)
    dep = capi.DeviceIBF.create(0, 300, 3, 13, 320 * 50021)
    tgt = capi.DeviceIBF.create(0, 64, 3, 13, 64 * 200003)
    dep.add_sequence(ref[:15000], 500)
    tgt.add_sequence(ref[15000:], 500)
    od, _k1 = oracle_view(dep)
    ot, _k2 = oracle_view(tgt)
    reads = make_reads(rng, ref, 300, lo=5, hi=1500, err=0.1, n_frac=0.3)
    reads += ["", "ACGTN", "n" * 400, ref[100:460].lower(), ref[20000:20360].replace("T", "U")]
    buf, offs, lens = H.pack_reads(reads)
    n = len(reads)
    eng = capi.Engine(0, [dep], [tgt])
    if form != "latency":
        eng.set_split_threshold(0)
        if form == "phased":
            eng.set_phased(0, 1 << 40, 200, 0, 1)
            eng.set_phase_slices(1, 32)
        else:
            eng.set_phased(0, 0, 0, 0, 0)
    packed, p_off, nmask, n_off = capi.pack_reads(buf, offs, lens)
    assert len(packed) < len(buf) // 3 + n  # ~4x smaller payload
    dev = torch.device("cuda:0")
    up = lambda a, dt: torch.from_numpy(a.view(dt)).to(dev)
    t_buf, t_offs, t_lens = up(buf, np.uint8), up(offs, np.int64), up(lens, np.int32)
    t_pk, t_poff, t_nm, t_noff = up(packed, np.uint8), up(p_off, np.int64), up(nmask, np.uint8), up(n_off, np.int64)
    max_len = int(lens.max())

    def run(packed_input, chunk_start, chunk_length, ids=None, mode=capi.RB_MODE_CLASSIFY_CHUNK):
        m = n if ids is None else len(ids)
        t_ids = None if ids is None else torch.from_numpy(ids.astype(np.int32)).to(dev)
        t_mc = torch.zeros((m, 2), dtype=torch.int16, device=dev)
        t_best = torch.zeros(m, dtype=torch.int32, device=dev)
        t_dec = torch.zeros(m, dtype=torch.uint8, device=dev)
        t_st = torch.zeros(m, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        eng.classify_device_ex(t_pk.data_ptr() if packed_input else t_buf.data_ptr(),
                               t_poff.data_ptr() if packed_input else t_offs.data_ptr(), t_lens.data_ptr(), m, max_len,
                               d_nmask=t_nm.data_ptr() if packed_input else None,
                               d_nmask_offsets=t_noff.data_ptr() if packed_input else None,
                               chunk_start=chunk_start, chunk_length=chunk_length,
                               d_read_ids=None if ids is None else t_ids.data_ptr(), mode=mode,
                               d_maxcount=t_mc.data_ptr(), d_best=t_best.data_ptr(), d_decision=t_dec.data_ptr(),
                               d_status=t_st.data_ptr())
        torch.cuda.synchronize()
        return (t_mc.cpu().numpy().view(np.uint16), t_best.cpu().numpy(), t_dec.cpu().numpy(), t_st.cpu().numpy())

    # whole reads: packed == ASCII == oracle
    whole = eng.classify(buf, offs, lens, mode=capi.RB_MODE_CLASSIFY_CHUNK)
    for packed_input in (False, True):
        got = run(packed_input, 0, 0)
        for a, b in zip(got, whole):
            assert np.array_equal(a, b)
    assert np.array_equal(whole[0][:, 0], po.batch_raw_max(od, buf, offs, lens, 4))
    assert np.array_equal(whole[0][:, 1], po.batch_raw_max(ot, buf, offs, lens, 4))

    # chunk c of length L, as classify_reads forms it
    for L in (250, 360):
        for c in range(0, 5):
            frags, bad = [], np.zeros(n, dtype=bool)
            for i, r in enumerate(reads):
                s, e = c * L, min((c + 1) * L, len(r))
                bad[i] = s > e
                frags.append("" if bad[i] else r[s:e])
            fb, fo, fl = H.pack_reads(frags)
            exp = eng.classify(fb, fo, fl, mode=capi.RB_MODE_CLASSIFY_CHUNK)
            for packed_input in (False, True):
                mc, best, dec, st = run(packed_input, c * L, L)
                ok = ~bad
                assert np.array_equal(mc[ok], exp[0][ok]) and np.array_equal(best[ok], exp[1][ok])
                assert np.array_equal(dec[ok], exp[2][ok]) and np.array_equal(st[ok], exp[3][ok])
                assert (st[bad] == capi.RB_ERR_BAD_CHUNK).all() and not dec[bad].any()
            # against the oracle's chunk driver semantics for this very chunk
            for i in rng.choice(n, size=40, replace=False):
                if bad[i] or len(frags[i]) == 0:
                    continue
                res = po.classify_read_chunks([od], [ot], frags[i], len(frags[i]), 1)
                assert res["status"] == st[i] and res["classified"] == bool(dec[i])

    # the 'still unclassified' subset through read ids (check_unblock mode this time)
    ids = np.sort(rng.choice(n, size=77, replace=False)).astype(np.uint32)
    sub_b, sub_o, sub_l = H.pack_reads([reads[i][360:720] for i in ids if len(reads[i]) >= 360])
    keep = np.array([i for i in ids if len(reads[i]) >= 360], dtype=np.uint32)
    exp = eng.classify(sub_b, sub_o, sub_l)
    got = run(True, 360, 360, ids=keep, mode=capi.RB_MODE_CHECK_UNBLOCK)
    for a, b in zip(got, exp):
        assert np.array_equal(a, b)


def test_engine_is_thread_safe():
    """SURVEY 8b threading: the reference's N classify threads share the filters read-only (adaptive_sampling.hpp:745-750).
    Several host threads call rb_classify_batch on ONE engine and on separate engines over the same filters."""
    import threading
    rng = np.random.default_rng(55)
    ref = H.random_dna(rng, 30000)
    d = capi.DeviceIBF.create(0, 500, 3, 13, 512 * 60013)
    d.add_sequence(ref, 200)
    o, _k = oracle_view(d)
    batches = []
    for t in range(6):
        reads = make_reads(np.random.default_rng(100 + t), ref, 150 + 40 * t, lo=5, hi=420)
        buf, offs, lens = H.pack_reads(reads)
        exp_max = po.batch_raw_max(o, buf, offs, lens, 4)
        exp_dec, exp_st = po.batch_check_unblock([o], [], buf, offs, lens, n_threads=4)
        batches.append((buf, offs, lens, exp_max, exp_dec, exp_st))
    shared = capi.Engine(0, [d], [])
    own = [capi.Engine(0, [d], []) for _ in range(3)]
    errors = []

    def worker(t):
        try:
            eng = shared if t < 3 else own[t - 3]
            buf, offs, lens, exp_max, exp_dec, exp_st = batches[t]
            for _ in range(25):
                mc, _, dec, st = eng.classify(buf, offs, lens)
                if not (np.array_equal(mc[:, 0], exp_max) and np.array_equal(dec, exp_dec) and np.array_equal(st, exp_st)):
                    errors.append("mismatch in thread %d" % t)
                    return
        except Exception as e:  # noqa: BLE001
            errors.append("thread %d: %r" % (t, e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert errors == []


@pytest.mark.parametrize("nd,nt", [(0, 1), (0, 3), (2, 1)])
def test_classify_any_matches_oracle(nd, nt):
    """Read::classify(std::vector<TIbf>&) (IBFClassify.cpp:181-226 -> find_matches :81-128 -> select_matches :16-38):
    any bin >= the uint16_t threshold in any filter of the list.  Differs from `classify(metas) > -1` where the
    threshold is 0 (123..130 bp at k=13, r=0.1): every such read is a hit there, with or without a match."""
    rng = np.random.default_rng(900 + 10 * nd + nt)
    ref = H.random_dna(rng, 30000)
    filters, views, keep = [], [], []
    geos = [(90, 13), (200, 15), (64, 13)]
    for i in range(nd + nt):
        n_bins, k = geos[i % len(geos)]
        W = (n_bins + 63) // 64
        d = capi.DeviceIBF.create(0, n_bins, 3, k, W * 64 * 200003)  # sparse: unrelated reads really count 0 everywhere
        d.add_sequence(ref[i * 5000: i * 5000 + 9000], 1000)
        o, kp = oracle_view(d)
        filters.append(d); views.append(o); keep.append(kp)
    zero_thr = [L for L in range(100, 160) if capi.threshold(L, 13, 0.1, 0.95) == 0]
    assert zero_thr and min(zero_thr) >= 120 and max(zero_thr) <= 135  # the window the reference's quirk lives in
    reads = make_reads(rng, ref, 300, lo=10, hi=450, err=0.12)
    reads += [H.random_dna(rng, L) for L in range(118, 136)]            # no hit, threshold 0 inside the window
    reads += [H.mutate(rng, ref[100:100 + L], 0.3) for L in range(118, 136)]
    reads += ["", "ACGTACGTACGT", "ACGTACGTACGTA", "ACGTACGTACGTAC", "ACGTACGTACGTACG"]
    buf, offs, lens = H.pack_reads(reads)
    eng = capi.Engine(0, filters[:nd], filters[nd:])
    for r in (0.1, 0.07, 0.14):
        _, _, found, status = eng.classify(buf, offs, lens, error_rate=r, mode=capi.RB_MODE_CLASSIFY_ANY)
        n_quirk = 0
        for i, rd in enumerate(reads):
            st, exp = po.classify_any(views, po.encode(rd), r=r)
            assert status[i] == st, (i, len(rd), r)
            if st == 0:
                assert bool(found[i]) == exp, (i, len(rd), r)
                best_st, best = po.classify_best(views, po.encode(rd), r=r)
                n_quirk += int(exp and best == -1)
        if r == 0.1:
            assert n_quirk >= 8  # reads that the bool overload accepts and the argmax overload does not


def test_multi_workgroup_latency_kernel_back_to_back():
    """The latency form on a wide filter spreads a read over several workgroups that meet through a workspace and an
    arrival counter per (filter, read) which the last workgroup resets (rb_kernels.hip, split_body).  Thousands of
    micro-batches back to back, at the largest number of parts and varying batch sizes, against the throughput form:
    a counter left non-zero, a stale partial sum or a missed release would show as a wrong maximum."""
    rng = np.random.default_rng(2024)
    ref = H.random_dna(rng, 30000)
    d = capi.DeviceIBF.create(0, 8192, 3, 13, 8192 * 4099)  # 128 word columns: 16-byte lanes, up to 8 parts
    d.fill_synth(99)
    d.add_sequence(ref, 100)
    t = capi.DeviceIBF.create(0, 64, 3, 13, 64 * 30011)     # a narrow target in the same launch
    t.add_sequence(ref[:6000], 100)
    reads = make_reads(rng, ref, 64, lo=300, hi=420, err=0.1)
    buf, offs, lens = H.pack_reads(reads)
    eng = capi.Engine(0, [d], [t])
    eng.set_split_threshold(0)
    exp_mc, _, exp_dec, exp_st = eng.classify(buf, offs, lens)  # throughput form
    od, _k1 = oracle_view(d)
    assert np.array_equal(exp_mc[:, 0], po.batch_raw_max(od, buf, offs, lens, 4))
    eng.set_split_threshold(2048)
    eng.set_split_parts(8, 4)
    bad = 0
    for it in range(3000):
        n = (1, 2, 3, 7, 14, 25, 64)[it % 7]
        lo = (it * 5) % (len(reads) - n + 1)
        sub = np.ascontiguousarray(buf[int(offs[lo]): int(offs[lo + n - 1]) + int(lens[lo + n - 1])])
        mc, _, dec, st = eng.classify(sub, offs[lo:lo + n] - offs[lo], lens[lo:lo + n])
        bad += int(not (np.array_equal(mc, exp_mc[lo:lo + n]) and np.array_equal(dec, exp_dec[lo:lo + n])))
    assert bad == 0


def test_threshold_tables_survive_eviction_and_growth():
    """The engine keeps the threshold tables of the two most recent (error rate, significance) pairs resident, appends rows
    when a longer read shows up and parks replaced copies while queued work may still read them.  Three error rates in
    rotation (every call evicts), lengths growing across calls, all back to back on one engine: decisions stay the
    oracle's."""
    rng = np.random.default_rng(77)
    ref = H.random_dna(rng, 20000)
    d = capi.DeviceIBF.create(0, 200, 3, 13, 256 * 40009)
    d.add_sequence(ref, 1000)
    o, _k = oracle_view(d)
    eng = capi.Engine(0, [d], [])
    rates = (0.1, 0.05, 0.15)
    for it in range(36):
        top = 300 + 150 * it  # 300 .. 5550: crosses the 1024-, 2048- and 4096-row table sizes
        reads = [H.mutate(rng, ref[s:s + L], 0.1) for s, L in zip(rng.integers(0, 12000, size=24), rng.integers(20, top, size=24))]
        buf, offs, lens = H.pack_reads(reads)
        r = rates[it % 3]
        _, _, dec, st = eng.classify(buf, offs, lens, error_rate=r)
        exp_dec, exp_st = po.batch_check_unblock([o], [], buf, offs, lens, r=r, n_threads=4)
        assert np.array_equal(dec, exp_dec) and np.array_equal(st, exp_st), (it, r, top)
    with pytest.raises(capi.RBError):  # NormalCDFInverse would throw (IBF.hpp:284-308)
        eng.classify(buf, offs, lens, significance=1.5)


N_REVERSE = "ATAATATATAANATCTCCTCTCTTTTGGGGCTCTCTCTCTCC"  # tests/test_oracle_kat.py: revcomp of test.fasta[30:72], one A -> N


@pytest.mark.parametrize("n_rule", [3, 4])
def test_n_reads_under_both_revcomp_rules(refdata, n_rule):
    """What the reverse strand holds for an N of the read (src/IBF/IBF.hpp:96-97: ModComplementDna over a Dna5String) is a
    recalled SeqAn fact with two candidates: T (3, the default) and N (4).  Kernels and oracle each keep ONE constant and
    a switch; both candidates run through every kernel form against the oracle under the same rule, and the KAT read
    whose reverse-strand count tells them apart (30 vs 18) goes through the GPU."""
    prev = po.set_revcomp_of_n(n_rule)
    try:
        rng = np.random.default_rng(40 + n_rule)
        ref = H.random_dna(rng, 12000)
        comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
        rc = lambda t: "".join(comp[c] for c in reversed(t))
        reads = []
        for i in range(160):
            L = int(rng.integers(20, 900 if i % 8 == 0 else 420))
            s = int(rng.integers(0, len(ref) - L))
            r = H.mutate(rng, ref[s:s + L], 0.03)
            if i % 2:
                r = rc(r)  # positives on the reverse strand: this is where the rule shows
            r = list(r)
            for p in rng.integers(0, L, size=int(rng.integers(1, 5))):
                r[int(p)] = "N" if rng.random() < 0.8 else "n"
            reads.append("".join(r))
        reads += ["N" * 40, "N", "ACGTN" * 30, "N" + ref[50:300], rc(ref[50:300]) + "N", rc(ref[400:760])[:100] + "NN" + rc(ref[400:760])[102:]]
        buf, offs, lens = H.pack_reads(reads)
        differs = 0
        for n_bins, n_blocks in ((40, 30011), (100, 20011), (200, 9973), (1024, 4099), (8192, 257)):
            W = (n_bins + 63) // 64
            d = capi.DeviceIBF.create(0, n_bins, 3, 13, W * 64 * n_blocks)
            d.add_sequence(ref, 12000 // min(n_bins, 60) + 1)
            o, _keep = oracle_view(d)
            expect = po.batch_raw_max(o, buf, offs, lens, 4)
            po.set_revcomp_of_n(7 - n_rule)
            differs += int((po.batch_raw_max(o, buf, offs, lens, 4) != expect).sum())
            po.set_revcomp_of_n(n_rule)
            exp_dec, exp_st = po.batch_check_unblock([o], [], buf, offs, lens, n_threads=4)
            eng = capi.Engine(0, [d], [])
            eng.set_revcomp_of_n(n_rule)
            for form in ("latency", "throughput", "phased"):
                eng.set_split_threshold(2048 if form == "latency" else 0)
                if form == "phased":
                    eng.set_phased(0, 1 << 40, 200, 0, 1)
                    eng.set_phase_slices(1, 8)
                else:
                    eng.set_phased(0, 0, 0, 0, 0) if form == "throughput" else eng.set_phased()
                mc, _, dec, st = eng.classify(buf, offs, lens)
                assert np.array_equal(mc[:, 0], expect), (n_bins, form)
                assert np.array_equal(dec, exp_dec) and np.array_equal(st, exp_st), (n_bins, form)
            with pytest.raises(capi.RBError):
                eng.set_revcomp_of_n(2)
        assert differs > 50  # the two rules really give different counts on this batch
        # the KAT: filter of the reference's test.fasta, read N_REVERSE -> 30 shared 13-mers under T-for-N, 18 under N-for-N
        (_, seq), = H.read_fasta(os.path.join(refdata, "libIBFTests_test.fasta"))
        c = capi.cut_out_nnns(seq)
        d = capi.DeviceIBF.create(0, 1, 3, 13, capi.calculate_filter_size_bits(100000, 13, 3, 0.01, 1))
        d.add_sequence(c, 100000, 0)
        eng = capi.Engine(0, [d], [])
        eng.set_revcomp_of_n(n_rule)
        kb, ko, kl = H.pack_reads([N_REVERSE])
        assert eng.classify(kb, ko, kl)[0][0, 0] == {3: 30, 4: 18}[n_rule]
        if n_rule == 3:  # and 3 is what an engine does when nobody tells it anything
            assert capi.Engine(0, [d], []).classify(kb, ko, kl)[0][0, 0] == 30
    finally:
        po.set_revcomp_of_n(prev)


def test_bin_sharded_rank_with_odd_stride_keeps_the_plain_kernel():
    """a bin-sharded rank can reach <= 8 word columns on a filter whose HBM block stride is not a power of two
    (3072 bins: 48 words, stride 48; 6 ranks x 8 columns).  The phased kernels take a lookup's slice from its byte offset
    by a shift, which needs a power-of-two stride: the planner must not pick them there, even when forced."""
    rng = np.random.default_rng(91)
    ref = H.random_dna(rng, 20000)
    for n_bins in (3072, 2560):
        W = (n_bins + 63) // 64
        d = capi.DeviceIBF.create(0, n_bins, 3, 13, W * 64 * 24007)  # 9.2 / 7.7 MB: inside the phased size range
        stride = d.device_stride()  # 48 for both: 48 words as they are, 40 words padded to the next multiple of 16
        assert stride == (W + 15) // 16 * 16 and (stride & (stride - 1)) != 0
        d.fill_synth(17)
        d.add_sequence(ref, 50)
        o, _keep = oracle_view(d)
        reads = make_reads(rng, ref, 300, lo=100, hi=400)
        buf, offs, lens = H.pack_reads(reads)
        expect = po.batch_raw_max(o, buf, offs, lens, 4)
        eng = capi.Engine(0, [d], [])
        eng.set_split_threshold(0)
        for forced in (False, True):
            if forced:
                eng.set_phased(0, 1 << 40, 300, 0, 1)
                eng.set_phase_slices(1, 32)
            acc = np.zeros(len(reads), dtype=np.uint16)
            for rank in range(6):
                eng.set_column_shard(rank, 6)
                acc = np.maximum(acc, eng.classify(buf, offs, lens)[0][:, 0])
            assert np.array_equal(acc, expect), (n_bins, forced)
        eng.set_column_shard(0, 1)


def test_pool_from_resident_filters_and_its_statistics():
    """rb_pool_create_from_device: replicas of filters that are already in HBM, copied device to device onto every listed device
    (the one GPU of the box, three times); outputs equal a single engine's; rb_pool_get_stats accounts for every read"""
    rng = np.random.default_rng(5150)
    ref = H.random_dna(rng, 30000)
    filters = []
    for n_bins, lo in ((300, 0), (64, 15000)):
        d = capi.DeviceIBF.create(0, n_bins, 3, 13, ((n_bins + 63) // 64) * 64 * 50021)
        d.add_sequence(ref[lo:lo + 15000], 700)
        filters.append(d)
    reads = make_reads(rng, ref, 9000, lo=30, hi=420)
    buf, offs, lens = H.pack_reads(reads)
    eng = capi.Engine(0, filters[:1], filters[1:])
    exp = eng.classify(buf, offs, lens)
    pool = capi.Pool.from_device([0, 0, 0], filters[:1], filters[1:])
    assert pool.size() == 3 and pool.replication_seconds >= 0.0
    pool.set_min_split(1000)
    got = pool.classify(buf, offs, lens)
    for a, b in zip(got, exp):
        assert np.array_equal(a, b)
    st = pool.stats()
    assert len(st) == 3 and sum(r for _, _, r, _ in st) == len(lens) and all(c >= 1 and b > 0 for _, b, _, c in st)
    assert pool.stats(reset=True) == st and all(r == 0 for _, _, r, _ in pool.stats())
    pool.destroy()
    # the source filters are untouched and still the caller's
    assert np.array_equal(eng.classify(buf, offs, lens)[0], exp[0])


def test_measurement_aids_answer():
    """rb_dibf_probe_read_peak, rb_engine_plan and rb_dibf_touch do what their headers say (no timing asserted)"""
    d = capi.DeviceIBF.create(0, 8192, 3, 13, 8192 * 20011)
    gbps, ms = d.probe_read_peak(1024, False, 12, target_ms=5.0)
    assert gbps > 0 and ms > 0
    with pytest.raises(capi.RBError):
        d.probe_read_peak(100, False)  # rows of 128, 1024 or 4096 bytes
    with pytest.raises(capi.RBError):  # a table larger than the filter's own would gather out of bounds
        d.probe_read_peak(1024, False, 12, table_bytes=d.info["n_blocks"] * d.device_stride() * 8 + 1024, target_ms=5.0)
    eng = capi.Engine(0, [d], [])
    pl = eng.plan(0, 100000, 360)
    assert pl["kernel"] == "ibf_count_max_kernel" and pl["lanes_per_block_log2"] == 6 and pl["words_per_lane"] == 2 and not pl["phased"]
    assert eng.plan(0, 64, 360)["kernel"] == "ibf_count_max_split_kernel"
    small = capi.DeviceIBF.create(0, 64, 3, 13, 64 * 1310000)  # 10 MiB of one-word blocks
    e2 = capi.Engine(0, [small], [])
    pl = e2.plan(0, 100000, 250)
    assert pl["kernel"] == "ibf_count_max_phased_multi_kernel" and pl["phased"] == 1 and pl["phase_shape_name"].startswith("four tiles")
    e2.set_reads_per_wave(0)  # (the build with the offsets in registers: the same plan otherwise)
    assert e2.plan(0, 100000, 250)["kernel"] == "ibf_count_max_phased_kernel"
    e2.set_reads_per_wave(1)
    assert pl["phase_slices"] * (1 << pl["phase_slice_log2"]) >= pl["table_bytes"] and 100 <= pl["phase_window_ticks"] <= 2000
    assert capi.lib().rb_dibf_touch(small.h) == 0 and capi.lib().rb_dibf_touch(None) != 0
    # rb_engine_calibrate: windows measured on this device replace the table's for exactly that table and shape; results stay
    rng = np.random.default_rng(99)
    ref = H.random_dna(rng, 20000)
    small.add_sequence(ref, 400)
    reads = make_reads(rng, ref, 3000, lo=100, hi=250)
    buf, offs, lens = H.pack_reads(reads)
    before = e2.classify(buf, offs, lens)
    rule = e2.plan(0, 100000, 250)
    assert rule["phase_rule_ticks"] == rule["phase_window_ticks"]
    n_tables, n_changed = e2.calibrate(100000, 250)
    assert n_tables == 1 and n_changed in (0, 1)
    after = e2.plan(0, 100000, 250)
    assert after["phase_rule_ticks"] == rule["phase_rule_ticks"] and after["phase_slice_log2"] == rule["phase_slice_log2"]
    assert 0.69 * rule["phase_rule_ticks"] <= after["phase_window_ticks"] <= 1.46 * rule["phase_rule_ticks"]
    assert (after["phase_window_ticks"] != rule["phase_window_ticks"]) == (n_changed == 1)
    got = e2.classify(buf, offs, lens)
    for a, b in zip(got, before):
        assert np.array_equal(a, b)
    e2.set_phased()  # the setters drop what calibration found
    assert e2.plan(0, 100000, 250)["phase_window_ticks"] == rule["phase_rule_ticks"]
    assert eng.calibrate(50000, 360) == (0, 0)  # nothing phased in an engine of wide filters


def _pool_threads_run(timed):
    """The reference's N classification threads behind one queue (adaptive_sampling.hpp:745-751): K host threads calling
    rb_pool_classify_batch keep K engines busy.  4 threads x 500 micro-batches on a pool of four engines (one GPU, listed
    four times): every output equals the single-engine run (always asserted); timed: returns (serialised, concurrent) wall
    seconds, best of three each (rb_pool_set_serialize = what round 2 did)."""
    import threading
    import time
    rng = np.random.default_rng(321)
    ref = H.random_dna(rng, 40000)
    filters, images = [], []
    for i, (n_bins, bits) in enumerate(((1024, 1024 * 40009), (64, 64 * 300007))):
        d = capi.DeviceIBF.create(0, n_bins, 3, 13, bits)
        d.add_sequence(ref[i * 20000:(i + 1) * 20000], 500)
        filters.append(d)
        images.append(d.download())
    n_threads, n_batches, m = 4, 500, 24
    work = []
    eng = capi.Engine(0, filters[:1], filters[1:])
    for t in range(n_threads):
        reads = make_reads(np.random.default_rng(900 + t), ref, n_batches * m, lo=200, hi=420)
        buf, offs, lens = H.pack_reads(reads)
        work.append((buf, offs, lens, eng.classify(buf, offs, lens)))
    pool = capi.Pool([0, 0, 0, 0], images[:1], images[1:])
    pool.set_min_split(4096)  # micro-batches are never split: each goes to one engine

    def run():
        errors, outs = [], [None] * n_threads

        def worker(t):
            # straight through the C ABI with buffers made beforehand: ctypes drops the GIL for the call, and the few
            # microseconds of interpreter around it are not what is being measured
            try:
                buf, offs, lens, _ = work[t]
                n = len(lens)
                mc = np.zeros((n, 2), dtype=np.uint16)
                best = np.full(n, -1, dtype=np.int32)
                dec = np.zeros(n, dtype=np.uint8)
                st = np.zeros(n, dtype=np.uint8)
                fn, h = capi.lib().rb_pool_classify_batch, pool.h
                args = [(buf.ctypes.data, offs[b * m:].ctypes.data, lens[b * m:].ctypes.data, m, 0.1, 0.95, 0,
                         mc[b * m:].ctypes.data, best[b * m:].ctypes.data, dec[b * m:].ctypes.data, st[b * m:].ctypes.data)
                        for b in range(n_batches)]
                for a in args:
                    rc = fn(h, *a)
                    assert rc == 0, rc
                outs[t] = [mc, best, dec, st]
            except Exception as ex:  # noqa: BLE001
                errors.append(ex)
        threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
        t0 = time.perf_counter()
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        wall = time.perf_counter() - t0
        assert not errors, errors
        for t in range(n_threads):
            for got, exp in zip(outs[t], work[t][3]):
                assert np.array_equal(got, exp), t
        return wall

    run()  # warm-up: code objects, staging buffers, threshold tables of all four engines; outputs checked inside
    pool.set_serialize(True)
    serial = min(run() for _ in range(3 if timed else 1))
    pool.set_serialize(False)
    concurrent = min(run() for _ in range(3 if timed else 1))
    pool.destroy()
    return m, serial, concurrent


def test_pool_runs_micro_batches_of_several_threads_concurrently():
    """outputs of 4 threads x 500 micro-batches through a pool of four engines == the single-engine run, with the calls
    serialised and concurrent (no assertion on time: that is the gpuperf twin below)"""
    _pool_threads_run(timed=False)


@pytest.mark.gpuperf
def test_pool_concurrency_speedup():
    """wall time of the concurrent pool at most 0.45 x that of the same pool with the calls serialised"""
    m, serial, concurrent = _pool_threads_run(timed=True)
    print("pool: 4 threads x 500 micro-batches of %d reads: serialised %.3f s, concurrent %.3f s (%.2fx)"
          % (m, serial, concurrent, concurrent / serial))
    assert concurrent <= 0.45 * serial, (concurrent, serial)




@pytest.mark.parametrize("widths", [(122, 43, 29, 49), (64, 64, 64, 64, 10), (130, 200), (40, 50, 60, 70, 80, 90, 100, 110, 120, 128, 5, 64),
                                    (60, 50), (60, 50, 40), (100, 60, 30), (64, 64, 64, 64), (129, 3),
                                    # bit-packed layouts: eight members in four words, members that straddle word boundaries, exactly 256 bins,
                                    # three small targets in two words (the engine packs when that brings a group down to <= 4 words)
                                    (30, 30, 30, 30, 30, 30, 30, 30), (10, 200), (64, 1, 63, 128), (250, 6), (43, 29, 49), (70, 50)])
def test_filters_of_one_hash_geometry_share_a_merged_table(widths):
    """Filters built with one fragment_size have the same noOfBlocks whatever their bin count (IBFBuild.cpp:404-413), so a k-mer
    hashes to the same block in all of them: the engine merges their blocks into one table and serves every member with ONE
    gather per (k-mer, hash function).  Same maxima and decisions as the filters on their own (merge off) and as the oracle, for
    the reference's README shape (122 + 43/29/49 bins), five filters, two wide-ish ones, twelve filters (two groups: the
    16-word limit); N-containing and reverse-strand reads, long reads (16 counter planes), packed reads with on-GPU chunking; and
    the merged copy follows a member that changes."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(sum(widths))
    n_blocks = 30011
    ref = H.random_dna(rng, 60000)
    filters, views, keep = [], [], []
    for i, bins in enumerate(widths):
        W = (bins + 63) // 64
        d = capi.DeviceIBF.create(0, bins, 3, 13, W * 64 * n_blocks + int(rng.integers(0, 64 * W)))
        assert d.info["n_blocks"] == n_blocks
        d.fill_synth(100 + i)
        lo = (i * 4000) % 50000
        d.add_sequence(ref[lo:lo + 9000], 9000 // min(bins, 40) + 1)
        if i == 0:  # one long fragment in one bin: counts beyond 1023 (16 counter planes)
            d.insert(ref, np.array([3000], dtype=np.uint64), np.array([5400], dtype=np.uint64), np.array([0], dtype=np.uint64))
        filters.append(d)
    def views_now():
        vs, ks = [], []
        for d in filters:
            h = d.download()
            ks.append(h)
            vs.append(po.OracleIBF.wrap(h.info["n_bins"], 3, 13, h.info["n_bits"], h.words()))
        return vs, ks
    views, keep = views_now()
    nd = 1 if len(widths) != 2 else 1
    reads = make_reads(rng, ref, 2600, lo=5, hi=420, err=0.1, n_frac=0.2) + [ref[100:1700], ref[3000:5300], "", "ACGT", "N" * 300]
    buf, offs, lens = H.pack_reads(reads)
    eng = capi.Engine(0, filters[:nd], filters[nd:])
    exp_max = np.stack([po.batch_raw_max(v, buf, offs, lens, 8) for v in views], axis=1)
    exp_dec, exp_st = po.batch_check_unblock(views[:nd], views[nd:], buf, offs, lens, n_threads=8)
    results = {}
    # what merges: at most 16 words per merged block, so the twelve filters make a table of ten and a pair; tables as small as
    # these (L2-resident on their own and merged) always pay, so mode 1 merges what mode 2 does
    expect = {(122, 43, 29, 49): (1, 4), (64, 64, 64, 64, 10): (1, 5), (130, 200): (1, 2),
              (40, 50, 60, 70, 80, 90, 100, 110, 120, 128, 5, 64): (2, 12),
              (60, 50): (1, 2), (60, 50, 40): (1, 3), (100, 60, 30): (1, 3), (64, 64, 64, 64): (1, 4), (129, 3): (1, 2),
              (30, 30, 30, 30, 30, 30, 30, 30): (1, 8), (10, 200): (1, 2), (64, 1, 63, 128): (1, 4), (250, 6): (1, 2), (43, 29, 49): (1, 3),
              (70, 50): (1, 2)}[widths]
    # the layout the engine chose, as it reports it: bins side by side bit to bit when whole words would need more columns and
    # the packed block fits the four words of the one-lane builds
    packed_words = (sum(widths) + 63) // 64
    whole_words = sum((b + 63) // 64 for b in widths)
    if len(widths) <= 8 and packed_words <= 4 and packed_words < whole_words:
        stride = {1: 1, 2: 2, 3: 4, 4: 4}[packed_words]
        eng.set_merge(2)
        assert eng.merge_info()[2] == (n_blocks * stride + 8) * 8, (eng.merge_info(), packed_words)
        pl = eng.plan(0, len(lens), 400)
        assert pl["merged_members"] == len(widths) and pl["block_words"] == packed_words and pl["table_bytes"] == n_blocks * stride * 8
    for mode in (0, 1, 2):
        eng.set_merge(mode)
        assert eng.merge_info()[:2] == ((0, 0) if mode == 0 else expect), (mode, eng.merge_info())
        mc, best, dec, st = eng.classify(buf, offs, lens)  # > 2048 reads: throughput form
        assert np.array_equal(mc, exp_max), mode
        assert np.array_equal(dec, exp_dec) and np.array_equal(st, exp_st), mode
        results[mode] = (mc, best, dec, st)
    assert np.array_equal(results[0][1], results[2][1])
    assert len(set(exp_dec.tolist())) >= 2 and exp_max.max() > 1023
    # batches of short reads only: merged blocks of two to four words are then held by ONE lane of the both-strands builds of the
    # phased kernel (<= 256 and <= 512 k-mers), with and without clock phases (the table cut into as many as 8 slices)
    for hi in (250, 420):
        sreads = make_reads(rng, ref, 2300, lo=5, hi=hi, err=0.1, n_frac=0.2) + ["", "ACGT", "N" * hi, ref[100:100 + hi]]
        sbuf, soffs, slens = H.pack_reads(sreads)
        sexp = np.stack([po.batch_raw_max(v, sbuf, soffs, slens, 8) for v in views], axis=1)
        sdec, sst = po.batch_check_unblock(views[:nd], views[nd:], sbuf, soffs, slens, n_threads=8)
        for mode, forced in ((2, False), (2, True), (0, True)):
            eng.set_merge(mode)
            if forced:
                eng.set_phased(0, 1 << 40, 150, 0, 1)
                eng.set_phase_slices(1, 8)
            got = eng.classify(sbuf, soffs, slens)
            eng.set_phased()
            eng.set_phase_slices()
            assert np.array_equal(got[0], sexp), (hi, mode, forced)
            assert np.array_equal(got[2], sdec) and np.array_equal(got[3], sst), (hi, mode, forced)
    eng.set_merge(2)
    # micro-batch of the same engine (latency kernels, no merged table) still agrees
    sub = eng.classify(buf, offs[:300], lens[:300])
    assert np.array_equal(sub[0], exp_max[:300])
    # packed reads + on-GPU chunking through the merged kernel
    packed, p_off, nmask, n_off = capi.pack_reads(buf, offs, lens)
    dev = torch.device("cuda:0")
    up = lambda a, dt: torch.from_numpy(a.view(dt)).to(dev)
    t_pk, t_poff, t_nm, t_noff, t_lens = up(packed, np.uint8), up(p_off, np.int64), up(nmask, np.uint8), up(n_off, np.int64), up(lens, np.int32)
    n = len(reads)
    t_mc = torch.zeros((n, len(widths)), dtype=torch.int16, device=dev)
    torch.cuda.synchronize()
    eng.classify_device_ex(t_pk.data_ptr(), t_poff.data_ptr(), t_lens.data_ptr(), n, int(lens.max()), d_nmask=t_nm.data_ptr(),
                           d_nmask_offsets=t_noff.data_ptr(), chunk_start=7, chunk_length=250, d_maxcount=t_mc.data_ptr())
    torch.cuda.synchronize()
    frags = [r[7:257] if len(r) >= 7 else "" for r in reads]
    fb, fo, fl = H.pack_reads(frags)
    ok = np.array([len(r) >= 7 for r in reads])
    exp_chunk = np.stack([po.batch_raw_max(v, fb, fo, fl, 8) for v in views], axis=1)
    assert np.array_equal(t_mc.cpu().numpy().view(np.uint16)[ok], exp_chunk[ok])
    # a member changes: the merged copy is made again
    filters[-1].add_sequence(ref[55000:59000], 4000 // min(widths[-1], 40) + 1)
    views, keep = views_now()
    exp2 = np.stack([po.batch_raw_max(v, buf, offs, lens, 8) for v in views], axis=1)
    assert not np.array_equal(exp2, exp_max)
    assert np.array_equal(eng.classify(buf, offs, lens)[0], exp2)
    # the other candidate of the N rule goes through the merged kernel as well
    prev = po.set_revcomp_of_n(4)
    try:
        eng.set_revcomp_of_n(4)
        exp4 = np.stack([po.batch_raw_max(v, buf, offs, lens, 8) for v in views], axis=1)
        assert np.array_equal(eng.classify(buf, offs, lens)[0], exp4) and not np.array_equal(exp4, exp2)
    finally:
        po.set_revcomp_of_n(prev)


def test_a_merged_copy_beyond_the_cap_is_not_made(monkeypatch):
    """The merged copy costs HBM beside its members: a group whose copy would exceed RB_MERGE_MAX_BYTES (default 16 GiB) is not
    made -- a smaller group of the remaining filters may be -- and what stays apart is served by the per-filter kernels: same
    results."""
    rng = np.random.default_rng(77)
    n_blocks = 20011
    ref = H.random_dna(rng, 30000)
    filters, views, keep = [], [], []
    for i, bins in enumerate((100, 30, 30, 30)):
        W = (bins + 63) // 64
        d = capi.DeviceIBF.create(0, bins, 3, 13, W * 64 * n_blocks)
        d.fill_synth(7 + i)
        d.add_sequence(ref[i * 5000:i * 5000 + 8000], 300)
        filters.append(d)
        h = d.download()
        keep.append(h)
        views.append(po.OracleIBF.wrap(bins, 3, 13, h.info["n_bits"], h.words()))
    reads = make_reads(rng, ref, 2300, lo=20, hi=400, err=0.08, n_frac=0.1)
    buf, offs, lens = H.pack_reads(reads)
    exp = np.stack([po.batch_raw_max(v, buf, offs, lens, 8) for v in views], axis=1)
    copy_bytes = (n_blocks * 4 + 8) * 8  # 100 + 30 + 30 + 30 bins side by side, bit to bit: three words, padded to four (whole words: five -> eight)
    small_copy = (n_blocks * 2 + 8) * 8  # the three 30-bin targets alone: 90 bins in two words
    for cap, expect in ((copy_bytes, (1, 4, copy_bytes)), (copy_bytes - 1, (1, 3, small_copy)), (1000, (0, 0, 0))):
        monkeypatch.setenv("RB_TUNING_ENV", "1")  # environment switches are read by measurement processes only
        monkeypatch.setenv("RB_MERGE_MAX_BYTES", str(cap))
        eng = capi.Engine(0, filters[:1], filters[1:])
        assert eng.merge_info() == expect
        assert np.array_equal(eng.classify(buf, offs, lens)[0], exp)
        eng.destroy()


@pytest.mark.parametrize("shape,merged", [
    ([(122, 20.0), (43, 10.4), (29, 10.4), (49, 10.4)], True),   # the reference's README shape: 33.6 against 25.5 ms estimated
    ([(40, 10.4), (50, 10.4), (60, 10.4)], True),               # three one-word filters of 10 MiB: one three-word table, one lane per block (1.2 x)
    ([(64, 12.0), (64, 12.0)], True),                          # one two-word table through the two-word phased kernel (1.4 x)
    ([(128, 32.0), (128, 32.0)], False),                       # two two-word filters of 32 MiB: the merged table is beyond the one-lane builds (0.93 x)
    ([(64, 24.0), (64, 24.0), (64, 24.0)], True),               # measured 1.24 x
    ([(64, 160.0), (128, 320.0)], True),                        # beyond the phased range each sits at the request wall: 2 x
    ([(64, 1.0), (64, 1.0)], True),                             # the merged copy still fits an L2: 1.65 x
])
def test_merge_when_it_pays(shape, merged):
    """Mode 1 of rb_engine_set_merge compares K1 time estimates (rb_engine.hip, est_filter_ms): the members one after the other
    against one pass over the merged table.  The cases are the measured ones of profiles/r03/merged_tables.txt."""
    n_blocks = None
    filters = []
    for bins, mib in shape:
        W = (bins + 63) // 64
        if n_blocks is None:
            n_blocks = int(mib * (1 << 20) / (8 * W))
        filters.append(capi.DeviceIBF.create(0, bins, 3, 13, W * 64 * n_blocks))
    eng = capi.Engine(0, filters[:1], filters[1:])
    assert eng.merge_info()[:2] == ((1, len(shape)) if merged else (0, 0)), eng.merge_info()
    eng.set_merge(2)
    assert eng.merge_info()[:2] == (1, len(shape))
    eng.destroy()
    for f in filters:
        f.free()


HANDLE_SWEEP = r"""
import ctypes as C, sys
import numpy as np
from readbouncer_amd import capi
L = capi.lib()
d = capi.DeviceIBF.create(0, 100, 3, 13, 128 * 4099)
eng = capi.Engine(0, [d], [])
live = capi.Live(eng, 0.1, 0.95, 2000)
handles = {"rb_engine_": eng.h, "rb_classify_": eng.h, "rb_decide_": eng.h, "rb_replay_": eng.h, "rb_dibf_": d.h, "rb_live_": live.h}
skip = {"rb_engine_create", "rb_engine_destroy", "rb_dibf_create", "rb_dibf_upload", "rb_dibf_open", "rb_dibf_free", "rb_live_create",
        "rb_live_destroy", "rb_dibf_clone_to", "rb_dibf_clone_to_ex"}
must_refuse = {"rb_classify_batch", "rb_classify_batch_ptrs", "rb_classify_batch_device", "rb_classify_batch_device_ex", "rb_decide_device",
               "rb_decide_device_parts", "rb_live_process", "rb_replay_arrivals", "rb_live_replay_arrivals", "rb_dibf_insert",
               "rb_dibf_add_sequence", "rb_dibf_download", "rb_dibf_resize_bins", "rb_dibf_compare", "rb_dibf_get_info"}
n_called = 0
for name, (restype, argtypes) in sorted(capi.SIGNATURES.items()):
    h = next((v for k, v in handles.items() if name.startswith(k)), None)
    if h is None or name in skip:
        continue
    args = [h]
    for t in argtypes[1:]:
        if t is C.c_double:
            args.append(0.1)
        elif t is C.c_size_t:
            args.append(5)            # five items behind NULL buffers
        elif t in (C.c_int, C.c_uint8, C.c_uint16, C.c_uint32, C.c_uint64):
            args.append(0)
        else:
            args.append(None)
    r = getattr(L, name)(*args)
    n_called += 1
    if name in must_refuse and r == 0:
        print("ACCEPTED", name); sys.exit(3)
# the engine still works afterwards
buf = np.frombuffer(b"ACGTACGTACGTACGTACGTACGTACGTACGT", dtype=np.uint8).copy()
mc, best, dec, st = eng.classify(buf, np.array([0], dtype=np.uint64), np.array([32], dtype=np.uint32))
assert st[0] == 0
print("CALLED", n_called)
"""


def test_c_abi_refuses_null_buffers_behind_valid_handles():
    """Every engine / filter / live-step entry point called with a VALID handle, five items and NULL for every buffer: no crash (child
    process), the batch and build functions refuse, and the engine classifies normally afterwards."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", HANDLE_SWEEP], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert "CALLED" in r.stdout


def test_engines_of_one_device_share_the_merged_copy():
    """The reference's N classification threads (adaptive_sampling.hpp:745-751) are N engines over the same filters: they gather from
    ONE merged table (a registry hands the copy out; K copies of a 40 MB table evicted each other's slices from the L2s), an
    insert into a member is followed by every engine, and an engine that merges the same filters in ANOTHER order has its own."""
    import threading
    import torch
    rng = np.random.default_rng(4242)
    ref = H.random_dna(rng, 40000)
    n_blocks = 4_000_037  # 4 words x 8 B x 4 M blocks = 128 MB per merged copy: far above what four engines allocate for themselves
    filters, views, keep = [], [], []

    def view_all():
        views.clear()
        keep.clear()
        for d in filters:
            h = d.download()
            keep.append(h)
            views.append(po.OracleIBF.wrap(h.info["n_bins"], 3, 13, h.info["n_bits"], h.words()))

    for i, bins in enumerate((122, 43, 29, 49)):
        W = (bins + 63) // 64
        d = capi.DeviceIBF.create(0, bins, 3, 13, W * 64 * n_blocks)
        d.add_sequence(ref[i * 6000:i * 6000 + 9000], 400)
        filters.append(d)
    view_all()
    reads = make_reads(rng, ref, 2600, lo=20, hi=250, err=0.06)
    buf, offs, lens = H.pack_reads(reads)
    exp = np.stack([po.batch_raw_max(v, buf, offs, lens, 8) for v in views], axis=1)
    copy_bytes = (n_blocks * 4 + 8) * 8
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(0)[0]
    engines = [capi.Engine(0, filters[:1], filters[1:]) for _ in range(4)]
    for e in engines:
        assert e.merge_info() == (1, 4, copy_bytes)
        assert np.array_equal(e.classify(buf, offs, lens)[0], exp)
    used = free0 - torch.cuda.mem_get_info(0)[0]
    assert copy_bytes <= used < 2 * copy_bytes, (used, copy_bytes)  # one copy (+ the engines' small staging buffers), not four
    # the same filters merged in another order: another block layout, another copy
    other = capi.Engine(0, filters[1:2], [filters[0]] + filters[2:])
    assert np.array_equal(other.classify(buf, offs, lens)[0], exp[:, [1, 0, 2, 3]])
    used2 = free0 - torch.cuda.mem_get_info(0)[0]
    assert used2 - used >= copy_bytes - (8 << 20), (used, used2)
    # an insert into a member while the engines classify on threads of their own: every engine ends on the new bits
    errors = []

    def worker(e):
        try:
            for _ in range(12):
                e.classify(buf, offs, lens)
        except Exception as ex:  # noqa: BLE001
            errors.append(repr(ex))
    threads = [threading.Thread(target=worker, args=(e,)) for e in engines]
    for t in threads:
        t.start()
    filters[2].add_sequence(ref[30000:38000], 300)
    for t in threads:
        t.join()
    assert errors == []
    view_all()
    exp2 = np.stack([po.batch_raw_max(v, buf, offs, lens, 8) for v in views], axis=1)
    assert not np.array_equal(exp, exp2)
    for e in engines:
        assert np.array_equal(e.classify(buf, offs, lens)[0], exp2)
    assert np.array_equal(other.classify(buf, offs, lens)[0], exp2[:, [1, 0, 2, 3]])
    # the copy goes with its last engine
    for e in engines + [other]:
        e.destroy()
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info(0)[0] < copy_bytes // 2
    for d in filters:
        d.free()


def test_equal_length_slices_through_the_planner():
    """A four-word table that 4 MiB slices cut wastefully (37.7 MiB: ten slices, the last one 1.7 MiB) is walked in eight equal slices of
    4.7 MiB (rb_phase_plan.h, phase_equal_slices): the plan says so, and the maxima equal the oracle's and those of the 4 MiB
    slices, for the four-tile build (250 bp) and the rounds of three tiles (360 bp), with blocks at both ends of every slice in play."""
    rng = np.random.default_rng(8088)
    ref = H.random_dna(rng, 60000)
    n_blocks = 1_236_269  # x 32 B = 37.73 MiB, the README shape's merged table
    d = capi.DeviceIBF.create(0, 250, 3, 13, 256 * n_blocks)
    d.fill_synth(21)
    d.add_sequence(ref, 250)
    o, _keep = oracle_view(d)
    eng = capi.Engine(0, [d], [])
    for L, shape in ((250, "wide, four tiles"), (360, "wide, rounds of three tiles")):
        reads = make_reads(np.random.default_rng(L), ref, 4200, lo=L - 60, hi=L + 1, err=0.05)  # (>= 4096: the phased form of a table this size)
        buf, offs, lens = H.pack_reads(reads)
        exp = po.batch_raw_max(o, buf, offs, lens, 8)
        pl = eng.plan(0, len(reads), L)
        assert pl["phased"] == 1 and pl["phase_shape_name"].startswith(shape), pl
        assert pl["phase_slices"] == 8 and pl["phase_slice_log2"] == 22, pl
        assert pl["phase_slice_bytes"] == -(-n_blocks // 8) * 32 and pl["phase_slice_bytes"] & (pl["phase_slice_bytes"] - 1), pl
        got = eng.classify(buf, offs, lens)[0][:, 0]
        assert np.array_equal(got, exp)
        eng.set_phase_slices(22, 32)  # the slices of 2^22 bytes they replace
        p2 = eng.plan(0, len(reads), L)
        assert p2["phase_slices"] == 10 and p2["phase_slice_bytes"] == 1 << 22
        assert np.array_equal(eng.classify(buf, offs, lens)[0][:, 0], exp)
        eng.set_phase_slices()
    eng.destroy()
    d.free()


def test_equal_length_slices_of_large_one_word_tables():
    """One-word tables from 50 MiB on are walked in fewer, equal-length slices that are LONGER than an L2 (rb_phase_plan.h,
    phase_equal_slices_one_word: 56 MiB -> 11 slices of 5.1 MiB instead of 14 of 4 MiB): the plan says so, the maxima equal the
    oracle's and those of the 4 MiB slices, for the four-tile build (250 bp) and the six-tile build (360 bp); 48 MiB keeps 4 MiB slices."""
    rng = np.random.default_rng(8090)
    ref = H.random_dna(rng, 60000)
    n_blocks = 56 * (1 << 20) // 8 - 5  # one-word blocks: 56 MiB
    d = capi.DeviceIBF.create(0, 64, 3, 13, 64 * n_blocks)
    d.fill_synth(23)
    d.add_sequence(ref, 1000)
    o, _keep = oracle_view(d)
    eng = capi.Engine(0, [d], [])
    for L, shape in ((250, "four tiles"), (360, "six tiles")):
        reads = make_reads(np.random.default_rng(L + 1), ref, 4200, lo=L - 60, hi=L + 1, err=0.05)
        buf, offs, lens = H.pack_reads(reads)
        exp = po.batch_raw_max(o, buf, offs, lens, 8)
        pl = eng.plan(0, len(reads), L)
        assert pl["phased"] == 1 and pl["phase_shape_name"].startswith(shape), pl
        assert pl["phase_slices"] == 11 and pl["phase_slice_bytes"] == -(-n_blocks // 11) * 8 and pl["phase_slice_bytes"] > (4 << 20), pl
        assert np.array_equal(eng.classify(buf, offs, lens)[0][:, 0], exp)
        eng.set_phase_slices(22, 32)  # the slices of 2^22 bytes they replace
        p2 = eng.plan(0, len(reads), L)
        assert p2["phase_slices"] == 14 and p2["phase_slice_bytes"] == 1 << 22
        assert np.array_equal(eng.classify(buf, offs, lens)[0][:, 0], exp)
        eng.set_phase_slices()
    eng.destroy()
    d.free()
    small = capi.DeviceIBF.create(0, 64, 3, 13, 64 * (48 * (1 << 20) // 8 - 5))
    e2 = capi.Engine(0, [small], [])
    p3 = e2.plan(0, 5000, 250)
    assert p3["phased"] == 1 and p3["phase_slices"] == 12 and p3["phase_slice_bytes"] == 1 << 22, p3
    e2.destroy()
    small.free()


def test_large_tables_are_placed_by_trial():
    """A table of 1 GiB and more is allocated two to five times, every candidate probed with random whole-block gathers, the best kept
    (rb_set_placement_tries; profiles/r05/placement_*.txt: the same table gathers 1.7-2.9 % slower or faster from one allocation to the
    next).  Nothing but the address depends on it: the filter round-trips and classifies like its oracle view; tries = 1 switches it off;
    small tables are never tried; a device on which an engine of the process is alive is never tried either (a process that is already
    classifying is not stalled by probe launches and transient copies) -- which is why the checks run in a process of their own: engines
    that earlier tests of this process created and never destroyed would count."""
    code = "import sys; sys.path.insert(0, %r); from tests.test_gpu_parity import placement_checks; placement_checks(); print('placement ok')" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and "placement ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def placement_checks():
    rng = np.random.default_rng(99)
    ref = H.random_dna(rng, 30000)
    n_blocks = (1200 << 20) // 1024  # 8192 bins: 1 KiB blocks, 1.17 GiB
    d = capi.DeviceIBF.create(0, 8192, 3, 13, 8192 * n_blocks)
    tries, kept, worst = d.placement()
    assert 2 <= tries <= 5 and kept >= worst > 1000.0, (tries, kept, worst)  # (GB/s of an MI355X, not a stopwatch assertion)
    d.add_sequence(ref, 100)
    o, _keep = oracle_view(d)
    reads = make_reads(rng, ref, 300, lo=100, hi=400)
    buf, offs, lens = H.pack_reads(reads)
    eng = capi.Engine(0, [d], [])
    assert np.array_equal(eng.classify(buf, offs, lens)[0][:, 0], po.batch_raw_max(o, buf, offs, lens, 8))
    cost = d.placement_cost()
    assert cost["skipped"] == 0 and cost["trial_s"] > 0.0 and cost["peak_bytes"] == tries * (8192 // 8 * n_blocks + 64), cost
    # a device that is already classifying is not stalled by probe launches and transient copies: with an engine alive the clone takes
    # the first allocation, and says why (rb_dibf_placement_cost)
    clone, _, _ = d.clone_to_ex(0)
    assert clone.placement()[0] == 0 and clone.placement_cost()["skipped"] == 1
    cmp_ = d.compare(clone)
    assert cmp_["new_bits"] == 0 and cmp_["file_bits"] == cmp_["rebuilt_bits"]
    clone.free()
    eng.destroy()
    clone, _, _ = d.clone_to_ex(0)  # no engine on the device any more: placed by trial again
    assert clone.placement()[0] >= 2 and clone.placement_cost()["skipped"] == 0
    clone.free()
    capi.set_placement_tries(1)
    try:
        off = capi.DeviceIBF.create(0, 8192, 3, 13, 8192 * n_blocks)
        assert off.placement() == (0, 0.0, 0.0)
        off.free()
    finally:
        capi.set_placement_tries(5)
    small = capi.DeviceIBF.create(0, 8192, 3, 13, 8192 * 4099)
    assert small.placement()[0] == 0
    small.free()
    with pytest.raises(capi.RBError):
        capi.set_placement_tries(9)
    d.free()


def test_micro_batches_enter_hbm_by_the_engines_own_copy_kernel(monkeypatch):
    """Micro-batches of up to 1 MiB are copied from the pinned host block by a kernel of the call's stream instead of the runtime's copy
    (profiles/r05/micro_copy_ab.txt: about 5 us less from 64 reads on).  Same bytes either way: every output of ragged batches around the
    16-byte units of the copy equals what the runtime's copy gives (an engine created with the measurement switch set to 0) and the oracle."""
    rng = np.random.default_rng(4242)
    ref = H.random_dna(rng, 30000)
    d = capi.DeviceIBF.create(0, 600, 3, 13, 640 * 40009)
    d.add_sequence(ref, 500)
    t = capi.DeviceIBF.create(0, 64, 3, 13, 64 * 100003)
    t.add_sequence(ref[10000:], 800)
    od, _k1 = oracle_view(d)
    ot, _k2 = oracle_view(t)
    reads = make_reads(rng, ref, 1500, lo=1, hi=420, err=0.08, n_frac=0.2)
    kernel_copy = capi.Engine(0, [d], [t])
    monkeypatch.setenv("RB_TUNING_ENV", "1")
    monkeypatch.setenv("RB_MICRO_COPY_KERNEL_BYTES", "0")
    runtime_copy = capi.Engine(0, [d], [t])
    monkeypatch.delenv("RB_MICRO_COPY_KERNEL_BYTES")
    for n in (1, 2, 3, 5, 17, 64, 333, 1500):
        buf, offs, lens = H.pack_reads(reads[:n])
        a = kernel_copy.classify(buf, offs, lens)
        b = runtime_copy.classify(buf, offs, lens)
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), n
    buf, offs, lens = H.pack_reads(reads[:300])
    got = kernel_copy.classify(buf, offs, lens)
    exp_dec, exp_st = po.batch_check_unblock([od], [ot], buf, offs, lens, n_threads=8)
    assert np.array_equal(got[2], exp_dec) and np.array_equal(got[3], exp_st)
    assert np.array_equal(got[0][:, 0], po.batch_raw_max(od, buf, offs, lens, 8)) and np.array_equal(got[0][:, 1], po.batch_raw_max(ot, buf, offs, lens, 8))
    kernel_copy.destroy()
    runtime_copy.destroy()
    d.free()
    t.free()


@pytest.mark.parametrize("setup", ["one_narrow", "one_wide", "wide_plus_narrow", "deplete_plus_three_targets", "targets_only"])
def test_latency_kernel_makes_the_decisions_itself(setup):
    """Micro-batches of a one-filter engine in the latency form: the workgroup that writes a read's raw maximum runs the decision for
    it (FoldJob, rb_engine_set_fold_decide) -- same maxima, best targets, decisions and statuses as with the decision kernel behind K1
    and as the oracle's check_unblock, in all three modes, with one and with several workgroups per read; engines with several filters
    keep the two launches whatever the switch says (the same calls, the same answers)."""
    rng = np.random.default_rng(len(setup) * 31)
    ref = H.random_dna(rng, 50000)

    def filt(n_bins, n_blocks, seq, frag):
        d = capi.DeviceIBF.create(0, n_bins, 3, 13, ((n_bins + 63) // 64) * 64 * n_blocks)
        d.fill_synth(n_bins)
        d.add_sequence(seq, frag)
        o, h = oracle_view(d)
        return d, o, h

    narrow = lambda i: filt(40 + 11 * i, 30011 + 2 * i, ref[10000 * i:10000 * (i + 1)], 10000 // (40 + 11 * i) + 1)
    wide = lambda i: filt(8192 - 64 * i, 211 + i, ref[10000 * i:10000 * (i + 2)], 20)
    if setup == "one_narrow":
        dep, tgt = [narrow(0)], []
    elif setup == "one_wide":
        dep, tgt = [wide(0)], []
    elif setup == "wide_plus_narrow":
        dep, tgt = [wide(0)], [narrow(2)]
    elif setup == "deplete_plus_three_targets":
        dep, tgt = [wide(1)], [narrow(0), filt(600, 997, ref[20000:30000], 20), narrow(3)]
    else:
        dep, tgt = [], [narrow(1), wide(3)]
    eng = capi.Engine(0, [x[0] for x in dep], [x[0] for x in tgt])
    od, ot = [x[1] for x in dep], [x[1] for x in tgt]
    reads = make_reads(rng, ref, 150, lo=5, hi=600) + ["", "ACGT", "N" * 40, ref[100:460], ref[25000:25360]]
    buf, offs, lens = H.pack_reads(reads)
    exp_mc = np.stack([po.batch_raw_max(o, buf, offs, lens, 4) for o in od + ot], axis=1)
    exp_dec, exp_st = po.batch_check_unblock(od, ot, buf, offs, lens, n_threads=4)
    assert len(set(exp_dec.tolist())) >= 2
    for mode in (capi.RB_MODE_CHECK_UNBLOCK, capi.RB_MODE_CLASSIFY_CHUNK, capi.RB_MODE_CLASSIFY_ANY):
        for parts in ((8, 4), (1, 1)):
            eng.set_split_parts(*parts)
            for n_sub in (len(reads), 1, 7, len(reads)):
                eng.set_fold_decide(False)
                two = eng.classify(buf, offs[:n_sub], lens[:n_sub], mode=mode)
                eng.set_fold_decide(True)
                for _ in range(2):
                    one = eng.classify(buf, offs[:n_sub], lens[:n_sub], mode=mode)
                    for a, b in zip(one, two):
                        assert np.array_equal(a, b), (setup, mode, parts, n_sub)
                assert np.array_equal(one[0], exp_mc[:n_sub])
                if mode == capi.RB_MODE_CHECK_UNBLOCK:
                    assert np.array_equal(one[2], exp_dec[:n_sub]) and np.array_equal(one[3], exp_st[:n_sub])
    # an error rate of its own per call (another threshold table) and reads longer than the first call announced
    long_reads = reads[:20] + [ref[:1500], ref[30000:32500]]
    b2, o2, l2 = H.pack_reads(long_reads)
    for r in (0.05, 0.15, 0.1):
        e_dec, e_st = po.batch_check_unblock(od, ot, b2, o2, l2, r=r, n_threads=4)
        got = eng.classify(b2, o2, l2, error_rate=r)
        assert np.array_equal(got[2], e_dec) and np.array_equal(got[3], e_st), r


@pytest.mark.parametrize("nd,nt", [(1, 0), (1, 1), (1, 3), (0, 2)])
def test_completion_word_of_the_micro_batch_path(nd, nt):
    """rb_classify_batch with up to 2 048 reads returns when the word the last kernel stores has arrived, not when the stream is idle
    (rb_engine_set_completion_word): same results as with the stream's wait at every batch size -- one read (the folded launch announces),
    up to 256 (one workgroup of the decision kernel), above (several workgroups, arrival counter) -- in all three modes, call after call
    without a pause, and the results of a call are complete when it returns (the output block is overwritten by the next call at once)."""
    rng = np.random.default_rng(100 * nd + nt)
    ref = H.random_dna(rng, 60000)
    fs = []
    for i in range(nd + nt):
        n_bins = (8192, 43, 600, 122)[i % 4]
        d = capi.DeviceIBF.create(0, n_bins, 3, 13, ((n_bins + 63) // 64) * 64 * (30011 if n_bins < 1000 else 211))
        d.fill_synth(i + 1)
        d.add_sequence(ref[15000 * i:15000 * (i + 1)], max(20, 15000 // n_bins + 1))
        fs.append((d,) + oracle_view(d))
    dep, tgt = fs[:nd], fs[nd:]
    eng = capi.Engine(0, [x[0] for x in dep], [x[0] for x in tgt])
    reads = make_reads(rng, ref, 2100, lo=5, hi=420)
    buf, offs, lens = H.pack_reads(reads)
    exp_dec, exp_st = po.batch_check_unblock([x[1] for x in dep], [x[1] for x in tgt], buf, offs, lens, n_threads=8)
    exp_mc = np.stack([po.batch_raw_max(x[1], buf, offs, lens, 8) for x in fs], axis=1)
    for mode in (capi.RB_MODE_CHECK_UNBLOCK, capi.RB_MODE_CLASSIFY_CHUNK, capi.RB_MODE_CLASSIFY_ANY):
        for n_sub in (1, 2, 64, 256, 257, 700, 2048, 2100, 1):
            eng.set_completion_word(False)
            ref_out = eng.classify(buf, offs[:n_sub], lens[:n_sub], mode=mode)
            eng.set_completion_word(True)
            for fold in (True, False):
                eng.set_fold_decide(fold)
                for _ in range(3):
                    out = eng.classify(buf, offs[:n_sub], lens[:n_sub], mode=mode)
                    for a, b in zip(out, ref_out):
                        assert np.array_equal(a, b), (mode, n_sub, fold)
            if mode == capi.RB_MODE_CHECK_UNBLOCK:
                assert np.array_equal(out[0], exp_mc[:n_sub]) and np.array_equal(out[2], exp_dec[:n_sub]) and np.array_equal(out[3], exp_st[:n_sub])
    # calls of changing size back to back: each returns its own results (a late or early word would hand out the neighbour's)
    sizes = rng.integers(1, 300, size=400)
    starts = rng.integers(0, len(reads) - 300, size=400)
    for n_sub, s in zip(sizes, starts):
        o2 = (offs[s:s + n_sub] - offs[s]).astype(np.uint64)
        b2 = buf[int(offs[s]):int(offs[s + n_sub - 1] + lens[s + n_sub - 1])]
        out = eng.classify(b2, o2, lens[s:s + n_sub])
        assert np.array_equal(out[0], exp_mc[s:s + n_sub]) and np.array_equal(out[2], exp_dec[s:s + n_sub]) and np.array_equal(out[3], exp_st[s:s + n_sub])


@pytest.mark.parametrize("n_bins,n_blocks", [(8192, 150001), (600, 1700003)])
def test_large_filter_files_stream_into_hbm(tmp_path, n_bins, n_blocks):
    """IBF::load_filter at a size where the loader's machinery is in play (several 64 MiB chunks, several reader threads, the padded layout
    widened on the device): rb_dibf_open and rb_ibf_open + rb_dibf_upload give the device image the file describes, bit for bit."""
    W = (n_bins + 63) // 64
    n_bits = n_blocks * W * 64 + 29
    d = capi.DeviceIBF.create(0, n_bins, 3, 13, n_bits)
    d.fill_synth(n_bins)
    host = d.download()
    want = host.words().copy()
    p = tmp_path / "f.ibf"
    host.store(str(p))
    assert os.path.getsize(p) > (128 << 20)  # three chunks of the staged copy
    a = capi.DeviceIBF.open(0, str(p))
    ha = a.download()
    assert np.array_equal(ha.words(), want)
    h = capi.HostIBF.open(str(p))
    nw = n_bits // 64  # (the words behind carry the metadata block in a file image, zeros in a downloaded one)
    assert np.array_equal(h.words()[:nw], want[:nw])
    b = capi.DeviceIBF.upload(0, h)
    hb = b.download()
    assert np.array_equal(hb.words(), want)
    # and the opened filter classifies like the one it was stored from
    rng = np.random.default_rng(3)
    reads = [H.random_dna(rng, 300) for _ in range(64)]
    buf, offs, lens = H.pack_reads(reads)
    r0 = capi.Engine(0, [d], []).classify(buf, offs, lens)
    r1 = capi.Engine(0, [a], []).classify(buf, offs, lens)
    r2 = capi.Engine(0, [b], []).classify(buf, offs, lens)
    for x, y, z in zip(r0, r1, r2):
        assert np.array_equal(x, y) and np.array_equal(x, z)


def test_pool_replicas_of_a_large_filter_started_side_by_side():
    """Replicas of a table of 1 GiB (placed by trial) for three workers, each started from a thread of its own -- what a pool over several
    GPUs does, one thread per GPU; here, on one GPU, by the testing build's switch -- give a pool that classifies like a single engine."""
    import subprocess
    child = ("import os, sys, numpy as np\n"
             "sys.path.insert(0, %r)\n"
             "from readbouncer_amd import capi\n"
             "from tests import helpers as H\n"
             "d = capi.DeviceIBF.create(0, 8192, 3, 13, 1 << 33)\n"
             "d.fill_synth(3)\n"
             "rng = np.random.default_rng(8)\n"
             "ref = H.random_dna(rng, 40000)\n"
             "d.add_sequence(ref, 1000)\n"
             "reads = [H.mutate(rng, ref[s:s + 360], 0.05) if i %% 2 else H.random_dna(rng, 360) for i, s in enumerate(rng.integers(0, 39000, size=3000))]\n"
             "buf, offs, lens = H.pack_reads(reads)\n"
             "want = capi.Engine(0, [d], []).classify(buf, offs, lens)\n"
             "pool = capi.Pool.from_device([0, 0, 0], [d], [])\n"
             "assert pool.size() == 3\n"
             "pool.set_min_split(200)\n"
             "got = pool.classify(buf, offs, lens)\n"
             "same = [bool(np.array_equal(a, b)) for a, b in zip(got, want)]\n"
             "print('same', same, 'decisions', np.bincount(want[2], minlength=3).tolist(), 'differing reads', np.nonzero(got[0][:, 0] != want[0][:, 0])[0][:10].tolist())\n"
             "assert all(same) and len(set(want[2].tolist())) == 2\n"
             "pool.destroy()\n"
             "print('side by side ok', d.placement()[0])\n") % (ROOT,)
    testing_lib = os.path.join(ROOT, "readbouncer_amd", "libreadbouncer_amd_testing.so")
    env = dict(os.environ, RB_POOL_TEST_THREAD_PER_WORKER="1", RB_AMD_LIBRARY=testing_lib)
    r = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0 and "side by side ok" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.parametrize("widths,n_blocks", [((100,), 2053), ((128,), 4096), ((65,), 30011), ((60, 50), 30011), ((43, 29, 49), 30011), ((64, 64), 8192),
                                             # blocks of three and four words (stride 4): filters on their own, the README shape's packed table, a triple
                                             ((130,), 2053), ((256,), 4096), ((200,), 30011), ((122, 43, 29, 49), 30011), ((100, 60, 30), 30011),
                                             ((64, 64, 64, 64), 8192),
                                             # one-word blocks (a filter of up to 64 bins on its own); the last two: more than 2^21 blocks -- 22-bit
                                             # block numbers whose third one keeps its top two bits in the spill word (Barrett and mask modulus)
                                             ((40,), 4099), ((64,), 4096), ((64,), 2470013), ((33,), 1 << 21), ((64,), (1 << 22) - 2)])
def test_several_reads_per_wave_match_oracle(widths, n_blocks):
    """Round 6: the builds of the phased kernel that keep a read's block numbers packed in LDS and carry one or two reads per wave
    through a pass of the windows (rb_engine_set_reads_per_wave; ibf_count_max_phased_multi_kernel) -- blocks of two, three and four words, reads of up to
    256 k-mers (four tiles per strand) and up to 384 (six tiles).
    Raw maxima against the oracle and decisions against the one-read build: a filter on its own (Barrett and mask modulus) and merged
    pairs / triples of small targets, odd batch sizes (the last wave has a read too few), empty / short / all-N reads, windows from
    far too short to far too long, the table cut into 1 to 32 slices (the last one ending with the table)."""
    rng = np.random.default_rng(sum(widths) + n_blocks)
    ref = H.random_dna(rng, 40000)
    filters = []
    for i, bins in enumerate(widths):
        W = (bins + 63) // 64
        d = capi.DeviceIBF.create(0, bins, 3, 13, W * 64 * n_blocks + int(rng.integers(0, 64 * W)))
        assert d.info["n_blocks"] == n_blocks
        d.fill_synth(300 + i)
        lo = (i * 7000) % 30000
        d.add_sequence(ref[lo:lo + 8000], 8000 // min(bins, 40) + 1)
        filters.append(d)
    views, keep = [], []
    for d in filters:
        h = d.download()
        keep.append(h)
        views.append(po.OracleIBF.wrap(h.info["n_bins"], 3, 13, h.info["n_bits"], h.words()))
    nd = 1 if len(widths) == 1 else 0
    eng = capi.Engine(0, filters[:nd], filters[nd:])
    eng.set_merge(2)
    for hi in (268, 396):  # four tiles per strand (up to 256 k-mers) and six (up to 384: 360 bp prefixes); k = 13
        reads = make_reads(rng, ref, 2299, lo=5, hi=hi, err=0.1, n_frac=0.2) + ["", "ACGT", "N" * hi, ref[100:100 + hi], "A" * 13, "A" * 12]
        assert len(reads) % 2 == 1
        buf, offs, lens = H.pack_reads(reads)
        exp = np.stack([po.batch_raw_max(v, buf, offs, lens, 8) for v in views], axis=1)
        assert exp.max() > 100
        base = None
        # (+ 16: merged tables keep the AND form instead of the complemented twin's OR form)
        for reads_per_wave in (0, 1, 2, 16 + 1, 16 + 2):
            eng.set_reads_per_wave(reads_per_wave)
            for base_ticks, max_slices in ((1, 8), (300, 32), (2000, 3), (150, 1)):
                eng.set_phased(0, 1 << 40, base_ticks, 0, 1)
                eng.set_phase_slices(1, max_slices)
                pl = eng.plan(0, len(lens), int(lens.max()))
                assert pl["kernel"] == ("ibf_count_max_phased_multi_kernel" if reads_per_wave else "ibf_count_max_phased_kernel"), pl
                assert pl["block_words"] == (sum(widths) + 63) // 64 and pl["reserved0"] == (reads_per_wave & 3)
                got = eng.classify(buf, offs, lens)
                assert np.array_equal(got[0], exp), (reads_per_wave, base_ticks, max_slices)
                if base is None:
                    base = got
                assert all(np.array_equal(a, b) for a, b in zip(got, base)), (reads_per_wave, base_ticks, max_slices)
            # slices of equal length, any number of blocks each (rb_engine_set_phase_equal_slices; what the planner's rule cuts real tables
            # into): 3, 7 and 31 of them -- the last one shorter, ending with the table
            eng.set_phase_slices(0, 32)
            for base_ticks, n_equal in ((200, 3), (1, 7), (500, 31)):
                eng.set_phased(0, 1 << 40, base_ticks, 0, 1)
                eng.set_phase_equal_slices(n_equal)
                pl = eng.plan(0, len(lens), int(lens.max()))
                assert pl["phased"] and pl["phase_slices"] == -(-n_blocks // -(-n_blocks // n_equal)) and pl["phase_slice_bytes"] % 8 == 0, pl
                assert pl["phase_slice_bytes"] == -(-n_blocks // n_equal) * 8 * pl["stride_words"], pl
                got = eng.classify(buf, offs, lens)
                assert np.array_equal(got[0], exp), (reads_per_wave, base_ticks, n_equal, "equal slices")
                assert all(np.array_equal(a, b) for a, b in zip(got, base)), (reads_per_wave, base_ticks, n_equal, "equal slices")
            eng.set_phase_equal_slices(0)
            eng.set_phased(0, 1 << 40, 150, 0, 1)
            eng.set_phase_slices(1, 1)
            # sub-batches: every remainder of the batch size modulo the reads per wave
            for n_sub in (2049, 2050, 2051):
                got = eng.classify(buf, offs[:n_sub], lens[:n_sub])
                assert np.array_equal(got[0], exp[:n_sub]), (reads_per_wave, n_sub)
    eng.set_phased()
    eng.set_phase_slices()
    # reads longer than the builds take: the engine plans the builds with the offsets in registers for the batch
    eng.set_reads_per_wave(2)
    assert eng.plan(0, 4096, 500)["kernel"] != "ibf_count_max_phased_multi_kernel"


def test_device_thresholds_match_the_reference_compiled_table():
    """SURVEY a.5 / a.6 on the device: the threshold table K2 reads is checked against tests/golden/thresholds_reference.json -- values the
    REFERENCE'S OWN calculateCI (src/IBF/IBF.hpp:268-338) computed when compiled under its own -Ofast, with the threshold expression of
    src/IBF/IBFClassify.cpp:154-159 (tests/golden/make_thresholds_reference.py) -- not against the oracle.  The table is observed through
    decisions: for a read of length L a prefix is inserted into one bin so that exactly m of its k-mers hit, once with m = t(L) and once
    with m = t(L) - 1; a deplete-only check_unblock says "unblock" for the first and "keep" for the second iff the device's threshold for
    L is the reference's t(L) (Read::classify, IBFClassify.cpp:262-273: a match needs count >= threshold and count > 0).  Both error rates
    of adaptive_sampling.hpp:55 (r and r - 0.02), micro-batch and throughput forms, lengths from below the int16 wrap (negative
    thresholds arrive as 65 5xx: never a match) up to 4 000."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_thresholds_reference as ref
    fx, full = ref.load_fixture()
    k = 13
    rng = np.random.default_rng(2026)
    for rate in (0.1, 0.1 - 0.02):
        d = capi.DeviceIBF.create(0, 64, 3, k, 64 << 20)  # sparse: the planted k-mers are the only hits
        eng = capi.Engine(0, [d], [])
        table = full[ref.key(k, rate)]  # index L - k
        lengths = sorted(set(list(range(20, 140, 3)) + list(range(140, 700, 7)) + list(range(700, 4000, 97)) + [35, 131, 250, 360, 1500]))
        reads, want_m, want_dec, ins_start, ins_end, ins_bin = [], [], [], [], [], []
        at = 0
        for L in lengths:
            t = int(table[L - k])
            n_kmers = L - k + 1
            for m in (t, t - 1):
                if t >= 32768:        # a negative int16 threshold: received as uint16 it is beyond any count -- all k-mers hit, no match
                    m = n_kmers
                if m < 0 or m > n_kmers:
                    continue
                r = H.random_dna(rng, L)
                reads.append(r)
                want_m.append(m)
                want_dec.append(1 if (m >= t and m > 0) else 0)
                if m > 0:
                    ins_start.append(at)
                    ins_end.append(at + m + k - 1)
                    ins_bin.append(len(reads) % 64)
                at += L
        d.insert("".join(reads), np.array(ins_start, dtype=np.uint64), np.array(ins_end, dtype=np.uint64), np.array(ins_bin, dtype=np.uint64))
        buf, offs, lens = H.pack_reads(reads)
        want_m, want_dec = np.array(want_m), np.array(want_dec, dtype=np.uint8)
        assert want_dec.sum() > 100 and (want_dec == 0).sum() > 100
        thr_of = np.array([int(table[int(L) - k]) for L in lens])
        for split in (2048, 0):  # latency form (sub-batches of up to 512 reads: the count kernel decides itself) and throughput form + K2
            eng.set_split_threshold(split)
            for lo in range(0, len(reads), 500 if split else len(reads)):
                hi = min(len(reads), lo + (500 if split else len(reads)))
                mc, _, dec, st = eng.classify(buf, offs[lo:hi], lens[lo:hi], error_rate=rate)
                # (a long random read now and then repeats one of its own planted 13-mers further down: the count the device reports is
                # the count -- verified against the oracle elsewhere -- and the decision must follow from IT and the reference's threshold)
                got_m = mc[:, 0].astype(np.int64)
                assert (got_m == want_m[lo:hi]).mean() > 0.97 and np.all(got_m >= want_m[lo:hi]), "the planted counts are not what the test built"
                expect = ((got_m >= thr_of[lo:hi]) & (got_m > 0)).astype(np.uint8)
                bad = np.nonzero(dec != expect)[0]
                assert len(bad) == 0, [(int(lens[lo + i]), int(mc[i, 0]), int(thr_of[lo + i])) for i in bad[:5]]
                assert (expect != want_dec[lo:hi]).sum() <= 3  # ... and nearly every read sits where it was put: one each side of t(L)
        eng.destroy()


@pytest.mark.parametrize("nd,nt", [(1, 0), (1, 1), (2, 2), (0, 2)])
def test_early_decision_mode_changes_no_output(nd, nt):
    """The opt-in early-decision mode (rb_engine_set_early_decision): RB_MODE_CHECK_UNBLOCK calls of the throughput form that do not ask
    for the raw maxima let a wave of the plain count kernel stop once a bin has reached the larger of the read's two thresholds
    (adaptive_sampling.hpp:47-86 looks at a count only through "count >= threshold(r)" and "count >= threshold(r - 0.02)").  Every output
    the call returns -- decision, status, best_target -- equals the mode being off and the oracle's check_unblock: wide filters (one wave
    per block, and several lane groups per block, whose counters are partial when a wave leaves), several column slices, three error
    rates, reads that match on either strand, reads right at their thresholds, short and empty reads."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(7 + 10 * nd + nt)
    ref = H.random_dna(rng, 60000)
    geos = [(8192, 13), (600, 13), (1024, 15), (8300, 13)]  # W = 128 (16-byte lanes), 10 (16 lanes per block), 16, 130 (two column slices)
    filters, views, keep = [], [], []
    for i in range(nd + nt):
        n_bins, k = geos[i % len(geos)]
        W = (n_bins + 63) // 64
        d = capi.DeviceIBF.create(0, n_bins, 3, k, W * 64 * 24001)  # (2 000-base bins: a fifth of a bin's bits set)
        lo = (i * 9000) % 40000
        d.add_sequence(ref[lo:lo + 16000], 2000)
        o, kp = oracle_view(d)
        filters.append(d); views.append(o); keep.append(kp)
    # > 2 048 reads: the throughput form.  Error rates from clean to hopeless put many reads near their thresholds.
    reads = []
    for e_ in (0.0, 0.05, 0.1, 0.14, 0.18, 0.25):
        reads += make_reads(rng, ref, 450, lo=10, hi=700, err=e_)
    reads += ["", "ACGT", "A" * 13, ref[500:860], ref[20000:20360]]
    buf, offs, lens = H.pack_reads(reads)
    n = len(lens)
    assert n > 2048
    up = lambda a, dt: torch.from_numpy(a.view(dt)).to(dev)
    t_buf, t_offs, t_lens = up(np.ascontiguousarray(buf), np.uint8), up(np.ascontiguousarray(offs, dtype=np.uint64), np.int64), up(np.ascontiguousarray(lens, dtype=np.uint32), np.int32)
    eng = capi.Engine(0, filters[:nd], filters[nd:])
    for r in (0.1, 0.05, 0.15):
        exp_dec, exp_st = po.batch_check_unblock(views[:nd], views[nd:], buf, offs, lens, r=r, n_threads=8)
        got = {}
        for early in (0, 1):
            eng.set_early_decision(early)
            for with_best in (True, False):
                t_dec = torch.full((n,), 9, dtype=torch.uint8, device=dev)
                t_st = torch.full((n,), 9, dtype=torch.uint8, device=dev)
                t_best = torch.full((n,), -9, dtype=torch.int32, device=dev)
                torch.cuda.synchronize()
                eng.classify_device(t_buf.data_ptr(), t_offs.data_ptr(), t_lens.data_ptr(), n, int(lens.max()), error_rate=r,
                                    d_best=t_best.data_ptr() if with_best else None, d_decision=t_dec.data_ptr(), d_status=t_st.data_ptr())
                torch.cuda.synchronize()
                got[(early, with_best)] = (t_dec.cpu().numpy(), t_st.cpu().numpy(), t_best.cpu().numpy())
                assert np.array_equal(got[(early, with_best)][0], exp_dec), (r, early, with_best)
                assert np.array_equal(got[(early, with_best)][1], exp_st), (r, early, with_best)
        assert np.array_equal(got[(1, True)][2], got[(0, True)][2])  # best_target as without the mode
        assert len(set(exp_dec.tolist())) >= 2
    # the mode never touches a call that asks for the raw maxima
    eng.set_early_decision(1)
    mc = eng.classify(buf, offs, lens)[0]
    assert np.array_equal(mc, np.stack([po.batch_raw_max(v, buf, offs, lens, 8) for v in views], axis=1))
    eng.set_early_decision(0)

"""Live micro-batch shim (rb_live_*) against a sequential restatement of classify_live_reads
(src/main/adaptive_sampling.hpp:227-350) driven by the oracle's check_unblock."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import pyoracle as po
from readbouncer_amd import capi
from tests import helpers as H


def reference_live(odep, otgt, events, r=0.1, cutoff=1500):
    once, out = {}, []
    for rid, seq in events:
        st, dec = po.check_unblock(odep, otgt, po.encode(seq), r=r)
        if st != po.OK:
            out.append((0, st))
            continue
        if dec in (1, 2):
            once.pop(rid, None)
            out.append((dec, 0))
            continue
        if rid in once:
            s2 = once[rid] + seq
            st2, d2 = po.check_unblock(odep, otgt, po.encode(s2), r=r)
            if st2 != po.OK:
                out.append((0, st2))
            elif d2 in (1, 2):
                del once[rid]
                out.append((d2, 0))
            elif len(s2) > cutoff:
                del once[rid]
                out.append((2, 0))
            else:
                once[rid] = s2
                out.append((0, 0))
        else:
            once[rid] = seq
            out.append((0, 0))
    return out, once


@pytest.mark.parametrize("nd,nt", [(1, 1), (1, 0), (0, 1)])
def test_live_shim_matches_sequential_reference(nd, nt):
    rng = np.random.default_rng(42 + nd * 2 + nt)
    host = H.random_dna(rng, 30000)
    bug = H.random_dna(rng, 30000)
    filters, views, keep = [], [], []
    for src, n_bins in ((host, 300), (bug, 64)):
        d = capi.DeviceIBF.create(0, n_bins, 3, 13, ((n_bins + 63) // 64) * 64 * 120011)  # roomy: few false positives
        d.add_sequence(src, 1000)
        h = d.download()
        keep.append(h)
        views.append(po.OracleIBF.wrap(h.info["n_bins"], 3, 13, h.info["n_bits"], h.words()))
        filters.append(d)
    dep, tgt = (filters[:1] if nd else []), (filters[1:] if nt else [])
    odep, otgt = (views[:1] if nd else []), (views[1:] if nt else [])
    # molecules: host-derived, target-derived, chimeric (host+target) and random, delivered in chunks
    events = []
    for m in range(150):
        kind = m % 5
        L = int(rng.integers(300, 2600))
        if kind == 0:
            s = int(rng.integers(0, 30000 - L)); mol = H.mutate(rng, host[s:s + L], 0.12)
        elif kind == 1:
            s = int(rng.integers(0, 30000 - L)); mol = H.mutate(rng, bug[s:s + L], 0.12)
        elif kind == 2:
            a = int(rng.integers(0, 29000)); b = int(rng.integers(0, 29000))
            mol = H.mutate(rng, (host[a:a + L // 2] + bug[b:b + L // 2]), 0.05)
        elif kind == 3:
            mol = H.random_dna(rng, L)
        else:
            s = int(rng.integers(0, 30000 - L)); mol = H.mutate(rng, host[s:s + L], 0.25)  # noisy: often undecided
        pos = 0
        while pos < len(mol):
            step = int(rng.choice([8, 60, 120, 250, 360, 400]))
            events.append(("read%03d" % m, mol[pos:pos + step]))
            pos += step
    order = rng.permutation(len(events))
    # keep the chunks of one read in order while interleaving reads
    by_read = {}
    for rid, ch in events:
        by_read.setdefault(rid, []).append(ch)
    stream = []
    for idx in order:
        rid = events[idx][0]
        if by_read[rid]:
            stream.append((rid, by_read[rid].pop(0)))
    exp, exp_once = reference_live(odep, otgt, stream)
    eng = capi.Engine(0, dep, tgt)
    live = capi.Live(eng)
    got = []
    pos = 0
    while pos < len(stream):
        m = int(rng.integers(1, 90))
        batch = stream[pos:pos + m]  # batches may hold several chunks of the same read
        a, s, _ = live.process([b[0] for b in batch], [b[1] for b in batch])
        got += list(zip(a.tolist(), s.tolist()))
        pos += m
    assert got == exp
    assert live.pending() == len(exp_once)
    acts = [g[0] for g in got]
    assert acts.count(1) > 5 or nd == 0
    assert acts.count(2) > 5
    assert any(g[1] == capi.RB_ERR_SHORT_READ for g in got) or (nd and nt)  # pair overload never throws
    live.forget(next(iter(exp_once)) if exp_once else "none")
    assert live.pending() == max(0, len(exp_once) - 1)


def test_replay_arrivals_dispatcher():
    """rb_replay_arrivals: the work-conserving micro-batch dispatcher over an arrival list (BASELINE configs[4]) -- every
    chunk is classified exactly once, in arrival order, with the decision rb_classify_batch gives for it; latencies are
    decision - arrival; the calls' sizes add up."""
    rng = np.random.default_rng(5)
    ref = H.random_dna(rng, 20000)
    d = capi.DeviceIBF.create(0, 300, 3, 13, 320 * 60013)
    d.add_sequence(ref, 1000)
    n, L = 3000, 360
    reads = [H.mutate(rng, ref[s:s + L], 0.08) if i % 2 else H.random_dna(rng, L)
             for i, s in enumerate(rng.integers(0, 20000 - L, size=n))]
    buf, offs, lens = H.pack_reads(reads)
    eng = capi.Engine(0, [d], [])
    _, _, exp_dec, _ = eng.classify(buf, offs, lens)
    arrival = np.cumsum(rng.exponential(1.0 / 60000.0, size=n))  # 60 k chunks/s for 50 ms
    dec, lat, call_reads, call_service, elapsed = eng.replay_arrivals(buf, L, arrival, max_batch=64)
    assert np.array_equal(dec, exp_dec) and len(set(dec.tolist())) == 2
    assert int(call_reads.sum()) == n and call_reads.max() <= 64 and len(call_reads) == len(call_service)
    assert (lat > 0).all() and elapsed >= arrival[-1] and (call_service > 0).all()
    assert np.median(lat) < 0.005  # far from the 1 ms SLO even on a busy box


def test_live_replay_equals_the_sequential_reference():
    """rb_live_replay_arrivals: the work-conserving dispatcher in front of rb_live_process.  Whatever micro-batches the
    arrival process cuts, the action per chunk equals the sequential restatement of classify_live_reads on the same stream
    (undecided chunks concatenated with what once_seen holds, 1500 bp cut-off)."""
    rng = np.random.default_rng(77)
    host = H.random_dna(rng, 30000)
    bug = H.random_dna(rng, 30000)
    filters, views, keep = [], [], []
    for src, n_bins in ((host, 300), (bug, 64)):
        d = capi.DeviceIBF.create(0, n_bins, 3, 13, ((n_bins + 63) // 64) * 64 * 120011)
        d.add_sequence(src, 1000)
        h = d.download()
        keep.append(h)
        views.append(po.OracleIBF.wrap(h.info["n_bins"], 3, 13, h.info["n_bits"], h.words()))
        filters.append(d)
    L, n_reads, n_chunks = 360, 400, 5
    mols = []
    for m in range(n_reads):
        kind = m % 4
        s = int(rng.integers(0, 30000 - L * n_chunks))
        mol = (H.mutate(rng, host[s:s + L * n_chunks], 0.1) if kind == 0 else H.mutate(rng, bug[s:s + L * n_chunks], 0.1)
               if kind == 1 else H.random_dna(rng, L * n_chunks) if kind == 2 else H.mutate(rng, host[s:s + L * n_chunks], 0.26))
        mols.append(mol)
    first = np.sort(rng.uniform(0.0, 0.02, size=n_reads))
    arr = (first[:, None] + 0.004 * np.arange(n_chunks)[None, :]).reshape(-1)
    ids = np.repeat(np.arange(n_reads, dtype=np.uint32), n_chunks)
    chunk_no = np.tile(np.arange(n_chunks), n_reads)
    order = np.argsort(arr, kind="stable")
    arr, ids, chunk_no = arr[order], ids[order], chunk_no[order]
    chunks = [mols[i][c * L:(c + 1) * L] for i, c in zip(ids, chunk_no)]
    buf = np.frombuffer("".join(chunks).encode(), dtype=np.uint8).copy()
    stream = [(ids[j].tobytes(), chunks[j]) for j in range(len(chunks))]
    exp, exp_once = reference_live(views[:1], views[1:], stream)
    eng = capi.Engine(0, filters[:1], filters[1:])
    live = capi.Live(eng)
    act, lat, clen, call_reads, service, elapsed = live.replay_arrivals(ids, buf, L, arr, max_batch=97)
    assert act.tolist() == [e[0] for e in exp]
    assert live.pending() == len(exp_once)
    assert int(call_reads.sum()) == len(chunks) and call_reads.max() <= 97 and (lat > 0).all() and elapsed >= arr[-1]
    assert (clen > L).sum() > 100 and clen.max() > 1080  # concatenations up to the cut-off went through the GPU
    assert set(act.tolist()) == {0, 1, 2}

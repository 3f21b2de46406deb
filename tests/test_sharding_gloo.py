"""N>1 path on CPU: world_size-2 gloo processes exercise the read-sharding and bin-sharding plumbing
(readbouncer_amd/sharding.py).  The per-rank "classification" is done by the oracle here (test stand-in for the
GPU engine, which cannot run without a device); what is under test is the partitioning, the host-side gather and
the all-reduce(max) of partial maxima."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import pyoracle as po
    from readbouncer_amd import sharding
    from tests import helpers as H

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(77)  # same data on every rank
        ref = H.random_dna(rng, 20000)
        f = po.OracleIBF(300, 3, 13, 320 * 4001)
        f.add_sequence(po.encode(ref), 1000)
        reads = [H.mutate(rng, ref[s:s + 300], 0.05) if i % 2 else H.random_dna(rng, 300)
                 for i, s in enumerate(rng.integers(0, 19000, size=101))]
        buf, offs, lens = H.pack_reads(reads)
        n = len(reads)
        # --- read-sharded: each rank classifies its contiguous slice, rank 0 gathers
        b, e = sharding.read_slice(n, rank, world)
        dec, _ = po.batch_check_unblock([f], [], buf, offs[b:e], lens[b:e])
        full = sharding.gather_decisions(dec, n, rank, world, dist)
        # --- bin-sharded: each rank counts only its word columns; all-reduce(max) of the partial maxima
        cb, ce = sharding.column_slice(f.bin_width, rank, world)
        part = np.zeros(n, dtype=np.uint16)
        for i, r in enumerate(reads):
            o = po.encode(r)
            c = np.maximum(f.count(o), f.count(po.revcomp(o)))
            sl = c[cb * 64: min(ce * 64, f.n_bins)]
            part[i] = sl.max() if len(sl) else 0
        red = sharding.allreduce_max_partial(part, dist)
        if rank == 0:
            exp_dec, _ = po.batch_check_unblock([f], [], buf, offs, lens)
            exp_max = po.batch_raw_max(f, buf, offs, lens)
            q.put((bool(np.array_equal(full, exp_dec)), bool(np.array_equal(red, exp_max)), int(exp_dec.sum())))
    finally:
        dist.destroy_process_group()


def test_world2_read_and_bin_sharding():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    ok_dec, ok_max, n_unblock = q.get(timeout=5)
    assert ok_dec and ok_max and n_unblock > 10


def test_slices_partition_everything():
    from readbouncer_amd import sharding
    for n in (0, 1, 7, 100, 101, 1_000_003):
        for world in (1, 2, 3, 4, 8):
            cover = []
            for r in range(world):
                b, e = sharding.read_slice(n, r, world)
                assert 0 <= b <= e <= n
                cover.append((b, e))
            assert cover[0][0] == 0 and cover[-1][1] == n
            assert all(cover[i][1] == cover[i + 1][0] for i in range(world - 1))
    for W in (1, 2, 3, 16, 17, 128, 130):
        for world in (1, 2, 3, 8):
            spans = [sharding.column_slice(W, r, world) for r in range(world)]
            assert spans[0][0] == 0 and max(s[1] for s in spans) == W
            assert all(spans[i][1] == spans[i + 1][0] or spans[i + 1][0] == W for i in range(world - 1))
            assert all(s[0] % 2 == 0 or s[0] == s[1] or world == 1 for s in spans)  # empty tail slices may start at W

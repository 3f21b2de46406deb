"""Read-sharding across the GPUs of one node: reads are independent units, every rank
holds a full replica of each IBF, batch slices are contiguous blocks of ceil(n/world) reads, and no
data-path collective exists.  The optional bin-sharded layout (each rank holds a word-column range of
every block) combines per-read partial maxima with one all-reduce(max)."""
import numpy as np


def read_slice(n_reads, rank, world):
    """[begin, end) of the contiguous block of reads rank owns"""
    per = (n_reads + world - 1) // world
    b = min(n_reads, per * rank)
    return b, min(n_reads, b + per)


def column_slice(bin_width, rank, world):
    """word-column range of every block a rank owns in the bin-sharded layout (mirrors rb_engine_set_column_shard)"""
    per = (bin_width + world - 1) // world
    if world > 1 and per & 1:
        per += 1
    b = min(bin_width, per * rank)
    return b, min(bin_width, b + per)


def gather_decisions(local, n_reads, rank, world, dist=None):
    """concatenate per-rank uint8 outputs of read-sharded classification on rank 0 (host-side gather)"""
    if world == 1 or dist is None:
        return local
    import torch
    per = (n_reads + world - 1) // world
    pad = np.zeros(per, dtype=np.uint8)
    pad[: len(local)] = local
    t = torch.from_numpy(pad)
    out = [torch.zeros(per, dtype=torch.uint8) for _ in range(world)] if rank == 0 else None
    dist.gather(t, out, dst=0)
    if rank != 0:
        return None
    full = np.concatenate([o.numpy() for o in out])[:n_reads]
    return full


def allreduce_max_partial(partial, dist=None):
    """bin-sharded layout: element-wise max of the per-rank partial maxima (uint16 carried as int32)"""
    if dist is None:
        return partial
    import torch
    t = torch.from_numpy(partial.astype(np.int32))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t.numpy().astype(np.uint16)

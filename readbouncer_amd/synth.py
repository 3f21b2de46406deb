"""Synthetic workloads of BASELINE.json's configs (host-side numpy plumbing, no classification logic).

A workload = an IBF geometry + a seeded way to fill it in HBM (random design-load bits plus planted
reference segments inserted by the GPU insert kernel) + seeded read batches (50 % sampled from the
planted segments with i.i.d. substitution errors, 50 % uniform random), as SURVEY.md section 8d lays out.
"""
import numpy as np

from . import capi

WORKLOADS = {
    # config 2: 1M x 360 bp vs a chr1-like IBF: B=1024, k=13, h=3, F=243 000 -> ~0.41 GB
    "c2": dict(name="config2: 1M synthetic 360bp prefixes vs chr1-like IBF (k=13, h=3, 1024 bins, F=243000, ~0.41 GB)",
               n_bins=1024, k=13, h=3, fragment=243000, n_bits=None, reads=1_000_000, read_len=360),
    # config 3: 10M x 360 bp vs a GRCh38-scale IBF: B=8192, 2^36 bits = 8 GiB
    "c3": dict(name="config3: 10M synthetic 360bp prefixes vs GRCh38-scale IBF (k=13, h=3, 8192 bins, 8 GiB)",
               n_bins=8192, k=13, h=3, fragment=None, n_bits=1 << 36, reads=10_000_000, read_len=360),
    # config 3 with the reference's own sizing rule (non-power-of-two block count)
    "c3np2": dict(name="config3 (BinSizeBits*8256 sizing): 360bp prefixes vs 8192-bin IBF, generic modulus",
                  n_bins=8192, k=13, h=3, fragment=370000, n_bits=None, reads=10_000_000, read_len=360),
    # the same filter size class with 128-byte aligned blocks (what a padded HBM layout of "zymo" would look like)
    "zymo16": dict(name="1024-bin IBF at F=100000 (aligned 128-byte blocks, 168 MB)",
                   n_bins=1024, k=13, h=3, fragment=100000, n_bits=None, reads=1_000_000, read_len=360),
    # GRCh38 exactly as ReadBouncer itself would build it: fragment_size 100000 (its default) -> ~31 000 bins,
    # W = 485 words (odd: 8-byte lanes, 8 column slices), 4.8 GB, odd block count
    "grch38_f100k": dict(name="GRCh38 at ReadBouncer's default fragment_size=100000: 31000 bins (3.9 KB blocks), 4.8 GB",
                         n_bins=31000, k=13, h=3, fragment=100000, n_bits=None, reads=2_000_000, read_len=360),
    # config 1 geometry (64 bins, F=100000) for completeness
    "c1": dict(name="config1 geometry: 360bp prefixes vs 64-bin IBF (k=13, F=100000)",
               n_bins=64, k=13, h=3, fragment=100000, n_bits=None, reads=1_000_000, read_len=360),
    # mock-community target of config 4: ~600 bins at F=100000
    "zymo": dict(name="Zymo-mock-like target IBF (600 bins, F=100000)",
                 n_bins=600, k=13, h=3, fragment=100000, n_bits=None, reads=1_000_000, read_len=360),
    # the shape of the reference's only published benchmark (README.md:254-262): 3 target + 1 deplete filters built from
    # single mock-community genomes at k=13, F=100000 -- all of them narrow (1-2 word columns, 5-20 MB)
    "mock_deplete": dict(name="12.1 Mbp genome (122 bins, F=100000)", n_bins=122, k=13, h=3, fragment=100000, n_bits=None,
                         reads=1_000_000, read_len=250),
    "mock_t1": dict(name="4.2 Mbp genome (43 bins)", n_bins=43, k=13, h=3, fragment=100000, n_bits=None, reads=1_000_000, read_len=250),
    "mock_t2": dict(name="2.8 Mbp genome (29 bins)", n_bins=29, k=13, h=3, fragment=100000, n_bits=None, reads=1_000_000, read_len=250),
    "mock_t3": dict(name="4.8 Mbp genome (49 bins)", n_bins=49, k=13, h=3, fragment=100000, n_bits=None, reads=1_000_000, read_len=250),
    # a one-word filter at a large fragment_size (64 bins of 615 kbp: a 39 Mbp genome), 64 MiB: the upper half of the range the
    # clock-phased gathers serve since round 3 (16 slices of 4 MiB)
    "w1_64mib": dict(name="64-bin IBF of 64 MiB (one-word blocks, 8 388 605 of them)", n_bins=64, k=13, h=3, fragment=None,
                     n_bits=64 * 8388605, reads=1_000_000, read_len=250),
}

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def filter_bits(w):
    if w["n_bits"] is not None:
        return w["n_bits"]
    return capi.calculate_filter_size_bits(w["fragment"], w["k"], w["h"], 0.01, w["n_bins"])


def planted_reference(seed, n_segments=2048, seg_len=2000):
    """seeded uniform ACGT segments, concatenated: (uint8 ascii array, starts, ends)"""
    rng = np.random.default_rng(seed)
    ref = _ACGT[rng.integers(0, 4, size=n_segments * seg_len, dtype=np.uint8)]
    starts = np.arange(n_segments, dtype=np.uint64) * seg_len
    return ref, starts, starts + np.uint64(seg_len)


def build_device_filter(device, w, fill_seed, plant_seed, n_segments=2048, seg_len=2000):
    """random design-load fill + planted segments (segment i -> bin i mod n_bins); returns (DeviceIBF, ref)"""
    d = capi.DeviceIBF.create(device, w["n_bins"], w["h"], w["k"], filter_bits(w))
    d.fill_synth(fill_seed)
    ref, starts, ends = planted_reference(plant_seed, n_segments, seg_len)
    bins = (np.arange(n_segments, dtype=np.uint64) * np.uint64(7919)) % np.uint64(w["n_bins"])
    d.insert(ref, starts, ends, bins)
    return d, ref


# Threshold-adjacent stratum: every tenth positive read gets NEAR_ERROR instead of `error_rate`.  A substitution draws from ACGT
# (a quarter of them restore the base), so the effective rate is 0.75 x nominal: 0.19-0.23 nominal = 14-17 % effective, where a
# 360 bp read keeps 348 x (1 - e)^13 = 30-48 of its 13-mers -- around the threshold of 38 (r = 0.1) and down to that of the
# r - 0.02 re-test's neighbourhood.  Without it nearly every read of a batch sits far from its threshold (positives ~125 shared
# k-mers, negatives ~10) and a count that is off by a few could not flip a decision.
NEAR_FRACTION = 0.1
NEAR_ERROR = (0.19, 0.23)


def make_reads(seed, n_reads, read_len, ref, positive_fraction=0.5, error_rate=0.10, seg_len=2000, near_fraction=NEAR_FRACTION):
    """fixed-length reads: positives sampled inside planted segments with substitutions (a `near_fraction` of them at the
    threshold-adjacent error rates NEAR_ERROR), negatives uniform.
    returns (uint8 [n_reads*read_len], uint64 offsets, uint32 lens)"""
    rng = np.random.default_rng(seed)
    n_pos = int(n_reads * positive_fraction)
    out = _ACGT[rng.integers(0, 4, size=(n_reads, read_len), dtype=np.uint8)]
    if n_pos and ref is not None and len(ref) >= seg_len >= read_len:
        n_seg = len(ref) // seg_len
        seg = rng.integers(0, n_seg, size=n_pos)
        off = rng.integers(0, seg_len - read_len + 1, size=n_pos)
        start = seg * seg_len + off
        idx = start[:, None] + np.arange(read_len)[None, :]
        pos = ref[idx]
        rate = np.full(n_pos, error_rate)
        if near_fraction > 0:
            near = rng.random(n_pos) < near_fraction
            rate[near] = rng.uniform(NEAR_ERROR[0], NEAR_ERROR[1], size=int(near.sum()))
        err = rng.random((n_pos, read_len)) < rate[:, None]
        pos[err] = _ACGT[rng.integers(0, 4, size=int(err.sum()), dtype=np.uint8)]
        # every second positive is given as its reverse complement
        comp = np.zeros(256, dtype=np.uint8)
        comp[_ACGT] = np.frombuffer(b"TGCA", dtype=np.uint8)
        rc = comp[pos[1::2, ::-1]]
        pos[1::2] = rc
        perm = rng.permutation(n_reads)[:n_pos]
        out[perm] = pos
    lens = np.full(n_reads, read_len, dtype=np.uint32)
    offs = np.arange(n_reads, dtype=np.uint64) * np.uint64(read_len)
    return np.ascontiguousarray(out.reshape(-1)), offs, lens


def make_reads_device(seed, n_reads, read_len, ref, device, positive_fraction=0.5, error_rate=0.10, seg_len=2000,
                      near_fraction=NEAR_FRACTION):
    """same construction as make_reads (threshold-adjacent stratum included), generated with torch on `device` (plumbing: 10M
    reads are 3.6 GB);
    returns torch tensors (uint8 [n_reads*read_len], int64 offsets, int32 lens) resident on the device"""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(int(seed))
    acgt = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    out = torch.empty((n_reads, read_len), dtype=torch.uint8, device=device)
    for b in range(0, n_reads, 1 << 20):  # in slabs: the index tensor of a 10 M-read batch would be 29 GB
        m = min(1 << 20, n_reads - b)
        out[b:b + m] = acgt[torch.randint(0, 4, (m, read_len), generator=g, device=device)]
    n_pos = int(n_reads * positive_fraction)
    if n_pos and ref is not None and len(ref) >= seg_len >= read_len:
        t_ref = torch.from_numpy(np.ascontiguousarray(ref)).to(device)
        n_seg = len(ref) // seg_len
        ar = torch.arange(read_len, device=device)
        perm = torch.randperm(n_reads, generator=g, device=device)[:n_pos]
        chunk = 1 << 20
        comp = torch.zeros(256, dtype=torch.uint8, device=device)
        comp[acgt.long()] = torch.tensor(list(b"TGCA"), dtype=torch.uint8, device=device)
        for b in range(0, n_pos, chunk):
            m = min(chunk, n_pos - b)
            seg = torch.randint(0, n_seg, (m,), generator=g, device=device)
            off = torch.randint(0, seg_len - read_len + 1, (m,), generator=g, device=device)
            idx = (seg * seg_len + off)[:, None] + ar[None, :]
            pos = t_ref[idx]
            rate = torch.full((m,), float(error_rate), device=device)
            if near_fraction > 0:
                near = torch.rand((m,), generator=g, device=device) < near_fraction
                span = torch.rand((m,), generator=g, device=device) * (NEAR_ERROR[1] - NEAR_ERROR[0]) + NEAR_ERROR[0]
                rate = torch.where(near, span, rate)
            err = torch.rand((m, read_len), generator=g, device=device) < rate[:, None]
            sub = acgt[torch.randint(0, 4, (m, read_len), generator=g, device=device)]
            pos = torch.where(err, sub, pos)
            rc = comp[pos.flip(1).long()]
            odd = (torch.arange(m, device=device) % 2 == 1)[:, None]
            pos = torch.where(odd, rc, pos)
            out[perm[b:b + m]] = pos
    lens = torch.full((n_reads,), read_len, dtype=torch.int32, device=device)
    offs = torch.arange(n_reads, dtype=torch.int64, device=device) * read_len
    out = out.reshape(-1).contiguous()
    if torch.device(device).type == "cuda":
        # torch filled these on ITS current stream; the engine launches on streams of its own (non-blocking: no implicit order with
        # the default stream).  A caller that goes straight to rb_classify_batch_device would race the fill -- lengths still holding a
        # freed tensor's bytes made K1 read far outside the batch (round 4, a profiling script) -- so the batch is complete on return.
        torch.cuda.synchronize(device)
    return out, offs, lens


def algorithmic_bytes_per_read(read_len, filters):
    """sum_filters 2*(L-k+1)*h*8*ceil(B/64) + L + 2*n_filters; filters = [(n_bins, k, h), ...]"""
    total = read_len + 2 * len(filters)
    for n_bins, k, h in filters:
        n = max(0, read_len - k + 1)
        total += 2 * n * h * 8 * ((n_bins + 63) // 64)
    return total

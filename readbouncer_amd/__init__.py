"""readbouncer_amd -- MI355X-native Interleaved Bloom Filter read classification.

The product is libreadbouncer_amd.so (C ABI in include/readbouncer_amd.h, HIP kernels in
readbouncer_amd/csrc).  This package only holds the ctypes plumbing used by tests and bench.py,
plus host-side helpers (read sharding, synthetic workloads).
"""
from . import capi  # noqa: F401

__all__ = ["capi"]

"""SURVEY 8 a.5 / a.6 pinned against the reference's own code: `calculateCI` / `NormalCDFInverse` (src/IBF/IBF.hpp:268-338) compile
stand-alone, so tests/golden/make_thresholds_reference.py builds them here under the reference's flag (-Ofast, src/CMakeLists.txt:30),
applies the threshold expression of src/IBF/IBFClassify.cpp:154-159 and keeps the result as tests/golden/thresholds_reference.json
(data generated FROM the reference's code; no reference source in the repository).

* every box: the oracle (oracle/ibf_oracle.c) and the product's host source of the device threshold table (rb_threshold =
  rb::threshold_u16, csrc/rb_host.cpp, the function rb_engine.hip fills the table from) against the fixture -- 54 (k, rate) tables x
  L = k .. 70 000, 3.8 M points, thresholds and interval bounds;
* where /root/reference and g++ exist (the build container): the fixture is regenerated from the reference tree and must be identical,
  under -Ofast and under -O2.
"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np
import pytest

from oracle import pyoracle as po
from readbouncer_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_thresholds_reference as ref  # noqa: E402


def product_table(k, r, l_max, conf):
    f = capi.lib().rb_threshold
    return np.fromiter((f(L, k, r, conf) for L in range(k, l_max + 1)), dtype=np.uint16, count=l_max - k + 1)


def product_bounds(k, r, l_max, conf):
    f = capi.lib().rb_calculate_ci
    lo, hi = C.c_uint16(), C.c_uint16()
    out = np.empty((l_max - k + 1, 2), dtype=np.uint16)
    for i, L in enumerate(range(k, l_max + 1)):
        f(r, k, L, conf, C.byref(lo), C.byref(hi))
        out[i, 0], out[i, 1] = lo.value, hi.value
    return out


def oracle_table(k, r, l_max, conf):
    f = po.lib().orc_threshold
    return np.fromiter((f(L, k, r, conf) for L in range(k, l_max + 1)), dtype=np.uint16, count=l_max - k + 1)


def oracle_bounds(k, r, l_max, conf):
    f = po.lib().orc_calculate_ci
    lo, hi = C.c_uint16(), C.c_uint16()
    out = np.empty((l_max - k + 1, 2), dtype=np.uint16)
    for i, L in enumerate(range(k, l_max + 1)):
        f(r, k, L, conf, C.byref(lo), C.byref(hi))
        out[i, 0], out[i, 1] = lo.value, hi.value
    return out


def test_oracle_and_product_match_the_reference_compiled_tables():
    fx, full = ref.load_fixture()
    conf, l_max = fx["significance"], fx["l_max"]
    assert l_max >= 70000 and len(fx["tables"]) >= 50
    points = 0
    for name, t in fx["tables"].items():
        k, r = t["k"], float.fromhex(t["rate"])
        thr_o, thr_p = oracle_table(k, r, l_max, conf), product_table(k, r, l_max, conf)
        assert ref.digest(thr_o) == t["thresholds_sha256"], ("oracle", name)
        assert ref.digest(thr_p) == t["thresholds_sha256"], ("product", name)
        if name in full:  # the full table: says WHERE, should a digest ever differ
            assert np.array_equal(thr_o, full[name]) and np.array_equal(thr_p, full[name])
            assert ref.digest(oracle_bounds(k, r, l_max, conf)) == t["ci_sha256"], ("oracle bounds", name)
            assert ref.digest(product_bounds(k, r, l_max, conf)) == t["ci_sha256"], ("product bounds", name)
        points += len(thr_o)
    assert points > 3_500_000
    # the reference's own spot values (src/test/libIBFTests/read.hpp:154-164) sit in the reference-compiled table too
    t13 = full[ref.key(13, 0.1)]
    assert t13[360 - 13] == 38 and int(np.int16(t13[35 - 13])) == -7


@pytest.mark.skipif(not ref.available(), reason="needs /root/reference (IBF.hpp) and g++: the build container")
@pytest.mark.parametrize("flags", [("-Ofast",), ("-O2",)])
def test_fixture_is_what_the_reference_code_computes_here(flags):
    """regenerate from the reference tree: the committed fixture must be exactly what IBF.hpp:268-338 computes, under the reference's
    -Ofast and under plain -O2 (no fast-math dependence)"""
    with tempfile.TemporaryDirectory() as tmp:
        tables = ref.run(ref.build(tmp, flags), ref.all_pairs())
    fresh = ref.make_fixture(tables)
    fx, _ = ref.load_fixture()
    assert fresh["tables"] == fx["tables"] and fresh["full"] == fx["full"]

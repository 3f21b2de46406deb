"""CPU-side checks of the product library: it loads, exports every symbol include/readbouncer_amd.h
declares, its host functions agree with the oracle, and compute entry points fail loudly without a GPU."""
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po
from readbouncer_amd import capi
from tests import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HEADERS = ("readbouncer_amd.h", "readbouncer_amd_tuning.h")  # the reference-mapped calls; measurement aids and scheduling knobs


def declared_symbols(headers=HEADERS):
    text = "".join(open(os.path.join(ROOT, "include", h)).read() for h in headers)
    return sorted(set(re.findall(r"RB_API[^;(]*?\b(rb_[a-z0-9_]+)\s*\(", text)))


def test_boundary_header_holds_no_lab_equipment():
    """include/readbouncer_amd.h is the drop-in boundary: calls a ReadBouncer integration binds.  Knobs that
    pick kernel forms, planners, probes, replays and synthetic fillers live in readbouncer_amd_tuning.h -- same library -- and no
    symbol is declared twice."""
    main, tuning = declared_symbols(HEADERS[:1]), declared_symbols(HEADERS[1:])
    assert not set(main) & set(tuning)
    lab = re.compile(r"probe|replay|calibrate|fill_synth|set_phased|phase_slices|split_parts|set_timing|kernel_time|get_stats|_plan$|serialize")
    assert [s for s in main if lab.search(s)] == []
    assert len(main) >= 50 and len(tuning) >= 20


def test_integration_recipe_names_every_call():
    """INTEGRATION.md section 1 maps every call of the boundary header to the reference interface it replaces, section 1b lists the
    tuning header's: no export without a line there"""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    one, one_b = text.split("### 1b.")[0], text.split("### 1b.")[1].split("Error conventions")[0]
    assert [s for s in declared_symbols(HEADERS[:1]) if s not in one] == []
    assert [s for s in declared_symbols(HEADERS[1:]) if s not in one_b] == []


def test_library_exports_every_declared_symbol():
    decl = declared_symbols()
    assert len(decl) >= 30
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH], text=True)
    exported = set(re.findall(r" T (rb_[a-z0-9_]+)", out))
    assert set(decl) <= exported, sorted(set(decl) - exported)
    assert set(capi.SIGNATURES) == set(decl)
    L = capi.lib()
    for name in decl:
        assert getattr(L, name) is not None
    # nothing but the C ABI leaks out of the shared object
    leaked = [s for s in re.findall(r" T (\S+)", out) if not s.startswith("rb_") and not s.startswith("_")]
    assert leaked == []


def test_product_does_not_link_or_import_the_oracle():
    out = subprocess.check_output(["ldd", capi.LIB_PATH], text=True)
    assert "oracle" not in out
    for d, _, files in os.walk(os.path.join(ROOT, "readbouncer_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                src = open(os.path.join(d, f)).read()
                assert "pyoracle" not in src and "ibf_oracle" not in src, f


def test_no_gpu_means_loud_failure():
    if capi.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(capi.RBError) as ei:
        capi.DeviceIBF.create(0, 64, 3, 13, 64 * 1024)
    assert ei.value.status == capi.RB_ERR_NO_DEVICE
    # every entry point that needs the GPU says so -- there is no CPU fallback anywhere in the product
    import ctypes as C
    p = C.c_void_p()
    assert capi.lib().rb_host_alloc(1024, C.byref(p)) == capi.RB_ERR_NO_DEVICE and not p.value
    capi.lib().rb_host_free(None)
    host = capi.HostIBF.create(64, 3, 13, 64 * 1024)
    with pytest.raises(capi.RBError) as ei:
        capi.DeviceIBF.upload(0, host)
    assert ei.value.status == capi.RB_ERR_NO_DEVICE
    assert capi.device_count() <= 0


def test_threshold_table_matches_oracle_and_golden(golden_dir):
    import json
    gold = json.load(open(os.path.join(golden_dir, "thresholds.json")))
    for key, vals in gold["tables"].items():
        k, r = key.split("/")
        k, r = int(k), float(r)
        for L, exp in zip(gold["lengths"], vals):
            assert capi.threshold(L, k, r, 0.95) == exp == po.threshold(L, k, r, 0.95), (L, k, r)
    # dense sweep product vs oracle, including r - 0.02 exactly as check_unblock forms it
    for k in (11, 13, 15, 21, 31):
        for r in (0.1, 0.1 - 0.02, 0.05, 0.15 - 0.02, 0.2):
            for L in list(range(0, 700)) + [1000, 1500, 2000, 4000, 65535, 65536, 65549, 70000]:
                assert capi.threshold(L, k, r, 0.95) == po.threshold(L, k, r, 0.95), (L, k, r)
    assert capi.calculate_ci(0.1, 13, 35, 0.95) == (0, 5, 30)
    st, _, _ = capi.calculate_ci(0.1, 13, 35, 1.0)  # NormalCDFInverse(1.0) throws in the reference
    assert st == capi.RB_ERR_INVALID_ARG


def test_build_helpers_match_oracle():
    assert capi.calculate_filter_size_bits(100000, 13, 3, 0.01, 2) == 79121216  # createfilter.hpp:140-148
    rng = np.random.default_rng(0)
    for _ in range(50):
        F = int(rng.integers(50, 200000)); k = int(rng.integers(5, 32)); nb = int(rng.integers(1, 9000))
        assert capi.calculate_filter_size_bits(F, k, 3, 0.01, nb) == po.calculate_filter_size_bits(F, k, 3, 0.01, nb)
    for s in ("AAAAAAAACCCCCCCCCGAGAGAGGAGAGAGGAGAGAGAGAGCCCCAAAAGAGAGGAGATTTTANNNNNNNNTATATTATA", "NNNN", "", "N",
              "ACGT", "ACGTN", "NACGTNNACGNNN", "NNACGTACGT", "A", "AN", "NA"):
        assert capi.cut_out_nnns(s) == po.cut_out_nnns(s), s
    for _ in range(200):
        s = H.random_dna(rng, int(rng.integers(0, 60)), with_n=0.3)
        assert capi.cut_out_nnns(s) == po.cut_out_nnns(s), s
    # fragmenter: createfilter.hpp:168-173 and the oracle's loop
    s, e = capi.fragment_bounds(72, 100000, 13)
    assert (s.tolist(), e.tolist()) == ([0], [72])
    s, e = capi.fragment_bounds(250001, 100000, 13)
    assert (s.tolist(), e.tolist()) == ([0, 99988, 199988], [100000, 200000, 250001])
    for L, F, k in [(100000, 100000, 13), (100001, 100000, 13), (12, 1000, 13), (0, 10, 13), (1, 10, 3), (2, 10, 3),
                    (5000, 333, 20)]:
        s, e = capi.fragment_bounds(L, F, k)
        o = po.OracleIBF(64, 3, k, 64 * 64)
        assert o.add_sequence(np.zeros(L, dtype=np.uint8), F, 0) == len(s)


def test_large_ibf_file_is_read_by_several_threads(tmp_path):
    """rb_ibf_open reads files of 8 MiB and more with several pread threads (csrc/rb_io.h): same image as the oracle's loader, byte for
    byte the same file when stored again; a file cut short is still a parse error, whichever thread meets the end."""
    rng = np.random.default_rng(5)
    n_bits = 128 * 1400003 + 17
    c = capi.HostIBF.create(70, 3, 13, n_bits)
    w = c.words()
    nw = n_bits // 64
    w[:nw] = rng.integers(0, 1 << 63, size=nw, dtype=np.uint64)
    p = tmp_path / "big.ibf"
    c.store(str(p))
    assert os.path.getsize(p) > (16 << 20)
    h = capi.HostIBF.open(str(p))
    assert np.array_equal(h.words()[:nw], w[:nw])
    o = po.OracleIBF.load(str(p))
    assert np.array_equal(o.words()[:nw], h.words()[:nw])
    q = tmp_path / "again.ibf"
    h.store(str(q))
    assert open(p, "rb").read() == open(q, "rb").read()
    cut = tmp_path / "cut.ibf"
    cut.write_bytes(open(p, "rb").read()[:-4096])
    with pytest.raises(capi.RBError) as ei:
        capi.HostIBF.open(str(cut))
    assert ei.value.status == capi.RB_ERR_PARSE_IBF


def test_ibf_file_io_against_oracle(tmp_path):
    rng = np.random.default_rng(2)
    o = po.OracleIBF(70, 3, 13, 128 * 1000 + 17)  # unaligned metadata
    o.insert(po.encode(H.random_dna(rng, 3000)), 69)
    o.insert(po.encode(H.random_dna(rng, 3000)), 0)
    p = tmp_path / "o.ibf"
    o.store(str(p))
    h = capi.HostIBF.open(str(p))
    assert (h.info["n_bins"], h.info["n_hash"], h.info["kmer_size"], h.info["n_bits"]) == (70, 3, 13, 128 * 1000 + 17)
    assert h.info["bin_width"] == 2 and h.info["n_blocks"] == 1000
    nw = o.n_bits // 64
    assert np.array_equal(h.words()[:nw], o.words()[:nw])
    q = tmp_path / "p.ibf"
    h.store(str(q))
    assert open(p, "rb").read() == open(q, "rb").read()
    # created empty image -> store -> oracle load
    c = capi.HostIBF.create(200, 3, 15, 256 * 77)
    c.words()[5] = np.uint64(0xDEADBEEF)
    r = tmp_path / "c.ibf"
    c.store(str(r))
    g = po.OracleIBF.load(str(r))
    assert (g.n_bins, g.n_hash, g.kmer_size, g.n_bits) == (200, 3, 15, 256 * 77) and int(g.words()[5]) == 0xDEADBEEF
    # error conventions: missing file vs not-an-IBF (configReader.cpp:210-224 depends on the latter)
    with pytest.raises(capi.RBError) as ei:
        capi.HostIBF.open(str(tmp_path / "missing.ibf"))
    assert ei.value.status == capi.RB_ERR_MISSING_FILE
    fa = tmp_path / "x.fasta"
    fa.write_text(">a\n" + "ACGT" * 100 + "\n")
    with pytest.raises(capi.RBError) as ei:
        capi.HostIBF.open(str(fa))
    assert ei.value.status == capi.RB_ERR_PARSE_IBF
    assert not capi.is_ibf_file(str(fa)) and capi.is_ibf_file(str(p))
    trunc = tmp_path / "t.ibf"
    trunc.write_bytes(open(p, "rb").read()[:-8])
    assert not capi.is_ibf_file(str(trunc))


def test_fastmod_header_against_modulo(tmp_path):
    # the Barrett reduction of ibf_spec.h, compiled for the host, against % over edge and random inputs
    src = tmp_path / "fm.cpp"
    src.write_text(r'''
#include <cstdio>
#include <cstdint>
#include <random>
#include "ibf_spec.h"
int main() {
    std::mt19937_64 g(1);
    uint64_t bad = 0, n = 0;
    const uint64_t edge[] = {0, 1, 2, 3, 0xFFFFFFFFull, 0x100000000ull, 0xFFFFFFFFFFFFFFFFull, 0x8000000000000000ull,
                             0x7FFFFFFFFFFFFFFFull, 0xFFFFFFFF00000000ull, 0xFFFFFFFEFFFFFFFFull};
    std::vector<uint32_t> ds = {2, 3, 5, 7, 977, 4099, 65535, 65536, 65537, 2472538, 1u << 23, (1u << 23) + 1, 3000000019u,
                                0x7FFFFFFFu, 0x80000000u, 0x80000001u, 0xFFFFFFFEu, 0xFFFFFFFFu};
    for (int i = 0; i < 2000; ++i) ds.push_back((uint32_t)(g() >> (g() % 31)) | 2u);
    for (uint32_t d : ds) {
        if (d < 2) continue;
        const uint64_t m = rbspec::fastmod_magic(d);
        const bool pow2 = (d & (d - 1)) == 0;
        for (uint64_t x : edge) { ++n; bad += rbspec::fastmod(x, d, m) != x % d; }
        for (uint64_t q = 0; q < 64; ++q) {  // multiples of d and their neighbours
            const uint64_t base = (g() / d) * d;
            for (int64_t o = -1; o <= 1; ++o) { uint64_t x = base + (uint64_t)o; ++n; bad += rbspec::fastmod(x, d, m) != x % d; }
        }
        for (int i = 0; i < 500; ++i) { uint64_t x = g(); ++n; bad += rbspec::fastmod(x, d, m) != x % d; }
        const uint32_t mask = pow2 ? d - 1 : 0xFFFFFFFFu;
        for (int i = 0; i < 200; ++i) {
            uint64_t v = g() % 1220703125ull, pre = rbspec::precalc(13, i % 3);
            uint64_t x = pre * v; x ^= x >> 27;
            ++n; bad += rbspec::block_index(v, pre, d, m, mask) != x % d;
        }
    }
    std::printf("%llu %llu\n", (unsigned long long)n, (unsigned long long)bad);
    return bad != 0;
}
''')
    exe = tmp_path / "fm"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "readbouncer_amd", "csrc"), str(src),
                           "-o", str(exe)])
    out = subprocess.check_output([str(exe)], text=True).split()
    assert int(out[0]) > 1_000_000 and int(out[1]) == 0


def test_odd_metadata_loads_with_a_warning(tmp_path):
    """A file whose metadata tail holds values the reference never writes still parses, but says so (rb_last_warning):
    the order {noOfBins, noOfHashFunc, kmerSize, spare} is recalled from SeqAn, not read from the reference tree."""
    from oracle import pyoracle as po
    good = po.OracleIBF(70, 3, 13, 128 * 1009)
    good.store(str(tmp_path / "good.ibf"))
    odd = po.OracleIBF(70, 2, 13, 128 * 1009)
    odd.store(str(tmp_path / "odd.ibf"))
    h = capi.HostIBF.open(str(tmp_path / "good.ibf"))
    assert capi.last_warning() == "" and h.info["n_hash"] == 3
    h = capi.HostIBF.open(str(tmp_path / "odd.ibf"))
    assert "noOfHashFunc = 2" in capi.last_warning() and h.info["n_hash"] == 2
    assert capi.is_ibf_file(str(tmp_path / "odd.ibf"))


def test_bench_refuses_more_gpus_than_the_node_has():
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "RB_BENCH_SAME_GPU", "RB_BENCH_ENGINE")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "64"], env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 2 and "GPU(s) visible" in p.stderr and p.stdout.strip() == ""


ZERO_SWEEP = r"""
import ctypes as C, sys
from readbouncer_amd import capi
L = capi.lib()
ok_allowed = {"rb_device_count", "rb_is_ibf_file", "rb_calculate_ci", "rb_pack_reads",   # fine with zeros (an empty batch packs to nothing)
              "rb_set_placement_tries"}                                                   # (0 tries = placement by trial off: a valid setting)
report = []
for name, (restype, argtypes) in sorted(capi.SIGNATURES.items()):
    args = []
    for t in argtypes:
        if t in (C.c_double,):
            args.append(0.0)
        elif t in (C.c_int, C.c_uint8, C.c_uint16, C.c_uint32, C.c_uint64, C.c_size_t):
            args.append(0)
        else:
            args.append(None)
    r = getattr(L, name)(*args)
    report.append((name, r if restype is not None else None))
    if restype is C.c_int and name not in ok_allowed and r == 0:
        print("RETURNED_OK", name); sys.exit(3)
print("SWEPT", len(report))
"""


def test_every_entry_point_survives_null_and_zero_arguments():
    """Every function of the C ABI called with NULL for each pointer and 0 for each scalar: no crash (the sweep runs in a child
    process, a segfault would show as its exit code), and every status-returning function refuses (nothing reports RB_OK for a
    NULL handle).  Runs with or without a GPU."""
    import sys
    r = subprocess.run([sys.executable, "-c", ZERO_SWEEP], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert "SWEPT %d" % len(capi.SIGNATURES) in r.stdout

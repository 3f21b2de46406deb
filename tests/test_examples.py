"""The C ABI from plain C: include/*.h must be consumable by a C99 compiler (what cgo and JNI's javah-side C see), and a
program written against it alone -- no Python, no C++ -- must build filters, classify and get check_unblock's decisions
(src/main/adaptive_sampling.hpp:35-113)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "readbouncer_amd")
EXAMPLE = os.path.join(ROOT, "examples", "adaptive_sampling_c_abi.c")


def _build(tmp_path):
    exe = str(tmp_path / "adaptive_sampling_c_abi")
    subprocess.check_call(["gcc", "-std=c99", "-O1", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
                           EXAMPLE, "-L", LIBDIR, "-lreadbouncer_amd", "-Wl,-rpath," + LIBDIR, "-o", exe])
    return exe


def test_headers_are_c99(tmp_path):
    src = tmp_path / "hdr.c"
    src.write_text('#include "readbouncer_amd.h"\n#include "readbouncer_amd_tuning.h"\nint main(void) { return 0; }\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I",
                           os.path.join(ROOT, "include"), str(src)])


def test_c_example_links_and_refuses_without_a_gpu(tmp_path):
    from readbouncer_amd import capi
    exe = _build(tmp_path)
    if capi.device_count() > 0:
        pytest.skip("GPU present: covered by the gpu test")
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 2, (p.returncode, p.stdout, p.stderr)  # loud refusal, not a CPU path
    assert "no CPU fallback" in p.stderr


@pytest.mark.gpu
def test_c_example_decisions(tmp_path):
    exe = _build(tmp_path)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, (p.returncode, p.stdout, p.stderr)
    assert "96 reads, 0 unexpected decisions" in p.stdout

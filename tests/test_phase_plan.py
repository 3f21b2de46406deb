"""The planner of the clock-phased gathers (readbouncer_amd/csrc/rb_phase_plan.h) pinned on a CPU: tests/cpp/dump_phase_plan.cpp
walks the one table of rules over a grid of kernel shapes, block widths, table sizes and read lengths and prints a digest per
(shape, block width) plus a dozen rows in full; tests/golden/phase_plan.txt holds what round 3's fitted rules give (generated
from the if-chains they were written as, before they became the table).  A change of a rule is a change of this fixture --
to be made together with a profiles/phase_rule_check.py run that justifies it."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_phase_rules_match_the_golden_table(tmp_path):
    exe = str(tmp_path / "dump_phase_plan")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", os.path.join(ROOT, "tests", "cpp", "dump_phase_plan.cpp"), "-o", exe])
    got = subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines()
    exp = open(os.path.join(ROOT, "tests", "golden", "phase_plan.txt")).read().splitlines()
    assert len(got) == len(exp) == 24
    for g, e in zip(got, exp):
        assert g.split("  (")[0] == e.split("  (")[0], (g, e)  # (the row names in brackets are commentary)


def test_every_shape_has_a_named_row():
    hdr = open(os.path.join(ROOT, "readbouncer_amd", "csrc", "rb_phase_plan.h")).read()
    for name in ("General", "FourTiles", "Rounds", "SixTiles", "WideRounds", "WideFourTiles", "Wide3FourTiles", "Wide3Rounds"):
        assert hdr.count("PhaseShape::" + name) >= 1, name
    assert "shape == 1" not in hdr and "shape == 5" not in hdr  # no integer-coded shapes left
    eng = open(os.path.join(ROOT, "readbouncer_amd", "csrc", "rb_engine.hip")).read()
    assert "shape == 5" not in eng and "shape == 6" not in eng


import sys

import pytest


@pytest.mark.gpu
@pytest.mark.gpuperf
def test_phase_rules_hold_between_the_fitted_points():
    """the planner guard as a test (timing: -m gpuperf only): profiles/phase_rule_check.py at its default 16 points exits 0, i.e. the
    rule is within 8 % of the best of a fresh sweep (and never slower than the plain kernel) on THIS box"""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "phase_rule_check.py")], capture_output=True, text=True, timeout=1500)
    print(p.stdout[-3000:])
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]

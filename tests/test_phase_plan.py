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


def test_equal_length_slices_of_the_four_word_builds(tmp_path):
    """phase_equal_slices (rb_phase_plan.h): the four-word one-lane builds get fewer, equal-length slices only where cutting the table
    at 4 MiB is wasteful -- at least two slices more than slices of up to 4.75 MiB need; the points are the measured ones of
    profiles/r04/equal_slices_check.txt (README table 37.73 MiB: 10 -> 8)"""
    src = tmp_path / "eq.cpp"
    src.write_text('''
#include <cstdint>
#include <cstdio>
#include "%s"
using namespace rbplan;
int main() {
    const double mib[] = {24, 28, 32, 34, 37.73, 38.5, 41.5, 44, 46, 48, 60, 100};
    for (double m : mib) {
        const uint64_t b = (uint64_t)(m * 1048576.0);
        if (phase_equal_slices(PhaseShape::Wide3FourTiles, 22, b) != phase_equal_slices(PhaseShape::WideFourTiles, 22, b) ||
            phase_equal_slices(PhaseShape::Wide3Rounds, 22, b) != phase_equal_slices(PhaseShape::WideRounds, 22, b)) return 1;  // the three-word builds: the same cut
        std::printf("%%.2f %%u %%u %%u %%u %%llu %%llu\\n", m, phase_equal_slices(PhaseShape::WideFourTiles, 22, b), phase_equal_slices(PhaseShape::WideRounds, 22, b),
                    phase_equal_slices(PhaseShape::WideFourTiles, 21, b), phase_equal_slices(PhaseShape::FourTiles, 22, b),
                    (unsigned long long)phase_equal_slices_ticks(PhaseShape::WideFourTiles, 2, 8, 238), (unsigned long long)phase_equal_slices_ticks(PhaseShape::WideRounds, 2, 8, 348));
    }
}
''' % os.path.join(ROOT, "readbouncer_amd", "csrc", "rb_phase_plan.h"))
    exe = str(tmp_path / "eq")
    subprocess.check_call(["g++", "-O1", "-std=c++17", str(src), "-o", exe])
    rows = [l.split() for l in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines()]
    got = {float(r[0]): int(r[1]) for r in rows}
    assert got[24] == got[28] == got[32] == got[34] == got[38.5] == got[44] == 0 and got[37.73] == 8 and got[41.5] == 9 and got[46] == 10
    for r in rows:
        m, n4, nr, n21, nft = float(r[0]), int(r[1]), int(r[2]), int(r[3]), int(r[4])
        assert n4 == nr and n21 == 0 and nft == 0  # the same for both four-word shapes; only with 4 MiB slices; no other shape
        if n4:
            pow2 = -(-int(m * 1048576) // (4 << 20))
            assert n4 + 2 <= pow2 and m / n4 <= 4.75 + 1e-9
        assert int(r[5]) == 500 and int(r[6]) == 420  # eight slices: the four-tile rule's 500 ticks; rounds of three tiles not below 3 360 / 8
    hdr = open(os.path.join(ROOT, "readbouncer_amd", "csrc", "rb_phase_plan.h")).read()
    assert "5200u / n" in hdr and "3600u / n" in hdr  # the three-word builds' window floors (profiles/r04/equal_slices_three_word_builds.txt)


def test_equal_length_slices_of_large_one_word_tables(tmp_path):
    """phase_equal_slices_one_word (rb_phase_plan.h): one-word tables from 50 MiB on, four- and six-tile builds only -- slice counts and
    windows at the sizes of profiles/r05/one_word_equal_slices*.txt"""
    src = tmp_path / "ow.cpp"
    src.write_text('''
#include <cstdint>
#include <cstdio>
#include "%s"
using namespace rbplan;
int main() {
    const double mib[] = {40, 48, 49.9, 50, 56, 64, 80, 96, 112, 127};
    for (double m : mib) {
        const uint64_t b = (uint64_t)(m * 1048576.0);
        const uint32_t n4 = phase_equal_slices_one_word(PhaseShape::FourTiles, 0, 22, b), n6 = phase_equal_slices_one_word(PhaseShape::SixTiles, 0, 22, b);
        std::printf("%%.1f %%u %%u %%u %%u %%u %%llu %%llu %%llu\\n", m, n4, n6, phase_equal_slices_one_word(PhaseShape::FourTiles, 1, 22, b),
                    phase_equal_slices_one_word(PhaseShape::FourTiles, 0, 21, b), phase_equal_slices_one_word(PhaseShape::Rounds, 0, 22, b),
                    (unsigned long long)phase_equal_slices_one_word_ticks(PhaseShape::FourTiles, n4 ? n4 : 1, b, 238),
                    (unsigned long long)phase_equal_slices_one_word_ticks(PhaseShape::SixTiles, n6 ? n6 : 1, b, 348),
                    (unsigned long long)phase_equal_slices_one_word_ticks(PhaseShape::FourTiles, n4 ? n4 : 1, b, 188));
    }
}
''' % os.path.join(ROOT, "readbouncer_amd", "csrc", "rb_phase_plan.h"))
    exe = str(tmp_path / "ow")
    subprocess.check_call(["g++", "-O1", "-std=c++17", str(src), "-o", exe])
    rows = {float(r[0]): [int(x) for x in r[1:]] for r in (l.split() for l in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines())}
    assert [rows[m][0] for m in (40, 48, 49.9, 50, 56, 64, 80, 96, 112, 127)] == [0, 0, 0, 11, 11, 12, 13, 14, 15, 16]
    for m, r in rows.items():
        assert r[0] == r[1] and r[2] == r[3] == r[4] == 0  # both one-word builds alike; not two-word blocks, not 2 MiB slices, not the rounds
        if r[0]:
            assert m / r[0] > 4.0  # slices longer than an L2 ...
            assert 0.9 * (5600 + 54 * m) / r[0] <= r[5] <= (5600 + 54 * m) / r[0] + 1  # ... a cycle of 5 600 + 54 per MiB ticks
            assert 1.2 < r[6] / r[5] < 1.28 and 0.85 < r[7] / r[5] < 0.95  # six tiles x 1.24; 200 bp reads leave the shape partly empty
    assert 660 <= rows[80][5] <= 800 and 700 <= rows[127][5] <= 800  # (the measured optima: 769-807 ticks at 80 MiB / 13 slices, 750 at 127 MiB / 16)


def test_equal_length_slices_of_the_two_word_lds_offset_builds(tmp_path):
    """phase_multi_equal_slices / phase_multi_equal_ticks (rb_phase_plan.h, round 6): only one- and two-word blocks whose rule asks for 4 MiB slices;
    slices of 2.3-3.3 MiB; four tiles: a window of about 500 ticks whatever the table, never below 1.05 x 168 ticks per MiB of slice; six
    tiles: a cycle of 2 350 + 130 per MiB ticks (+ 8 %); slice counts and windows at the measured points of
    profiles/r06/multi/equal_slices_fit_two_word.txt"""
    src = tmp_path / "mw.cpp"
    src.write_text('''
#include <cstdint>
#include <cstdio>
#include "%s"
using namespace rbplan;
int main() {
    const double mib[] = {13, 16, 18.9, 22, 26, 31};
    const uint32_t kmers[] = {188, 238, 288, 348};
    for (double m : mib)
        for (uint32_t km : kmers) {
            const uint64_t b = (uint64_t)(m * 1048576.0);
            const PhaseShape sh = km <= 256 ? PhaseShape::FourTiles : PhaseShape::SixTiles;
            const uint32_t n = phase_multi_equal_slices(sh, 1, 22, b, km);
            std::printf("%%.1f %%u %%u %%llu %%u %%u %%u\\n", m, km, n, (unsigned long long)phase_multi_equal_ticks(sh, n ? n : 1, b, km),
                        phase_multi_equal_slices(sh, 1, 21, b, km), phase_multi_equal_slices(sh, 0, 22, b, km), phase_multi_equal_slices(PhaseShape::WideFourTiles, 2, 22, b, km));
        }
}
''' % os.path.join(ROOT, "readbouncer_amd", "csrc", "rb_phase_plan.h"))
    exe = str(tmp_path / "mw")
    subprocess.check_call(["g++", "-O1", "-std=c++17", str(src), "-o", exe])
    rows = {(float(r[0]), int(r[1])): [int(x) for x in r[2:]] for r in (l.split() for l in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines())}
    for (m, km), (n, ticks, n21, n_one_word, n_wide) in rows.items():
        assert n21 == n_wide == 0 and n_one_word == n  # 4 MiB rule only; one- and two-word blocks alike; not the wide shapes
        assert 2.1 <= m / n <= 3.4, (m, km, n)
        if km <= 256:
            assert 1.05 * 168 * (m / n) * 0.95 <= ticks <= 600 and ticks >= 460, (m, km, n, ticks)
        else:
            assert 0.9 * (2350 + 130 * m) <= ticks * n <= 1.09 * (2350 + 130 * m), (m, km, n, ticks)
    # the measured optima (slices, window): the rule's cut is the best or second-best one measured, its window 0-10 % above the optimum
    assert rows[(18.9, 238)][0] == 8 and 500 <= rows[(18.9, 238)][1] <= 540      # 7.66 ms at 8 x 500 ticks (4 MiB slices: 8.41)
    assert rows[(22.0, 238)][0] == 9 and 500 <= rows[(22.0, 238)][1] <= 550      # 7.97 at 9 x 511
    assert rows[(31.0, 238)][0] in (10, 11) and 490 <= rows[(31.0, 238)][1] <= 580
    assert rows[(18.9, 188)][0] == 7 and 470 <= rows[(18.9, 188)][1] <= 520      # 6.64 at 7 x 485
    assert rows[(18.9, 348)][0] == 7 and 700 <= rows[(18.9, 348)][1] <= 760      # 10.95 at 7 x 687; the merged form's cliff ends at ~700
    assert rows[(26.0, 348)][0] == 9 and 621 <= rows[(26.0, 348)][1] <= 690      # 12.24 at 9 x 621
    assert rows[(18.9, 288)][0] == 7 and 640 <= rows[(18.9, 288)][1] <= 730      # 9.93 at 7 x 631, 10.28 at 687


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


@pytest.mark.gpu
@pytest.mark.gpuperf
def test_calibration_never_makes_it_slower():
    """rb_engine_calibrate on the shapes where round 4's first versions went wrong (a window measured on low-entropy reads, or at
    another batch size, walked two-word tables off a cliff: +13 ... +35 %): K1 after calibration is at most 3 % above K1 before"""
    import numpy as np
    torch = pytest.importorskip("torch")
    from readbouncer_amd import capi, synth
    dev = torch.device("cuda:0")
    N = 500_000
    cases = []
    targets = [synth.build_device_filter(0, synth.WORKLOADS[k], fill_seed=12 + i, plant_seed=111 + i, n_segments=512)[0]
               for i, k in enumerate(("mock_t1", "mock_t2", "mock_t3"))]
    cases.append(("three targets, packed two-word table", [], targets, 200))
    big = capi.DeviceIBF.create(0, 128, 3, 13, 128 * (int(45 * (1 << 20) / 16) - 3))
    big.fill_synth(3)
    cases.append(("two-word filter of 45 MiB", [big], [], 250))
    for name, dep, tgt, L in cases:
        seqs, offs, lens = synth.make_reads_device(5, N, L, None, dev)
        mc = torch.zeros((N, len(dep) + len(tgt)), dtype=torch.int16, device=dev)
        eng = capi.Engine(0, dep, tgt)
        eng.set_timing(True)

        def k1():
            ts = []
            for it in range(6):
                eng.kernel_time()
                eng.classify_device(seqs.data_ptr(), offs.data_ptr(), lens.data_ptr(), N, L, d_maxcount=mc.data_ptr())
                torch.cuda.synchronize()
                ms, calls = eng.kernel_time()
                if it:
                    ts.append(ms / calls)
            return float(np.median(ts))
        before = k1()
        ref = mc.clone()
        n_tables, n_changed = eng.calibrate(N, L)
        after = k1()
        print("%s: %.2f ms -> %.2f ms, %d table(s), %d changed, window %d ticks" % (name, before, after, n_tables, n_changed,
                                                                                    eng.plan(0, N, L)["phase_window_ticks"]))
        assert n_tables == 1 and torch.equal(ref, mc)
        assert after <= 1.03 * before, (name, before, after)
        eng.destroy()

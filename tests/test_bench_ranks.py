"""The N>1 path of bench.py: `python bench.py --gpus N` starts N ranks itself (children created before the parent touches
the GPU), the ranks meet in a torch.distributed group, time = max over ranks, rank 0 prints ONE JSON line.

CPU tests (world size 2, gloo): the rank flow runs with RB_BENCH_ENGINE=none -- a control-flow hook with NO classification
behind it (there is no CPU implementation of the hot path to fall back to), so what is under test is the launcher, the
rendezvous, the barriers, the timing reduction, the per-rank gather and the all-gather + max of the bin-sharded layout.
GPU tests (`-m gpu`): the same command with the real engine, two ranks on the one GPU of the box (RCCL refuses duplicate
devices, so the collectives go through gloo there), decisions equal to the 1-rank run."""
import json
import os
import socket
import subprocess
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _strict(text):
    """strict JSON: NaN / Infinity tokens are refused like a non-Python parser would refuse them"""
    def refuse(tok):
        raise ValueError("non-standard JSON token %s" % tok)
    return json.loads(text, parse_constant=refuse)


def _run(argv, env_extra, timeout=240, launcher=None):
    """-> (process, full result).  bench.py prints ONE bounded line (the driver's record) and writes everything else to the sidecar
    file RB_BENCH_DETAIL names; the tests read the sidecar and find the line itself under "_line" / "_compact"."""
    import tempfile
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    fd, side = tempfile.mkstemp(prefix="rb_bench_detail_", suffix=".json")
    os.close(fd)
    os.unlink(side)
    env["RB_BENCH_DETAIL"] = side
    env.update(env_extra)
    cmd = (launcher or [sys.executable]) + [BENCH] + argv
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        if not lines:
            return p, None
        compact = _strict(lines[-1])
        d = _strict(open(side).read()) if os.path.exists(side) else dict(compact)
        d["_line"], d["_compact"] = lines[-1], compact
        return p, d
    finally:
        if os.path.exists(side):
            os.unlink(side)


def _check_line(d, n_gpus):
    """the contract of the final line: bounded, strict JSON, the headline with its roofline, CPU baseline and
    parity keys, one small row per other leg"""
    line, c = d["_line"], d["_compact"]
    assert len(line.encode()) <= 4096, len(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert key in c, key
    assert c["n_gpus"] == n_gpus and c["value"] == pytest.approx(d["value"], rel=1e-4)
    assert isinstance(c["config"]["workload"], str) and "model" not in c["config"]
    for leg, row in (c.get("other_configs") or {}).items():
        assert set(row) <= {"value", "ms_per_step", "frac", "frac_of_read_peak", "Glines_per_s", "request_bound_frac", "p99_ms", "live_p99_ms", "cpu_reads_per_s",
                            "parity_ok", "checked_reads", "error", "work_skipped"}, (leg, row)
        assert ("work_skipped" not in row or leg == "c3_early") and not (row.get("work_skipped") and "frac" in row)  # skipped work never carries a roofline fraction
        assert len(json.dumps(row)) < 300
    return c


CPU_HOOKS = {"RB_BENCH_ENGINE": "none", "RB_BENCH_BACKEND": "gloo"}


def test_gpus2_self_launch_read_sharded_cpu():
    p, d = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--reads", "400"], CPU_HOOKS)
    assert p.returncode == 0, p.stderr[-2000:]
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["ranks"]["self_launched"] is True and d["ranks"]["backend"] == "gloo"
    assert len(d["ranks"]["per_rank_reads_per_s"]) == 2 and all(x > 0 for x in d["ranks"]["per_rank_reads_per_s"])
    assert "no classification ran" in d["engine"]  # the hook can never be mistaken for a measurement
    # exactly one JSON line on stdout
    assert len([l for l in p.stdout.splitlines() if l.strip()]) == 1
    _check_line(d, 2)
    # whole-job aggregate: both ranks' reads over the max time
    assert abs(d["value"] - 2 * 400 * 3 / (d["ms_per_step"] * 3 / 1e3)) / d["value"] < 1e-6


def test_gpus2_default_line_structure_cpu():
    """the DEFAULT command (what the driver runs at N > 1): headline = config 3, the other BASELINE configs as sub-runs of all
    ranks, per-rank device records -- the structure a SCALE record will carry, checked on the null engine"""
    p, d = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], CPU_HOOKS)
    assert p.returncode == 0, p.stderr[-2000:]
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "c3" in d["config"]["workload"]
    for key in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data",
                "roofline", "cpu_baseline", "parity", "ranks", "other_configs"):
        assert key in d, key
    # N > 1 carries the multi-GPU configs and the pool legs only (four SCALE runs back to back); N = 1 carries every leg
    assert set(d["other_configs"]) == {"c3np2", "c4", "c5", "pool_c3", "pool_c4"}
    c = _check_line(d, 2)
    assert set(c["other_configs"]) == set(d["other_configs"]) and len(c["ranks"]["per_rank_reads_per_s"]) == 2
    for sub in d["other_configs"].values():
        assert sub["n_gpus"] == 2 and sub["value"] > 0
    # the one-process pool legs: rank 0 measures alone, the other rank waits for it on the HOST (rendezvous store), not in a
    # collective that would spin on its GPU
    for leg in ("pool_c3", "pool_c4"):
        assert d["other_configs"][leg]["rank0_only"] is True and d["other_configs"][leg]["waited_on_store"] is True
    r = d["ranks"]
    assert "rccl_ranks" in r and "xgmi_preflight" in r and len(r["devices"]) == 2 and len(r["per_rank_reads_per_s"]) == 2
    assert sorted(x["rank"] for x in r["devices"]) == [0, 1]


def test_visible_gpu_count_does_not_touch_hip():
    """the launcher counts GPUs from the environment lists / the KFD topology, never through the HIP runtime"""
    code = ("import sys, os; sys.path.insert(0, %r); os.environ['HIP_VISIBLE_DEVICES'] = '0,1,2'; import bench; "
            "assert bench.visible_gpu_count() == 3; assert 'torch' not in sys.modules; "
            "del os.environ['HIP_VISIBLE_DEVICES']; n = bench.visible_gpu_count(); assert n is None or n >= 0; "
            "assert 'torch' not in sys.modules") % ROOT
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stderr[-2000:]


def test_gpus2_self_launch_bin_sharded_cpu():
    p, d = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "300", "--bin-sharded"], CPU_HOOKS)
    assert p.returncode == 0, p.stderr[-2000:]
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["bin_sharded_reduce_ok"] is True  # all-gather of the u16 partial tables + max == max of what every rank made


def test_under_torchrun_cpu():
    """the driver's own launch line: python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                "127.0.0.1", "--master-port", str(port)]
    p, d = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "200"], CPU_HOOKS, launcher=launcher)
    assert p.returncode == 0, p.stderr[-2000:]
    assert d["n_gpus"] == 2 and d["ranks"]["self_launched"] is False


def test_dead_rank_does_not_hang_the_launcher():
    """a rank that dies: the launcher ends the others, exits non-zero and still leaves ONE parseable line with an `error` field"""
    p, d = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "200"],
                dict(CPU_HOOKS, RB_BENCH_TEST_DIE_RANK="1"), timeout=120)
    assert p.returncode != 0
    # (rank 0 notices the broken collective itself, or the launcher reports that rank 0 left no line: either way a reason)
    assert d is not None and d["_compact"]["value"] == 0 and d["_compact"]["error"]
    assert len(d["_line"]) <= 4096
    p, d = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "200"],
                dict(CPU_HOOKS, RB_BENCH_TEST_DIE_RANK="0"), timeout=120)
    assert p.returncode != 0 and d is not None and d["_compact"]["value"] == 0 and d["_compact"]["error"]


def test_headline_survives_a_process_that_dies_later_cpu():
    """Rank 0 measures as a child of a supervisor and checkpoints its result after the headline and after every leg: a process that
    dies later (SIGSEGV here, as a GPU fault in a sub-leg would end it) still leaves ONE parseable line -- the headline with an
    `error` -- and a non-zero exit code.  N = 1 (the driver's own command), N = 2 self-launched (rank 0 dies; then rank 1 dies and
    takes the job down), and under torchrun, which answers a dead rank by sending SIGTERM to the others."""
    p, d = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--reads", "300"], dict(CPU_HOOKS, RB_BENCH_TEST_DIE_LATE="0"), timeout=120)
    assert p.returncode != 0 and d is not None
    c = d["_compact"]
    assert c["value"] > 0 and c["steps"] == 2 and "signal 11" in c["error"] and len(d["_line"]) <= 4096
    assert len([l for l in p.stdout.splitlines() if l.startswith("{")]) == 1
    for die in ("0", "1"):
        p, d = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "300"], dict(CPU_HOOKS, RB_BENCH_TEST_DIE_LATE=die), timeout=120)
        assert p.returncode != 0 and d is not None, die
        assert d["_compact"]["value"] > 0 and d["_compact"]["n_gpus"] == 2 and d["_compact"]["error"], die
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    p, d = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--reads", "300"], dict(CPU_HOOKS, RB_BENCH_TEST_DIE_LATE="1"), timeout=180,
                launcher=launcher)
    assert p.returncode != 0 and d is not None, p.stderr[-1500:]
    assert d["_compact"]["value"] > 0 and d["_compact"]["error"]


@pytest.mark.parametrize("n", [1, 8])
def test_default_line_is_bounded_cpu(n):
    """the DEFAULT command at N = 1 and N = 8 on the null engine: one line, at most 4 KB, strict JSON, roofline / cpu_baseline /
    parity keys present, every leg of the run summarised in a row; the sidecar holds the rest"""
    p, d = _run(["--gpus", str(n), "--steps", "2", "--warmup", "1"], CPU_HOOKS, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert len([l for l in p.stdout.splitlines() if l.strip()]) == 1
    c = _check_line(d, n)
    import bench
    assert tuple(c["other_configs"]) == (bench.FULL_LEGS if n == 1 else bench.MULTI_LEGS)
    assert "grch38_f100k" in bench.FULL_LEGS  # the reference-default human filter rides in the default N = 1 run
    if n > 1:
        assert len(c["ranks"]["per_rank_reads_per_s"]) == n and len(d["ranks"]["devices"]) == n


def test_compact_line_sheds_detail_but_keeps_the_contract():
    """compact_line on a result as fat as round 4's (29.6 KB: plans, probes, request bounds, pool statistics, per-rank device records
    of eight ranks): <= 4 KB, the headline's roofline.frac and cpu_baseline.value intact, request bounds never under a key `frac`"""
    import bench
    plan = [{"kernel": "ibf_count_max_phased_kernel", "table_bytes": 41943040, "merged_members": 4, "phased": 1,
             "phase_shape_name": "four tiles, four-word one-lane" * 2, "phase_slices": 8, "phase_window_ticks": 575}] * 4
    def leg(i):
        return {"metric": bench.METRIC, "value": 1e6 * (i + 1) + 0.123456789, "unit": "reads/s", "n_gpus": 8, "steps": 5, "warmup": 1,
                "ms_per_step": 12.3456789, "config": {"workload": "w" * 400, "filters": [{"n_bins": 8192}] * 4, "decisions": [1, 2, 3]},
                "roofline": {"bound": "hbm", "achieved": 3700.123456, "peak": 8000.0, "unit": "GB/s", "frac": 0.4625154, "traffic": 1.0e13,
                             "traffic_source": "s" * 300, "plan": plan, "read_peak_probe": {"GBps": 6900.5, "source": "x" * 300},
                             "frac_of_measured_read_peak": 0.9981,
                             "request_bound": {"request_bound_frac": 0.93, "source": "y" * 400, "l2_Grequests_per_s": 250.0}},
                "cpu_baseline": {"value": 15400.7, "unit": "reads/s", "cores": 16, "kind": "port", "sample": "z" * 300},
                "parity": {"checked_reads": 220000, "decision_mismatches": 0, "raw_max_mismatches": 0, "near_threshold_reads": 2800,
                           "against": "a" * 200},
                "latency": {"by_batch": {"64": {"p50_ms": 0.066, "p99_ms": 0.081}, "1024": {"p50_ms": 0.38, "p99_ms": 0.39}}},
                "setup_s": 1.0}
    head = leg(0)
    head["ranks"] = {"backend": "nccl", "rccl_ranks": 8, "self_launched": False, "per_rank_reads_per_s": [3.2e6 + i for i in range(8)],
                     "devices": [{"rank": i, "device": i, "name": "AMD Instinct MI355X", "uuid": "u" * 36, "peer_access_possible": [True] * 8}
                                 for i in range(8)], "xgmi_preflight": {"ran": True, "GBps": 50.0}}
    head["other_configs"] = {name: leg(i + 1) for i, name in enumerate(bench.FULL_LEGS)}
    head["other_configs"]["pool_c3"]["parity"] = {"pool_outputs_equal_single_engine": True, "oracle_mismatches": 0, "checked_reads": 5}
    head["other_configs"]["c5"]["latency"] = {"p99_ms": 0.16, "p50_ms": 0.06}
    head["other_configs"]["c5"]["parity"] = {"replayed_decisions_equal_one_batch": True, "checked_reads": 300000}
    assert len(json.dumps(head)) > 25000
    line = bench.compact_line(head, "bench_detail.json")
    assert len(line.encode()) <= bench.COMPACT_LIMIT == 4096
    c = _strict(line)
    assert c["roofline"]["frac"] == pytest.approx(0.4625154, rel=1e-5) and c["cpu_baseline"]["value"] == pytest.approx(15400.7)
    assert c["roofline"]["request_bound_frac"] == 0.93 and c["parity"]["decision_mismatches"] == 0
    assert set(c["other_configs"]) == set(bench.FULL_LEGS)
    assert c["other_configs"]["c5"]["p99_ms"] == 0.16 and c["other_configs"]["pool_c3"]["parity_ok"] is True
    assert all(r["parity_ok"] is True for r in c["other_configs"].values())
    assert c["detail"] == "bench_detail.json" and "plan" not in line and "request_roofline" not in line
    # NaN / Infinity never reach the line; a mismatch shows
    head["value"] = float("nan")
    head["other_configs"]["c4"]["parity"]["raw_max_mismatches"] = 3
    c = _strict(bench.compact_line(head))
    assert c["value"] is None and c["other_configs"]["c4"]["parity_ok"] is False
    # and far beyond anything the run produces (64 ranks, 40 legs) the bound still holds
    head["value"] = 1.0
    head["ranks"]["per_rank_reads_per_s"] = [3.2e6] * 64
    head["other_configs"] = {"leg%d" % i: leg(i) for i in range(40)}
    assert len(bench.compact_line(head, "bench_detail.json")) <= 4096


GPU_HOOKS = {"RB_BENCH_BACKEND": "gloo", "RB_BENCH_SAME_GPU": "1", "RB_BENCH_DUMP_DECISIONS": "1"}
SMALL = ["--workload", "zymo", "--reads", "20000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-latency"]


@pytest.mark.gpu
def test_gpus2_real_engine_decisions_equal_one_rank():
    p1, d1 = _run(["--gpus", "1"] + SMALL, {"RB_BENCH_DUMP_DECISIONS": "1"}, timeout=600)
    assert p1.returncode == 0, p1.stderr[-2000:]
    p2, d2 = _run(["--gpus", "2"] + SMALL, GPU_HOOKS, timeout=600)
    assert p2.returncode == 0, p2.stderr[-2000:]
    assert d2["n_gpus"] == 2 and d2["ranks"]["self_launched"] is True and len(d2["ranks"]["per_rank_reads_per_s"]) == 2
    # rank 0 of the read-sharded run holds the same shard (seed) as the single rank
    assert d2["config"]["decisions_sha1"] == d1["config"]["decisions_sha1"]
    assert d2["config"]["decisions"] == d1["config"]["decisions"] and min(d1["config"]["decisions"][:2]) > 0
    p3, d3 = _run(["--gpus", "2", "--bin-sharded"] + SMALL, GPU_HOOKS, timeout=600)
    assert p3.returncode == 0, p3.stderr[-2000:]
    assert d3["n_gpus"] == 2 and d3["scaling"] == "strong"
    # every rank counts its word columns of every block; all-gather + max in the decision kernel == the unsharded run
    assert d3["config"]["decisions_sha1"] == d1["config"]["decisions_sha1"]


@pytest.mark.gpu
def test_rccl_code_paths_with_a_group_of_one():
    """RCCL refuses two ranks on one GPU, so the box cannot run `--gpus 2` over RCCL; a process group of ONE rank still
    drives every RCCL call of bench.py (init with device_id, barrier, max-reduce, gathers, the bin-sharded all-gather of
    u16 maxima enqueued on the engine's stream + the decision over gathered parts) on real hardware."""
    p1, d1 = _run(["--gpus", "1"] + SMALL, {"RB_BENCH_DUMP_DECISIONS": "1"}, timeout=600)
    assert p1.returncode == 0, p1.stderr[-2000:]
    for extra in ([], ["--bin-sharded"]):
        p, d = _run(["--gpus", "1"] + extra + SMALL, {"RB_BENCH_DUMP_DECISIONS": "1", "RB_BENCH_FORCE_GROUP": "1"}, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        assert d["ranks"]["backend"] == "nccl" and d["n_gpus"] == 1
        assert d["config"]["decisions_sha1"] == d1["config"]["decisions_sha1"]
        assert d["scaling"] == ("strong" if extra else "weak")


@pytest.mark.gpu
def test_gpus2_default_line_decisions_equal_one_rank_on_c3_and_c4():
    """The default command at N = 2 (two ranks on the one GPU of the box, gloo) against N = 1: rank 0's decisions on the
    headline (config 3, the 8 GiB filter) and on config 4 are the 1-rank decisions; the line has the SCALE structure.
    Batches divided by 50 (test hook) so that both runs fit a unit test."""
    small = {"RB_BENCH_READS_DIVISOR": "50", "RB_BENCH_DUMP_DECISIONS": "1"}
    argv = ["--steps", "2", "--warmup", "1", "--cpu-seconds", "2"]
    p1, d1 = _run(["--gpus", "1"] + argv, small, timeout=900)
    assert p1.returncode == 0, p1.stderr[-2000:]
    p2, d2 = _run(["--gpus", "2"] + argv, dict(GPU_HOOKS, **small), timeout=900)
    assert p2.returncode == 0, p2.stderr[-2000:]
    assert d1["test_reads_divisor"] == 50 and d2["n_gpus"] == 2 and "config3" in d2["config"]["workload"]
    assert d1["cpu_baseline"]["value"] > 0 and d2["cpu_baseline"] is None  # timed at N = 1 only
    for d in (d1, d2):
        assert d["parity"]["decision_mismatches"] == 0 and d["parity"]["raw_max_mismatches"] == 0 and d["parity"]["checked_reads"] > 0
        assert d["parity"]["near_threshold_reads"] > 0  # the sample holds reads on which a count off by a few would flip the decision
        import bench
        legs = bench.FULL_LEGS if d is d1 else bench.MULTI_LEGS  # N > 1 carries the multi-GPU configs and the pool legs only
        assert tuple(d["other_configs"]) == legs
        for leg in legs:
            assert "error" not in d["other_configs"][leg], (leg, d["other_configs"][leg].get("error"))
            if leg.startswith("pool_") or leg == "c5":
                continue
            par = d["other_configs"][leg]["parity"]
            assert par["decision_mismatches"] == 0 and par["raw_max_mismatches"] == 0 and par["checked_reads"] > 0, leg
            if d is d1 and leg not in ("cli_readme", "c3_early"):  # (the host CLI's leg is checked against the oracle's chunk driver; it has no CPU-throughput twin)
                assert d["other_configs"][leg]["cpu_baseline"]["value"] > 0, leg  # every throughput leg has its CPU figure at N = 1
        for leg in ("pool_c3", "pool_c4"):  # one process through rb_pool: outputs equal to a single engine's AND to the oracle's
            par = d["other_configs"][leg]["parity"]
            assert par["pool_outputs_equal_single_engine"] is True and par["oracle_mismatches"] == 0 and par["oracle_checked_reads"] > 0, leg
            assert d["other_configs"][leg]["pool"]["per_device"][0]["reads"] > 0
            assert d["other_configs"][leg]["roofline"]["frac"] > 0 and d["_compact"]["other_configs"][leg]["frac"] > 0
        _check_line(d, d["n_gpus"])
        if d is d1:
            early = d["other_configs"]["c3_early"]  # the opt-in mode's leg: labelled as skipped work in the detail AND in the line, never a roofline
            assert early["work_skipped"] is True and early["roofline"] is None and d["_compact"]["other_configs"]["c3_early"]["work_skipped"] is True
            assert early["parity"]["oracle_mismatches"] == 0 and early["parity"]["decision_mismatches"] == 0
            g = d["other_configs"]["grch38_f100k"]  # the reference-default human filter: W = 485, non-power-of-two block count
            assert g["config"]["filters"][0]["n_bins"] == 31000 and g["parity"]["near_threshold_reads"] > 0
        assert "generic modulus" in d["other_configs"]["c3np2"]["config"]["workload"]
        assert d["other_configs"]["c3np2"]["config"]["filters"][0]["bytes"] % (1 << 20) != 0  # BinSizeBits x 8256: not a power of two
        # (latency SLO and keep-up of the c5 leg: wall-clock figures, asserted by test_c5_leg_meets_its_slo under -m gpuperf)
        assert d["other_configs"]["c5"]["parity"]["replayed_decisions_equal_one_batch"] is True
        assert d["other_configs"]["c5"]["live_step"]["concatenated_share"] > 0.3
    assert d2["config"]["decisions_sha1"] == d1["config"]["decisions_sha1"]
    assert d2["other_configs"]["c4"]["config"]["decisions_sha1"] == d1["other_configs"]["c4"]["config"]["decisions_sha1"]
    assert min(d1["other_configs"]["c4"]["config"]["decisions"]) > 0
    assert len(d2["ranks"]["devices"]) == 2 and d2["ranks"]["devices"][0]["device"] == 0


@pytest.mark.gpu
@pytest.mark.gpuperf
def test_c5_leg_meets_its_slo():
    """wall-clock half of the default line's c5 leg (only under -m gpuperf): p99 below 1 ms, the live step keeps up"""
    p, d = _run(["--gpus", "1", "--workload", "c5"], {}, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    assert d["latency"]["slo_met"] is True and d["live_step"]["kept_up"] is True and d["live_step"]["slo_met"] is True


@pytest.mark.gpu
def test_pool_leg_two_workers_on_one_gpu():
    """`bench.py --pool` with the test hook RB_BENCH_POOL_DEVICES=0,0: two workers (engine + host thread + replica each) on the one
    GPU of the box, every call cut into two slices -- the structure of the line, both workers served reads, outputs equal to a
    single engine's (the N-GPU form of this leg needs an N-GPU node; the pool's replication and slicing are the same code)"""
    p, d = _run(["--gpus", "1", "--pool", "--steps", "2"], {"RB_BENCH_POOL_DEVICES": "0,0", "RB_BENCH_READS_DIVISOR": "10"}, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    for leg in (d, d["other_configs"]["pool_c4"]):
        assert leg["config"]["devices"] == [0, 0] and leg["n_gpus"] == 2 and leg["value"] > 0
        assert leg["parity"]["pool_outputs_equal_single_engine"] is True
        per = leg["pool"]["per_device"]
        assert len(per) == 2 and all(x["reads"] > 0 and x["calls"] >= 2 and 0 < x["busy_share"] <= 1.05 for x in per)
        assert leg["pool"]["replication_seconds"] >= 0 and leg["pool"]["replicated_bytes_per_device"] >= (8 << 30)


@pytest.mark.gpu
def test_pool_legs_in_a_child_of_rank0():
    """On a node with several GPUs rank 0 runs the one-process legs (and the device-to-device pre-flight) in a CHILD process, so
    that a fault in code that has never met distinct devices costs two sub-legs and not the headline.  Forced here on the one
    GPU of the box (RB_BENCH_POOL_CHILD=1) over two workers: the legs come back through the child's line, marked as such."""
    p, d = _run(["--gpus", "1", "--pool", "--steps", "2"],
                {"RB_BENCH_POOL_CHILD": "1", "RB_BENCH_POOL_DEVICES": "0,0", "RB_BENCH_READS_DIVISOR": "10"}, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    for leg in (d, d["other_configs"]["pool_c4"]):
        assert leg["in_child_process"] is True and leg["child_seconds"] > 0
        assert leg["config"]["devices"] == [0, 0] and leg["n_gpus"] == 2 and leg["value"] > 0
        assert leg["parity"]["pool_outputs_equal_single_engine"] is True


def test_pool_child_that_dies_costs_only_its_legs():
    """no GPU here: the child of pool_child cannot even select a device and exits without a line -- the caller gets two legs
    that say so (and a pre-flight record that says so), nothing is raised"""
    sys.path.insert(0, ROOT)
    try:
        import bench
    finally:
        sys.path.pop(0)

    class Ctx:
        world, same_gpu, dev_index = 2, False, 0
    out = bench.pool_child(Ctx(), steps=1, timeout_s=300)
    if out["pool_c3"].get("error") is None:
        pytest.skip("a GPU is present: the child ran")
    assert set(out) == {"pool_c3", "pool_c4", "xgmi_preflight"}
    for k in ("pool_c3", "pool_c4"):
        assert out[k]["value"] == 0.0 and "child" in out[k]["error"]
    assert out["xgmi_preflight"]["ran"] is False
    assert bench.pool_in_child(Ctx()) is True
    Ctx.world = 1
    assert bench.pool_in_child(Ctx()) is False


@pytest.mark.gpu
def test_narrow_leg_carries_a_request_bound():
    """a narrow-filter workload on its own: besides the byte roofline the result carries the request bound (a lower bound on the kernel
    time from two rates probed in the same run and the replayed hit / miss counts) -- under a name that cannot be read as an HBM
    fraction (`request_bound.request_bound_frac`); the structure is checked here, the numbers are bench output"""
    p, d = _run(["--gpus", "1", "--workload", "targets3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-latency"], {}, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "request_roofline" not in d["roofline"]
    q = d["roofline"]["request_bound"]
    assert "error" not in q, q
    assert q["bound"] in ("fabric line requests", "L2 requests")
    for key in ("l2_Grequests_per_s", "fabric_Glines_per_s", "l2_requests_per_read", "fabric_lines_per_read", "fabric_ms_per_launch",
                "l2_ms_per_launch", "model_ms_per_launch", "request_bound_frac", "sum_of_terms_over_kernel_ms", "achieved_fabric_Glines_per_s"):
        assert q[key] > 0, key
    assert q["model_ms_per_launch"] == max(q["fabric_ms_per_launch"], q["l2_ms_per_launch"])
    assert q["l2_Grequests_per_s"] > q["fabric_Glines_per_s"]  # an L2-resident table serves more lines than the fabric
    assert d["roofline"]["plan"][0]["phased"] == 1
    assert "frac" not in q and d["_compact"]["roofline"]["request_bound_frac"] == pytest.approx(q["request_bound_frac"], rel=1e-4)


@pytest.mark.gpu
@pytest.mark.gpuperf
def test_request_bound_is_a_bound():
    """the request bound is meant as a true lower bound of the kernel time: no narrow leg may beat it (an earlier, additive form
    of the model was beaten by 18 % on a 64 MiB one-word table)"""
    for w in ("targets3", "w1_64mib", "readme"):
        p, d = _run(["--gpus", "1", "--workload", w, "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-latency"], {}, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        q = d["roofline"]["request_bound"]
        assert 0.3 < q["request_bound_frac"] <= 1.04, (w, q)

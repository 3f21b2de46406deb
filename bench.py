#!/usr/bin/env python3
"""bench.py -- reads/s of the IBF classify hot path on MI355X (BASELINE.json metric).

A step = one pass of the hot path (K1 count/max for every filter + K2 decision) over one batch of synthetic 360 bp read
prefixes that is already resident in HBM, against IBF(s) resident in HBM.

Headline workload (N = 1 and N > 1 alike) = BASELINE.json configs[2], the configuration the north star's target is stated
on: 10 M reads x 360 bp against the GRCh38-scale depletion IBF (8192 bins, 8 GiB) -- ONE launch of 10 M reads per step and
GPU.  The same line carries `other_configs`: configs[1] (c2, whose 0.41 GB table is partly Infinity-Cache resident),
configs[3] (c4: deplete + target, full check_unblock; full steps, CPU baseline, parity), configs[4] (c5: the 48-flowcell
replay with its latency percentiles, plus a leg through the live step with concatenated undecided chunks), the shape of
the reference's README benchmark (at 250 and at 360 bp) and two narrower shapes of the same filters (three targets alone; one
deplete + one target) -- run in this process (and, for N > 1, by all ranks), after the headline measurement.

With N > 1 every rank (one process per GPU) holds a replica of the IBF and its own shard of reads (weak scaling, no
data-path collective); time = max over ranks, value = all reads / that time.  `python bench.py --gpus N` starts the N ranks
itself (fresh child processes, started before this process touches the GPU); under torchrun (WORLD_SIZE set) it is one of
the ranks.

Prints ONE JSON line on stdout -- the contract fields plus "roofline", "cpu_baseline", "parity" and a one-row summary of every
`other_configs` leg -- bounded to COMPACT_LIMIT bytes (compact_line); everything else (plans, probes, request bounds, pool
statistics, per-rank device records, latency tables) goes to the sidecar file `bench_detail.json` beside this script (and
into gpurun_out/ when that directory exists), or wherever RB_BENCH_DETAIL points.  A run that fails still ends in one
parseable line with an "error" field and a non-zero exit code.
The CPU oracle is used here only as the checker / cpu_baseline leg, never in the timed path.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); what the box of the run delivers on random whole-block gathers is measured live (roofline.read_peak_probe)
METRIC = "reads/sec (360bp prefixes classified vs IBF, unblock/keep decisions)"
# test hook (tests/test_bench_ranks.py only; the line then carries "test_reads_divisor"): every leg's default batch divided by
# this, so that the full default run -- headline + other_configs -- fits a unit test
TEST_DIVISOR = max(1, int(os.environ.get("RB_BENCH_READS_DIVISOR", "1")))


def host_cores():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU box exposes 256
    logical CPUs but grants 16 of them; oversubscribing the quota makes the CPU baseline slower, not faster)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(p)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n


COMPACT_LIMIT = 4096  # bytes of the final stdout line: the driver keeps an 8 KB tail of stdout and stderr together


def _sig(x, digits=6):
    """floats to `digits` significant digits (the line is for reading and for a parser with a small window)"""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None  # strict JSON: no NaN / Infinity tokens
        return float("%.*g" % (digits, x))
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (np.floating,)):
        return _sig(float(x), digits)
    return x


def _cut(text, n):
    return text if not isinstance(text, str) or len(text) <= n else text[: n - 3] + "..."


def leg_summary(r):
    """one row of `other_configs` in the final line: value, HBM fraction, p99, parity -- nothing else"""
    if not isinstance(r, dict):
        return {"error": "no result"}
    out = {}
    if r.get("error"):
        out["error"] = _cut(str(r["error"]), 120)
    if r.get("value") is not None:
        out["value"] = r["value"]
    if r.get("work_skipped"):
        out["work_skipped"] = True  # (the early-decision leg: a product figure, never a roofline figure)
    if r.get("ms_per_step") is not None:
        out["ms_per_step"] = r["ms_per_step"]
    roof = r.get("roofline") or {}
    if roof.get("frac") is not None:
        out["frac"] = roof["frac"]
    if roof.get("frac_of_measured_read_peak") is not None:
        out["frac_of_read_peak"] = roof["frac_of_measured_read_peak"]
    if roof.get("fabric_Glines_per_s") is not None:
        out["Glines_per_s"] = roof["fabric_Glines_per_s"]
    rb = (roof.get("request_bound") or {}).get("request_bound_frac")
    if rb is not None:
        out["request_bound_frac"] = rb
    lat = r.get("latency") or {}
    if lat.get("p99_ms") is not None:
        out["p99_ms"] = lat["p99_ms"]
    live = r.get("live_step") or {}
    if live.get("p99_ms") is not None:
        out["live_p99_ms"] = live["p99_ms"]
    cb = r.get("cpu_baseline") or {}
    if cb.get("value") is not None:
        out["cpu_reads_per_s"] = cb["value"]
    par = r.get("parity")
    if isinstance(par, dict):
        if "decision_mismatches" in par:
            out["parity_ok"] = not (par.get("decision_mismatches") or par.get("raw_max_mismatches"))
            out["checked_reads"] = par.get("checked_reads")
        elif "pool_outputs_equal_single_engine" in par:
            out["parity_ok"] = bool(par["pool_outputs_equal_single_engine"]) and par.get("oracle_mismatches", 0) == 0
            out["checked_reads"] = par.get("checked_reads")
        elif "replayed_decisions_equal_one_batch" in par:
            out["parity_ok"] = bool(par["replayed_decisions_equal_one_batch"])
            out["checked_reads"] = par.get("checked_reads")
    else:
        out["parity_ok"] = None
    return out


def compact_line(result, detail_path=None):
    """The driver-facing line: contract fields, roofline, cpu_baseline, parity and one summary row per other leg, never more than
    COMPACT_LIMIT bytes.  Everything the full result carries beyond that is in the sidecar file."""
    roof = result.get("roofline") or None
    c = {k: result.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                    "vs_baseline", "dtype", "data")}
    cfg = result.get("config") or {}
    c["config"] = {"workload": _cut(cfg.get("workload", ""), 200)}
    for k in ("reads_per_gpu_per_step", "reads_per_call", "read_len", "parallelism", "decisions", "decisions_sha1"):
        if cfg.get(k) is not None:
            c["config"][k] = _cut(cfg[k], 120)
    if isinstance(roof, dict):
        c["roofline"] = {k: roof.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")}
        for k in ("kernel", "avg_kernel_ms", "algorithmic_bytes_per_read", "algorithmic_bytes_per_launch", "frac_of_measured_read_peak",
                  "traffic_frac_of_measured_read_peak", "fabric_Glines_per_s", "basis"):
            if roof.get(k) is not None:
                c["roofline"][k] = _cut(roof[k], 120)
        if (roof.get("read_peak_probe") or {}).get("GBps"):
            c["roofline"]["read_peak_probe_GBps"] = roof["read_peak_probe"]["GBps"]
        if roof.get("traffic_source"):
            c["roofline"]["traffic_source"] = _cut(roof["traffic_source"], 100)
        rb = roof.get("request_bound") or {}
        if rb.get("request_bound_frac") is not None:
            c["roofline"]["request_bound_frac"] = rb["request_bound_frac"]
    else:
        c["roofline"] = None
    cb = result.get("cpu_baseline")
    c["cpu_baseline"] = ({k: _cut(cb.get(k), 160) for k in ("value", "unit", "cores", "kind", "sample")} if isinstance(cb, dict) else None)
    par = result.get("parity")
    c["parity"] = ({k: v for k, v in par.items() if k != "against"} if isinstance(par, dict) else None)
    if isinstance(c["parity"], dict) and "error" in c["parity"]:
        c["parity"]["error"] = _cut(c["parity"]["error"], 100)
    lat = result.get("latency") or {}
    if lat.get("p99_ms") is not None:  # (c5 as the headline)
        c["latency"] = {k: lat[k] for k in ("p50_ms", "p99_ms", "p99.9_ms", "max_ms", "slo_met") if k in lat}
    elif lat.get("by_batch"):
        c["latency"] = {"host_to_host_p99_ms_by_batch": {k: v.get("p99_ms") for k, v in lat["by_batch"].items()}}
    ranks = result.get("ranks") or {}
    if ranks:
        c["ranks"] = {k: ranks.get(k) for k in ("backend", "rccl_ranks", "self_launched", "per_rank_reads_per_s") if k in ranks}
        if ranks.get("devices"):
            c["ranks"]["devices"] = [d.get("device") if isinstance(d, dict) else None for d in ranks["devices"]]
    for k in ("engine", "bin_sharded_reduce_ok", "test_reads_divisor", "error", "bench_seconds"):
        if result.get(k) is not None:
            c[k] = _cut(result[k], 300)
    others = result.get("other_configs")
    if isinstance(others, dict):
        c["other_configs"] = {k: leg_summary(v) for k, v in others.items()}
    if detail_path:
        c["detail"] = detail_path
    c = _sig(c)
    line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    # a hard bound, whatever a future leg adds: shed the optional parts in order of how little the record needs them
    for shed in ("ranks.devices", "latency", "config.decisions", "other_configs.checked_reads", "other_configs.min", "cpu_baseline.sample",
                 "other_configs", "ranks"):
        if len(line) <= COMPACT_LIMIT:
            break
        if shed == "ranks.devices":
            (c.get("ranks") or {}).pop("devices", None)
        elif shed == "config.decisions":
            c["config"].pop("decisions", None)
        elif shed == "other_configs.checked_reads":
            for v in (c.get("other_configs") or {}).values():
                v.pop("checked_reads", None)
                v.pop("ms_per_step", None)
        elif shed == "other_configs.min":
            c["other_configs"] = {k: {kk: v[kk] for kk in ("value", "frac", "p99_ms", "parity_ok", "error") if kk in v}
                                  for k, v in (c.get("other_configs") or {}).items()}
        elif shed == "cpu_baseline.sample":
            if isinstance(c.get("cpu_baseline"), dict):
                c["cpu_baseline"]["sample"] = _cut(c["cpu_baseline"].get("sample"), 40)
        else:
            c.pop(shed, None)
        line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    return line


def detail_paths():
    """where the full result goes: RB_BENCH_DETAIL if set (tests), else bench_detail.json beside this script and a copy under
    gpurun_out/ when that scratch directory exists (gpurun merges it back)"""
    env = os.environ.get("RB_BENCH_DETAIL")
    if env:
        return [env]
    out = [os.path.join(ROOT, "bench_detail.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        out.append(os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    return out


def emit(result):
    """rank 0, once: the sidecar file(s) with everything, then the one bounded line on stdout"""
    written = None
    for path in detail_paths():
        try:
            tmp = path + ".tmp%d" % os.getpid()
            with open(tmp, "w") as f:
                json.dump(_sig(result, 9), f, allow_nan=False)
                f.write("\n")
            os.replace(tmp, path)
            if written is None:
                written = os.path.relpath(path, ROOT) if path.startswith(ROOT + os.sep) else path
        except Exception as ex:  # noqa: BLE001  (a read-only tree must not cost the line)
            print("bench.py: could not write %s: %s" % (path, ex), file=sys.stderr)
    sys.stdout.write(compact_line(result, written) + "\n")
    sys.stdout.flush()


def error_line(msg, world=1, args=None):
    """a run that cannot finish still ends in ONE parseable line (value 0, the reason) -- the caller exits non-zero"""
    r = {"metric": METRIC, "value": 0.0, "unit": "reads/s", "n_gpus": world, "steps": getattr(args, "steps", 0),
         "warmup": getattr(args, "warmup", 0), "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
         "dtype": "u64", "data": "synthetic", "config": {"workload": "failed before a measurement"}, "roofline": None,
         "cpu_baseline": None, "parity": None, "error": _cut(str(msg), 300)}
    emit(r)


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpu_count():
    """GPUs of this node WITHOUT touching the HIP runtime (the launcher must not initialise the GPU before its children
    exist): the KFD topology lists one node per agent, GPUs are the ones with SIMDs; HIP_/ROCR_VISIBLE_DEVICES narrow it.
    None = cannot tell (no KFD here): the children then find out themselves."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(base):
            props = dict(l.split()[:2] for l in open(os.path.join(base, node, "properties")) if len(l.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        return n
    except Exception:
        return None


def finish_without_line(why, world, partial_path):
    """rank 0 ended without its line (a fault inside a leg, a kill, a dead peer): what it had checkpointed -- the headline, measured first,
    and every leg finished since -- still goes out as the ONE line, with the reason under "error"; with no checkpoint, the error line."""
    part = None
    try:
        if partial_path and os.path.exists(partial_path):
            part = json.load(open(partial_path))
    except Exception:  # noqa: BLE001
        part = None
    if isinstance(part, dict) and part.get("value"):
        part["error"] = _cut(str(why), 300) + " -- the line carries what was complete at that point"
        emit(part)
    else:
        error_line(why, world)


def supervise(argv):
    """Rank 0 of every run is a CHILD of this supervisor (started before anything touches the GPU; the supervisor never imports torch
    and never execs).  The child checkpoints its result after the headline and after every further leg (RB_BENCH_PARTIAL); if it dies --
    a GPU fault in a leg, a signal, a launcher tearing the job down because a peer rank died -- the supervisor still leaves ONE parseable
    line on stdout (the checkpoint with an "error", or the bare error line) and exits non-zero.  A clean child is passed through."""
    import signal
    import subprocess
    import tempfile
    fd, partial = tempfile.mkstemp(prefix="rb_bench_partial_", suffix=".json")
    os.close(fd)
    os.unlink(partial)
    env = dict(os.environ, RB_BENCH_SUPERVISED="1", RB_BENCH_PARTIAL=partial)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    child = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE, text=True)
    killed = []

    def on_term(signum, _frame):  # (the supervisor sits in Python waiting for the child: handlers run at once)
        killed.append(signum)
        try:
            child.terminate()
        except Exception:  # noqa: BLE001
            pass
    for sg in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
        try:
            signal.signal(sg, on_term)
        except Exception:  # noqa: BLE001
            pass
    out = []
    try:
        for line in child.stdout:  # (read() would sit in C; iterating returns to the interpreter per line)
            out.append(line.rstrip("\n"))
    except Exception:  # noqa: BLE001
        pass
    try:
        rc = child.wait(timeout=30)
    except Exception:  # noqa: BLE001
        child.kill()
        rc = child.wait()
    lines = [l for l in out if l.startswith("{")]
    for l in out:
        if not l.startswith("{"):
            sys.stderr.write(l + "\n")
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    else:
        why = ("terminated by signal %d (the launcher took the job down: a peer rank died?)" % killed[0]) if killed else \
              ("rank 0 died with signal %d" % -rc if rc < 0 else "rank 0 exited with code %d and no line" % rc)
        finish_without_line(why, world, partial)
        rc = rc or 1
    try:
        if os.path.exists(partial):
            os.unlink(partial)
    except Exception:  # noqa: BLE001
        pass
    sys.exit(rc if rc > 0 else (1 if rc < 0 else 0))


def checkpoint(result):
    """rank 0, supervised: the result so far, where the supervisor finds it if this process does not live to print it"""
    path = os.environ.get("RB_BENCH_PARTIAL")
    if not path:
        return
    try:
        tmp = path + ".tmp"
        with open(tmp, "w") as f:
            json.dump(_sig(result, 9), f, allow_nan=False)
        os.replace(tmp, path)
    except Exception:  # noqa: BLE001
        pass


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks as fresh child processes (rank i -> GPU i, RCCL
    rendezvous on 127.0.0.1) and pass rank 0's JSON line through.  This process never initialises the GPU (no HIP call, no
    torch import: devices are counted from the KFD topology) and it never execs: it waits for the children and exits with
    the worst of their codes.  The reference's scaling model is N classification workers behind one queue
    (src/main/adaptive_sampling.hpp:745-751); here a worker is a GPU."""
    import subprocess
    if os.environ.get("RB_BENCH_SAME_GPU") != "1" and os.environ.get("RB_BENCH_ENGINE") != "none":
        have = visible_gpu_count()
        if have is not None and have < n:
            print("bench.py --gpus %d: only %d GPU(s) visible on this node" % (n, have), file=sys.stderr)
            sys.exit(2)
    port = _free_port()
    import tempfile
    fd, partial = tempfile.mkstemp(prefix="rb_bench_partial_", suffix=".json")
    os.close(fd)
    os.unlink(partial)
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   RB_BENCH_SELF_LAUNCHED="1", RB_BENCH_SUPERVISED="1")  # (this launcher is the supervisor of its ranks)
        if r == 0:
            env["RB_BENCH_PARTIAL"] = partial
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = None
    while True:
        alive = [p for p in procs if p.poll() is None]
        if not alive:
            break
        failed = any(p.returncode for p in procs if p.returncode is not None)
        if failed and deadline is None:
            deadline = time.time() + 15.0  # a rank died: the others would wait in a collective for ever
        if deadline is not None and time.time() > deadline:
            for p in alive:
                p.terminate()  # exact children of this process, by handle
            deadline = float("inf")
        time.sleep(0.2)
    reader.join(10)
    lines = "".join(out0).splitlines()
    for line in lines:  # the contract is ONE JSON line on stdout; library chatter goes to stderr
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    rc = 0
    for p in procs:
        rc = rc or (p.returncode or 0)
    if not any(l.startswith("{") for l in lines):
        # rank 0 never got to its line (it died, or a dead rank left it in a collective until it was terminated)
        finish_without_line("no line from rank 0; exit codes of the ranks: %s" % [p.returncode for p in procs], n, partial)
        rc = rc or 1
    try:
        if os.path.exists(partial):
            os.unlink(partial)
    except Exception:  # noqa: BLE001
        pass
    sys.exit(rc if rc >= 0 else 1)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="",
                    help="default: c3 (BASELINE configs[2], 10 M reads per step) + other_configs; or one of c2, c3, c3np2, c1, c4, c5, "
                         "readme, grch38_f100k, zymo, zymo16 on its own")
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU per step (default: the config's batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--check-reads", type=int, default=2048, help="N>1: reads of rank 0 checked against the oracle")
    ap.add_argument("--bin-sharded", action="store_true",
                    help="N>1: every rank holds a word-column slice of each block and sees ALL reads; per-read partial "
                         "maxima are all-gathered and the max is taken in the decision kernel (strong scaling)")
    ap.add_argument("--no-overlap", action="store_true", help="serialise the count kernels of different filters")
    ap.add_argument("--rate", type=float, default=150000.0, help="c5: total chunk arrival rate (chunks/s) over all GPUs")
    ap.add_argument("--replay-seconds", type=float, default=2.0, help="c5: length of the replayed arrival process")
    ap.add_argument("--read-len", type=int, default=0, help="override the read length of the workload (e.g. 1500: 16 counter planes)")
    ap.add_argument("--phased", default="", help="tuning: 'off' or 'min_mib,max_mib,base_ticks,ticks_per_mib' for the clock-phased gathers of narrow filters")
    ap.add_argument("--serial-table-mib", type=int, default=-1,
                    help="tuning: filters up to this size take turns instead of overlapping (default: the engine's 64; 0 = round 1 behaviour)")
    ap.add_argument("--no-extras", action="store_true", help="default run: leave `other_configs` out")
    ap.add_argument("--pool", action="store_true",
                    help="only the single-process multi-GPU legs: c3 and c4 through rb_pool_classify_batch from ONE host process "
                         "(rank 0 drives every GPU of the job; N = 1: the one GPU) from page-locked host memory")
    return ap.parse_args()


def load_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return {}


def host_wait_for_rank0(dist, rank, tag):
    """The other ranks wait on the HOST for rank 0 (a key in the job's rendezvous store) -- not in a collective: an RCCL barrier
    is a kernel that spins on the waiting ranks' GPUs, and rank 0 is about to measure on those very GPUs (run_pool).
    True when the store was used."""
    if dist is None:
        return False
    try:
        store = dist.distributed_c10d._get_default_store()
        if rank == 0:
            store.set("rb_bench_" + tag, "1")
        else:
            import datetime
            store.wait(["rb_bench_" + tag], datetime.timedelta(seconds=1500))  # (longer than any leg rank 0 runs alone, child included)
        return True
    except Exception:  # noqa: BLE001  (no store to be had: the caller's collective does the waiting)
        return False


class Ctx:
    """what every leg of the run shares: the process group, the device, the filters already resident in HBM"""

    def __init__(self, args, torch, dist, world, rank, dev_index, backend, same_gpu, force_group):
        self.args, self.torch, self.dist = args, torch, dist
        self.world, self.rank, self.dev_index, self.backend = world, rank, dev_index, backend
        self.same_gpu, self.force_group = same_gpu, force_group
        self.dev = torch.device("cuda", dev_index)
        self.red_dev = self.dev if backend == "nccl" else torch.device("cpu")  # where the tensors of the collectives live
        self.filters = {}  # workload key -> (DeviceIBF, planted reference)
        self.views = {}    # workload key -> (host image, oracle view of it)
        self.filter_setup_s = 0.0
        self.create_s = {}  # key -> seconds of its build_device_filter (allocation with the placement trial, synthetic fill, planted inserts)

    SEEDS = {"c2": (2, 20), "c3": (4, 40), "c3np2": (4, 40), "c1": (1, 10), "zymo": (6, 60), "grch38_f100k": (8, 80),
             "zymo16": (6, 60), "mock_deplete": (11, 110), "mock_t1": (12, 111), "mock_t2": (13, 112), "mock_t3": (14, 113), "w1_64mib": (15, 114)}

    def filter(self, key):
        from readbouncer_amd import synth
        if key not in self.filters:
            t = time.time()
            fs, ps = self.SEEDS[key]
            seg = 512 if key.startswith("mock_") or key.startswith("w1_") else 2048
            self.filters[key] = synth.build_device_filter(self.dev_index, synth.WORKLOADS[key], fill_seed=fs, plant_seed=ps,
                                                          n_segments=seg)
            self.torch.cuda.synchronize()
            self.filter_setup_s += time.time() - t
            self.create_s[key] = round(time.time() - t, 3)
        return self.filters[key]

    def oracle_view(self, key):
        """the checker's view of a resident filter (rank 0's parity / cpu_baseline legs only): downloaded from HBM once per run and
        kept while the filter is -- config 3's table is 8 GiB and four legs check against it"""
        from oracle import pyoracle as po
        if key not in self.views:
            h = self.filter(key)[0].download()
            self.views[key] = (h, po.OracleIBF.wrap(h.info["n_bins"], h.info["n_hash"], h.info["kmer_size"], h.info["n_bits"], h.words()))
        return self.views[key][1]

    def release(self, keys):
        for k in keys:
            self.views.pop(k, None)
            f = self.filters.pop(k, None)
            if f is not None:
                try:
                    f[0].free()
                except Exception:  # noqa: BLE001
                    pass

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def wait_for_rank0(self, tag):
        host_wait_for_rank0(self.dist, self.rank, tag)
        self.barrier()  # (all ranks are here within microseconds of each other now)

    def max_over_ranks(self, values):
        """element-wise max of a list of floats over the ranks (the contract's max-over-ranks timing)"""
        if self.dist is None:
            return list(values)
        t = self.torch.tensor(list(values), dtype=self.torch.float64, device=self.red_dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(x) for x in t.tolist()]

    def gather_floats(self, value):
        if self.dist is None:
            return [value]
        g = self.torch.zeros(self.world, dtype=self.torch.float64, device=self.red_dev)
        self.dist.all_gather_into_tensor(g, self.torch.tensor([value], dtype=self.torch.float64, device=self.red_dev))
        return [float(x) for x in g.tolist()]


def workload_spec(ctx, name):
    """-> (deplete keys, target keys, text, default reads per step and GPU, read length)"""
    from readbouncer_amd import synth
    if name == "c4":
        return (["c3"], ["zymo"], "config4: deplete=GRCh38-scale IBF (8192 bins, 8 GiB) + target=Zymo-mock-like IBF (600 bins), "
                "check_unblock", 2_000_000, 360)
    if name == "readme":
        # the reference's own (only) published benchmark shape: README.md:254-262, 250 bp prefixes, 1 deplete + 3 targets
        return (["mock_deplete"], ["mock_t1", "mock_t2", "mock_t3"],
                "README benchmark shape: 250bp prefixes vs 1 deplete (122 bins) + 3 target (43/29/49 bins) IBFs, k=13, F=100000, "
                "check_unblock", 1_000_000, 250)
    if name == "targets3":
        # target-only adaptive sampling with three bacterial targets: three one-word filters of one hash geometry, which the engine
        # serves from ONE three-word table through the one-lane-per-block build of the phased kernel
        return ([], ["mock_t1", "mock_t2", "mock_t3"],
                "three target IBFs (43/29/49 bins, 10.4 MB each), k=13, F=100000, 250bp prefixes, target-only check_unblock", 1_000_000, 250)
    if name == "deplete_target":
        return (["mock_t3"], ["mock_t1"], "one deplete (49 bins) + one target (43 bins) IBF, k=13, F=100000, 250bp prefixes, check_unblock",
                1_000_000, 250)
    w = synth.WORKLOADS[name]
    return [name], [], w["name"], w["reads"], w["read_len"]


def run_throughput(ctx, name, n_reads=0, read_len=0, steps=None, warmup=None, cpu_seconds=None, latency=False,
                   bin_sharded=False):
    """One throughput measurement as the contract describes it: W untimed steps, then K steps between barriers, time = max
    over ranks.  Returns the result dict on rank 0 (None elsewhere): value, roofline (live hipEvent kernel time), parity
    against the oracle, CPU baseline (N = 1)."""
    from readbouncer_amd import capi, synth
    args, torch, dist = ctx.args, ctx.torch, ctx.dist
    world, rank, dev, dev_index = ctx.world, ctx.rank, ctx.dev, ctx.dev_index
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    cpu_seconds = args.cpu_seconds if cpu_seconds is None else cpu_seconds
    t_setup = time.time()
    dep_keys, tgt_keys, wname, default_reads, default_len = workload_spec(ctx, name)
    deplete = [ctx.filter(k)[0] for k in dep_keys]
    target = [ctx.filter(k)[0] for k in tgt_keys]
    ref = np.concatenate([ctx.filter(k)[1] for k in dep_keys + tgt_keys])
    if not n_reads and TEST_DIVISOR > 1:
        n_reads = max(4096, default_reads // TEST_DIVISOR)
    n_reads = n_reads or default_reads
    read_len = read_len or default_len
    if n_reads != default_reads:
        wname += " [%d reads per step and GPU]" % n_reads
    elif name in ("c3", "c3np2"):
        wname += " [one launch of %d reads per step and GPU]" % n_reads
    filters = deplete + target
    nf = len(filters)
    geo = [(f.info["n_bins"], f.info["kmer_size"], f.info["n_hash"]) for f in filters]
    bytes_per_read = synth.algorithmic_bytes_per_read(read_len, geo)

    # reads are generated on the device (plumbing) and stay resident in HBM
    seed = 1000 if bin_sharded else 1000 + rank  # bin-sharded: every rank classifies the SAME reads against its columns
    t_seq, t_off, t_len = synth.make_reads_device(seed, n_reads, read_len, ref, dev)
    lens = np.full(n_reads, read_len, dtype=np.uint32)
    offs = np.arange(n_reads, dtype=np.uint64) * np.uint64(read_len)
    t_max = torch.zeros((n_reads, nf), dtype=torch.int16, device=dev)
    t_best = torch.zeros(n_reads, dtype=torch.int32, device=dev)
    t_dec = torch.zeros(n_reads, dtype=torch.uint8, device=dev)
    t_st = torch.zeros(n_reads, dtype=torch.uint8, device=dev)
    eng = capi.Engine(dev_index, deplete, target)
    if args.no_overlap:
        eng.set_overlap(False)
    if args.serial_table_mib >= 0:
        eng.set_serial_table_bytes(args.serial_table_mib << 20)
    if args.phased == "off":
        eng.set_phased(0, 0, 0, 0, 0)
    elif args.phased:
        lo, hi, base, tk = [int(x) for x in args.phased.split(",")]
        eng.set_phased(lo << 20, hi << 20, base, tk, 1024)  # "0,0,300,3": plain gathers, but the short-read kernel for one-word filters
    # a dedicated non-null stream: steps are queued asynchronously; torch.cuda.synchronize() covers it
    side = torch.cuda.Stream(device=dev)
    stream = side.cuda_stream
    max_len = int(read_len)
    if bin_sharded:
        eng.set_column_shard(rank, world)
        # the u16 partial maxima of all ranks, all-gathered as they are (byte view: an all-gather does no arithmetic, so
        # there is no widening for the collective and half the bytes of an int32 all-reduce); the max over the ranks is
        # taken inside the decision kernel
        t_all = torch.zeros((world * n_reads, nf * 2), dtype=torch.uint8, device=ctx.red_dev)  # rank-major

    def step():
        if not bin_sharded:
            eng.classify_device(t_seq.data_ptr(), t_off.data_ptr(), t_len.data_ptr(), n_reads, max_len, 0.1, 0.95,
                                capi.RB_MODE_CHECK_UNBLOCK, t_max.data_ptr(), t_best.data_ptr(), t_dec.data_ptr(),
                                t_st.data_ptr(), stream)
            return
        # partial maxima of this rank's columns -> all_gather over xGMI -> decision over the gathered tables.
        # Everything is ordered on `side` (the collective is enqueued with `side` current: RCCL's own stream waits for
        # it and `side` waits for the collective); the host never waits inside a step.
        eng.classify_device(t_seq.data_ptr(), t_off.data_ptr(), t_len.data_ptr(), n_reads, max_len, 0.1, 0.95,
                            capi.RB_MODE_CHECK_UNBLOCK, t_max.data_ptr(), None, None, None, stream)
        with torch.cuda.stream(side):
            if ctx.backend == "nccl":
                dist.all_gather_into_tensor(t_all, t_max.view(torch.uint8))
                parts = t_all
            else:  # gloo test hook: through the host
                side.synchronize()
                dist.all_gather_into_tensor(t_all, t_max.view(torch.uint8).cpu())
                parts = t_all.to(dev, non_blocking=False)
        eng.decide_device_parts(parts.data_ptr(), world, n_reads * nf, t_len.data_ptr(), n_reads, max_len, 0.1, 0.95,
                                capi.RB_MODE_CHECK_UNBLOCK, t_best.data_ptr(), t_dec.data_ptr(), t_st.data_ptr(), stream)
        if ctx.backend != "nccl":
            side.synchronize()  # `parts` is a temporary of this step

    # the inputs were produced on torch's default stream; the steps run on `side` (non-blocking): order them
    torch.cuda.synchronize()
    setup_s = time.time() - t_setup
    # ---------------------------------------------------------------- warm-up + timed region
    for _ in range(warmup):
        step()
    ctx.barrier()
    eng.set_timing(True)  # hipEvent pairs around the count kernels, on the launch stream, no sync
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0  # this rank's own time for its K steps
    ctx.barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, n_calls = eng.kernel_time()
    eng.set_timing(False)
    elapsed = ctx.max_over_ranks([elapsed])[0]
    per_rank_s = ctx.gather_floats(t_local)
    total_reads = n_reads * (1 if bin_sharded else world) * steps
    value = total_reads / elapsed

    result = None
    decisions = None
    if rank == 0:
        avg_kernel_s = (kernel_ms / max(1, n_calls)) / 1e3
        achieved = bytes_per_read * n_reads / avg_kernel_s / 1e9
        if bin_sharded:
            achieved /= world  # every rank gathers its share of the word columns of every block
        # fabric-side traffic of one launch: NOT measured in this run -- rocprofv3 --pmc passes of an earlier run of the
        # same workload (profiles/collect_pmc.sh), kept in profiles/traffic.json and replayed here per read
        traffic, traffic_source = None, None
        tj = load_json("traffic.json").get("readme360" if (name == "readme" and read_len == 360) else name, {})
        if tj.get("hbm_bytes_per_read") and not bin_sharded and (read_len == default_len or (name == "readme" and read_len == 360)):
            traffic = tj["hbm_bytes_per_read"] * n_reads
            traffic_source = "replayed: profiles/traffic.json (%s), per read x reads per launch" % tj.get("source", "rocprofv3 --pmc, separate passes")
        decisions = t_dec.cpu().numpy()
        # read peak of THIS device for the dominant filter's access pattern, measured now: random whole-block gathers from the
        # same table with no compute attached (rb_dibf_probe_read_peak, rb_probe.hip), 12 and 24 loads in flight per wave
        probe = None
        if not bin_sharded and not os.environ.get("RB_BENCH_NO_PROBE"):
            dom = max(filters, key=lambda f: f.info["bin_width"])
            bb = dom.device_stride() * 8
            row = 4096 if bb >= 3072 else 1024 if bb >= 1024 else 128 if bb >= 128 else 0
            if row:
                try:
                    ntl = dom.info["n_blocks"] * bb > (512 << 20)  # the engine's own rule (rb_engine_set_nt_threshold)
                    runs = [(dom.probe_read_peak(row, ntl, lif, target_ms=150.0), lif) for lif in (12, 24)]
                    (gb, pms), lif = max(runs)
                    probe = {"GBps": gb, "row_bytes": row, "block_bytes": bb, "nontemporal": bool(ntl), "loads_in_flight_per_wave": lif,
                             "run_ms": pms, "table_bytes": dom.info["n_blocks"] * bb,
                             "source": "rb_dibf_probe_read_peak: this run, this device, this filter's table; no compute attached"}
                except Exception as ex:  # noqa: BLE001  (a measurement aid never fails the bench)
                    probe = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:160])}

        # which form of K1 the engine launched for these filters on this batch, as the engine itself reports it (rb_engine_plan):
        # kernel, and for the clock-phased form the row of the planner's table with its slice size and window length
        plans = []
        if not bin_sharded:
            for fi in range(len(filters)):
                try:
                    pl = eng.plan(fi, n_reads, read_len)
                    plans.append({k: pl[k] for k in ("kernel", "table_bytes", "merged_members", "phased", "phase_shape_name", "phase_slice_log2",
                                                     "phase_slices", "phase_slice_bytes", "phase_window_ticks", "column_slices", "nontemporal") if pl[k] not in (0, "")})
                except Exception as ex:  # noqa: BLE001
                    plans.append({"error": str(ex)[:120]})
        forms = {pl.get("kernel", "?") for pl in plans} or {"ibf_count_max_kernel"}
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "kernel": " + ".join(sorted(forms)), "plan": plans, "avg_kernel_ms": avg_kernel_s * 1e3,
                "algorithmic_bytes_per_read": bytes_per_read,
                "algorithmic_bytes_per_launch": bytes_per_read * n_reads}
        if traffic:
            # the fabric serves LINES: what every wide shape runs into is the chip's rate of 128-byte line requests (c3, c3np2, c4, c2 and the
            # reference-default GRCh38 filter all sit at 54-55.5 G lines/s, profiles/r05/line_rate_invariant.md), so a shape whose blocks
            # do not fill their lines (c4's 600-bin target: 80 of 128 bytes; W = 485: 3 880 of 3 968) shows a lower ALGORITHMIC fraction at
            # the same line rate
            roof["fabric_Glines_per_s"] = traffic / 128.0 / avg_kernel_s / 1e9
        if probe:
            roof["read_peak_probe"] = probe
            if probe.get("GBps"):
                roof["frac_of_measured_read_peak"] = achieved / probe["GBps"]
                if traffic:
                    roof["traffic_frac_of_measured_read_peak"] = traffic / avg_kernel_s / 1e9 / probe["GBps"]
        # Narrow filters (blocks of less than a cache line, tables of a few L2 sizes) are bound by REQUESTS, not bytes: every lookup is
        # one request to an XCD's L2, and every miss one 128-byte line request to the fabric.  Their roofline is
        #   t >= max( misses / (fabric line rate),  (hits + misses) / (L2 request rate) ),
        # a true lower bound: the no-compute probe on random single lines (profiles/r04/line_rate_probe*.txt) shows the fabric
        # serving ~57-61 G lines/s whatever share of the requests hits the L2, hits riding along for free until the L2's own
        # ~250 G requests/s (an ADDITIVE model -- hits / L2 rate + misses / fabric rate -- overstates the time: a 64 MiB one-word
        # table ran 18 % faster than it allows).  Both rates are measured now (a 4 MiB table = L2 resident; a scratch table of the
        # kernel's table size, taking the L2's share of it as hits); hits and misses per read are replayed from the counter passes
        # of profiles/traffic.json (TCC_HIT / TCC_MISS = TCC_EA0_RDREQ).
        tj_req = load_json("traffic.json").get("readme360" if (name == "readme" and read_len == 360) else name, {}) \
            if (read_len == default_len or (name == "readme" and read_len == 360)) else {}
        narrow = bool(plans) and all(pl.get("phased") or "merged" in pl.get("kernel", "") for pl in plans)
        if narrow and tj_req.get("TCC_EA0_RDREQ") and tj_req.get("l2_hit_rate") is not None and not bin_sharded \
                and not os.environ.get("RB_BENCH_NO_PROBE"):
            try:
                ktab = max(int(pl.get("table_bytes", 0)) for pl in plans)
                scratch = capi.DeviceIBF.create(ctx.dev_index, 256, 3, 13, 256 * max(1 << 17, ktab // 32))
                scratch.fill_synth(7)
                torch.cuda.synchronize()
                l2_bytes = 4 << 20
                r_l2 = max(scratch.probe_read_peak(128, 0, 24, l2_bytes, 60.0)[0] for _ in range(2)) / 128.0
                r_full = max(scratch.probe_read_peak(128, 0, 24, 0, 100.0)[0] for _ in range(2)) / 128.0
                scratch.free()
                share = min(1.0, l2_bytes / float(ktab))
                r_fabric = (1.0 - share) * r_full if share < 1.0 else r_full  # the probe's misses per second on a table of this size
                miss_pr = tj_req["TCC_EA0_RDREQ"] / tj_req["reads_per_launch"]
                hr = tj_req["l2_hit_rate"]
                hit_pr = miss_pr * hr / max(1e-9, 1.0 - hr)
                fabric_ms = n_reads * miss_pr / r_fabric / 1e6  # (rates in G lines/s)
                l2_ms = n_reads * (hit_pr + miss_pr) / r_l2 / 1e6
                model_ms = max(fabric_ms, l2_ms)
                roof["request_bound"] = {
                    "what": "lower bound on the kernel time from REQUEST rates (L2 requests, fabric line requests), not a byte fraction: "
                            "request_bound_frac = bound / kernel time",
                    "bound": "fabric line requests" if fabric_ms >= l2_ms else "L2 requests",
                    "l2_Grequests_per_s": r_l2, "fabric_Glines_per_s": r_fabric,
                    "probe_table_bytes": ktab, "probe_full_table_Glines_per_s": r_full,
                    "l2_requests_per_read": hit_pr + miss_pr, "l2_hits_per_read": hit_pr, "fabric_lines_per_read": miss_pr,
                    "fabric_ms_per_launch": fabric_ms, "l2_ms_per_launch": l2_ms, "model_ms_per_launch": model_ms,
                    "request_bound_frac": model_ms / (avg_kernel_s * 1e3),
                    # where the clock-phased kernels really are: a window starts on a cold slice (fabric bound) and ends on a warm one
                    # (L2 bound), and the two parts overlap little -- the kernel time is within 10 % of the SUM of the two terms
                    "sum_of_terms_over_kernel_ms": (fabric_ms + l2_ms) / (avg_kernel_s * 1e3),
                    "achieved_fabric_Glines_per_s": n_reads * miss_pr / avg_kernel_s / 1e9,
                    "source": "rates: rb_dibf_probe_read_peak with 128-byte rows, this run (4 MiB table; a scratch table of the kernel's "
                              "table size, its misses = the share beyond the L2); hits / misses per read: " + str(tj_req.get("source", "profiles/traffic.json"))}
            except Exception as ex:  # noqa: BLE001  (a measurement aid never fails the bench)
                roof["request_bound"] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:160])}
        try:  # tables of >= 1 GiB are placed by trial (rb_set_placement_tries): allocations probed, GB/s of the kept and of the slowest one,
            # and what the trial cost this filter: seconds probing, seconds waited afterwards, the most HBM its candidates held at once
            # (create_s = the whole rb_dibf_create incl. the trial: without it the load would have taken create_s - trial_s - settle_s)
            roof["placement"] = []
            for kk, f in zip(dep_keys + tgt_keys, filters):
                rec = dict(zip(("tries", "kept_GBps", "slowest_GBps"), f.placement()))
                rec.update(f.placement_cost())
                rec["create_s"] = ctx.create_s.get(kk)
                rec["table_bytes"] = int(f.info["n_words"] * 8)
                roof["placement"].append(rec)
        except Exception:  # noqa: BLE001
            pass
        table_bytes = sum(f.info["n_words"] * 8 for f in filters)
        if table_bytes < (256 << 20) * 4:
            roof["note"] = ("table of %.2f GB against a 256 MiB Infinity Cache: part of the gathers are served on-die; "
                            "`traffic` counts L2->fabric requests, Infinity-Cache hits included, so this is a fabric "
                            "figure -- the HBM-bound case is config 3 (the headline)" % (table_bytes / 1e9))
        result = {
            "metric": METRIC, "value": value, "unit": "reads/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if bin_sharded else "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": wname, "reads_per_gpu_per_step": n_reads, "launches_per_step": 1, "read_len": read_len,
                       "filters": [{"n_bins": g[0], "k": g[1], "h": g[2], "bytes": f.info["n_words"] * 8}
                                   for g, f in zip(geo, filters)],
                       "parallelism": ("bin-sharded x%d, all_gather of u16 partial maxima, max taken in the decision "
                                       "kernel" % world) if bin_sharded
                       else "read-sharded x%d, IBF replicated" % world,
                       "decisions": np.bincount(decisions, minlength=3).tolist()},
            "per_rank_reads_per_s": [n_reads * steps / x for x in per_rank_s],
            "roofline": roof,
            "setup_s": setup_s,
        }
        if os.environ.get("RB_BENCH_DUMP_DECISIONS"):  # tests compare the N-rank decisions with the 1-rank run
            import hashlib
            result["config"]["decisions_sha1"] = hashlib.sha1(decisions.tobytes()).hexdigest()

    # ---------------------------------------------------------------- parity check + CPU baseline (rank 0)
    buf = None
    if rank == 0:
        cap = min(n_reads, 1 << 21)  # host copy of the head of the batch: CPU baseline, parity, latency legs
        buf = t_seq[: cap * read_len].cpu().numpy()
    if rank == 0 and not args.no_cpu_baseline and cpu_seconds > 0:
        from oracle import pyoracle as po
        views = [ctx.oracle_view(k) for k in dep_keys + tgt_keys]
        od, ot = views[:len(deplete)], views[len(deplete):]
        cores = host_cores()
        cap = len(buf) // read_len
        if world == 1 and not ctx.force_group:
            # the CPU baseline proper (N = 1 only): a bounded sample of the same batch, all granted cores
            pilot = min(cap, 64 * min(cores, 64))
            tp = time.perf_counter()
            po.batch_check_unblock(od, ot, buf, offs[:pilot], lens[:pilot], n_threads=cores)
            pilot_s = time.perf_counter() - tp
            sample = int(min(cap, max(pilot, pilot * cpu_seconds / max(pilot_s, 1e-6))))
            tp = time.perf_counter()
            cpu_dec, cpu_st = po.batch_check_unblock(od, ot, buf, offs[:sample], lens[:sample], n_threads=cores)
            cpu_s = time.perf_counter() - tp
            t1 = time.perf_counter()
            n1 = min(sample, max(16, int(sample / cores / 4)))
            po.batch_check_unblock(od, ot, buf, offs[:n1], lens[:n1], n_threads=1)
            one_s = time.perf_counter() - t1
            result["cpu_baseline"] = {"value": sample / cpu_s, "unit": "reads/s", "cores": cores, "kind": "port",
                                      "sample": "first %d reads of the same batch, oracle check_unblock, read-parallel "
                                                "pthreads; single-thread rate %.1f reads/s on %d reads"
                                                % (sample, n1 / one_s, n1),
                                      "single_thread_reads_per_s": n1 / one_s}
        else:
            # N > 1: no CPU baseline (the contract times it at N = 1), but rank 0's decisions are still checked
            sample = min(cap, max(1, args.check_reads))
            cpu_dec, cpu_st = po.batch_check_unblock(od, ot, buf, offs[:sample], lens[:sample], n_threads=cores)
            result["cpu_baseline"] = None
        mism = int((decisions[:sample] != cpu_dec).sum())
        # bit level, not only decisions: the raw maxima K1 wrote for the timed batch (every filter's column of t_max) against
        # the oracle's max over bins and strands of the same reads; `near_threshold_reads` = reads of the sample whose
        # maximum lies within +-5 of the threshold of their length (synth's threshold-adjacent stratum), i.e. the reads on
        # which a count that is off by a few WOULD flip the decision
        gpu_max = t_max[:sample].cpu().numpy().view(np.uint16)
        cpu_max = np.stack([po.batch_raw_max(v, buf, offs[:sample], lens[:sample], cores) for v in views], axis=1)
        max_mism = int((gpu_max != cpu_max).any(axis=1).sum())
        thr = np.array([po.threshold(read_len, int(f.info["kmer_size"]), 0.1, 0.95) for f in filters], dtype=np.int64)
        near = int((np.abs(cpu_max.astype(np.int64) - thr[None, :]) <= 5).any(axis=1).sum())
        result["parity"] = {"checked_reads": sample, "decision_mismatches": mism, "raw_max_mismatches": max_mism,
                            "near_threshold_reads": near,
                            "against": "oracle check_unblock AND oracle raw maxima (every filter) on rank 0's first reads of "
                                       "the timed batch"}
        if mism or max_mism:
            result["parity"]["error"] = "GPU decisions or raw maxima differ from the oracle"
        del views
    elif rank == 0:
        result["cpu_baseline"] = None

    # ---------------------------------------------------------------- per-read classify latency (small batches)
    if rank == 0 and latency and not args.no_latency and not bin_sharded:
        lat = {}
        for mb in (64, 256, 1024):
            m = min(mb, len(buf) // read_len)
            sub = np.ascontiguousarray(buf[: m * read_len])
            so, sl = offs[:m].copy(), lens[:m].copy()
            for _ in range(5):
                eng.classify(sub, so, sl)
            ts = []
            for _ in range(200):
                a = time.perf_counter()
                eng.classify(sub, so, sl)  # host buffers in, decisions back on the host
                ts.append((time.perf_counter() - a) * 1e3)
            ts = np.sort(np.array(ts))
            lat[str(mb)] = {"p50_ms": float(ts[len(ts) // 2]), "p99_ms": float(ts[int(len(ts) * 0.99) - 1]),
                            "reads_per_s": m / (float(ts[len(ts) // 2]) / 1e3)}
        # PCIe-inclusive throughput of one large host-side batch (never `value`)
        m = min(len(buf) // read_len, 1 << 20)
        sub = np.ascontiguousarray(buf[: m * read_len])
        so, sl = offs[:m].copy(), lens[:m].copy()
        eng.classify(sub, so, sl)
        a = time.perf_counter()
        for _ in range(3):
            eng.classify(sub, so, sl)
        big_s = (time.perf_counter() - a) / 3
        result["latency"] = {"what": "host-to-host rb_classify_batch wall time per micro-batch (H2D + kernels + D2H)",
                             "by_batch": lat,
                             "pcie_inclusive": {"batch_reads": m, "ms": big_s * 1e3, "reads_per_s": m / big_s,
                                                "note": "pageable host buffers in, all outputs back"}}
    ctx.barrier()
    eng.destroy()
    del t_seq, t_off, t_len, t_max, t_best, t_dec, t_st
    torch.cuda.empty_cache()
    return result


def run_pool(ctx, name, steps=3):
    """The single-process multi-GPU form (rb_pool.cpp): ONE host process -- rank 0 -- drives every GPU of the job
    through rb_pool_classify_batch, which is what a MinKNOW front-end or the CLI's --devices would do; the other ranks of a
    --gpus N run sit at the barrier.  Filters are replicated device to device from rank 0's resident copies
    (rb_pool_create_from_device: xGMI between peers); reads, offsets, lengths and every output live in page-locked host memory;
    a call carries 1 M reads per device and is cut into one contiguous slice per device.  PCIe-inclusive by construction, so
    never the headline `value`.  Reports reads/s, replication seconds, and per device the share of the wall time its engine
    was busy."""
    from readbouncer_amd import capi, synth
    torch, world, rank = ctx.torch, ctx.world, ctx.rank
    def measure():
        env = os.environ.get("RB_BENCH_POOL_DEVICES")  # test hook: "0,0" = two workers on the one GPU of the box
        devices = [int(x) for x in env.split(",")] if env else (list(range(world)) if (world > 1 and not ctx.same_gpu) else [ctx.dev_index])
        dep_keys, tgt_keys, wname, _, L = workload_spec(ctx, name)
        deplete = [ctx.filter(k)[0] for k in dep_keys]
        target = [ctx.filter(k)[0] for k in tgt_keys]
        ref = np.concatenate([ctx.filter(k)[1] for k in dep_keys + tgt_keys])
        nf = len(deplete) + len(target)
        per_dev = max(8192, 1_000_000 // TEST_DIVISOR)
        n = per_dev * len(devices)
        t0 = time.time()
        pool = capi.Pool.from_device(devices, deplete, target)
        create_s = time.time() - t0
        torch.cuda.set_device(ctx.dev_index)  # (creating replicas on other devices moved this thread's current device)
        t_seq, _, _ = synth.make_reads_device(2000, n, L, ref, ctx.dev)
        blocks = {"seq": capi.HostBlock(n * L, np.uint8), "off": capi.HostBlock(n, np.uint64), "len": capi.HostBlock(n, np.uint32),
                  "max": capi.HostBlock(n * nf, np.uint16), "best": capi.HostBlock(n, np.int32), "dec": capi.HostBlock(n, np.uint8),
                  "st": capi.HostBlock(n, np.uint8)}
        blocks["seq"].array[:] = t_seq.cpu().numpy()
        blocks["off"].array[:] = np.arange(n, dtype=np.uint64) * np.uint64(L)
        blocks["len"].array[:] = L
        del t_seq

        def call():
            pool.classify_into(blocks["seq"].ptr, blocks["off"].ptr, blocks["len"].ptr, n, blocks["max"].ptr, blocks["best"].ptr,
                               blocks["dec"].ptr, blocks["st"].ptr)
        call()  # engines' staging buffers, threshold tables, code objects on every device
        pool.stats(reset=True)
        pool.set_timing(True)  # hipEvent pairs around every engine's count kernels (rb_engine_set_timing per worker)
        t0 = time.perf_counter()
        for _ in range(steps):
            call()
        wall = time.perf_counter() - t0
        st = pool.stats()
        ktimes = pool.kernel_time()
        pool.set_timing(False)
        # parity: the pool's outputs for the first reads against ONE engine on this rank's device (same resident filters)
        m = min(n, 200_000)
        eng = capi.Engine(ctx.dev_index, deplete, target)
        mc, _, dec, status = eng.classify(blocks["seq"].array[: m * L], blocks["off"].array[:m].copy(), blocks["len"].array[:m].copy())
        eng.destroy()
        equal = bool(np.array_equal(mc.reshape(-1), blocks["max"].array[: m * nf]) and np.array_equal(dec, blocks["dec"].array[:m])
                     and np.array_equal(status, blocks["st"].array[:m]))
        # ... and the LAST device's slice against the same engine (a replica that travelled device to device)
        lo = n - min(per_dev, 50_000)
        eng = capi.Engine(ctx.dev_index, deplete, target)
        mc2, _, dec2, _ = eng.classify(blocks["seq"].array[lo * L:], np.arange(n - lo, dtype=np.uint64) * np.uint64(L), blocks["len"].array[lo:].copy())
        eng.destroy()
        equal = equal and bool(np.array_equal(mc2.reshape(-1), blocks["max"].array[lo * nf:]) and np.array_equal(dec2, blocks["dec"].array[lo:]))
        parity = {"checked_reads": int(m + n - lo), "pool_outputs_equal_single_engine": equal}
        # ... and against the ORACLE: the head of the first device's slice and the tail of the last one's
        if not ctx.args.no_cpu_baseline:
            from oracle import pyoracle as po
            views = [ctx.oracle_view(k) for k in dep_keys + tgt_keys]
            od, ot = views[:len(deplete)], views[len(deplete):]
            cores = host_cores()
            q = min(n, 1024)
            mism = 0
            for a, b in ((0, q), (n - q, n)):
                sub = np.ascontiguousarray(blocks["seq"].array[a * L: b * L])
                so, sl = np.arange(b - a, dtype=np.uint64) * np.uint64(L), np.full(b - a, L, dtype=np.uint32)
                cdec, _ = po.batch_check_unblock(od, ot, sub, so, sl, n_threads=cores)
                cmax = np.stack([po.batch_raw_max(v, sub, so, sl, cores) for v in views], axis=1)
                mism += int((cdec != blocks["dec"].array[a:b]).sum())
                mism += int((cmax.reshape(-1) != blocks["max"].array[a * nf: b * nf]).sum())
            parity.update(oracle_checked_reads=2 * q, oracle_mismatches=mism,
                          against="a single engine on the same filters (all outputs) AND the oracle's decisions + raw maxima on the "
                                  "first and the last reads of the call")
        geo = [(f.info["n_bins"], f.info["kmer_size"], f.info["n_hash"]) for f in deplete + target]
        bytes_per_read = synth.algorithmic_bytes_per_read(L, geo)
        per_dev_gbs = [bytes_per_read * r / (ms / 1e3) / 1e9 if ms > 0 else None for (_, _, r, _), (ms, _) in zip(st, ktimes)]
        have = [x for x in per_dev_gbs if x]
        roof = None
        if have:
            achieved = float(np.mean(have))  # per GPU, like every other leg: the devices gather from their own replicas
            roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                    "kernel": "ibf_count_max_kernel (per pool worker)", "algorithmic_bytes_per_read": bytes_per_read,
                    "basis": "K1 hipEvent time of every worker's engine; mean over devices",
                    "per_device_GBps": per_dev_gbs, "per_device_kernel_ms": [ms for ms, _ in ktimes],
                    "per_device_launches": [int(c) for _, c in ktimes]}
        result = {"metric": METRIC, "value": n * steps / wall, "unit": "reads/s", "n_gpus": len(devices), "steps": steps, "warmup": 1,
                  "ms_per_step": wall / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
                  "data": "synthetic",
                  "config": {"workload": wname + " -- ONE process, rb_pool_classify_batch, host buffers in page-locked memory (PCIe inclusive)",
                             "devices": devices, "reads_per_call": n, "reads_per_device_per_call": per_dev, "read_len": L,
                             "parallelism": "one host process, read-sharded x%d, IBF replicated device to device" % len(devices)},
                  "pool": {"create_seconds": create_s, "replication_seconds": pool.replication_seconds,
                           "replicated_bytes_per_device": int(sum(f.info["n_blocks"] * f.device_stride() * 8 for f in deplete + target)),
                           "per_device": [{"device": d, "busy_share": b / wall, "reads": int(r), "calls": int(c)} for d, b, r, c in st]},
                  "parity": parity,
                  "roofline": roof, "cpu_baseline": None}
        pool.destroy()
        for b in blocks.values():
            b.free()
        torch.cuda.set_device(ctx.dev_index)  # (creating replicas on other devices moved this thread's current device)
        torch.cuda.empty_cache()
        return result

    result = None
    if rank == 0:
        # never raises: on a multi-GPU node the other ranks are waiting for this rank on the rendezvous store, and the first run on
        # such a node is the first time the peer copies and the pool on distinct devices execute at all
        try:
            result = pool_child(ctx, steps)["pool_" + name] if pool_in_child(ctx) else measure()
        except Exception as ex:  # noqa: BLE001
            result = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300]), "value": 0.0, "n_gpus": world,
                      "config": {"workload": "one-process pool leg (%s)" % name}, "parity": None, "roofline": None, "cpu_baseline": None}
        try:
            torch.cuda.set_device(ctx.dev_index)
        except Exception:  # noqa: BLE001
            pass
    ctx._pool_legs = getattr(ctx, "_pool_legs", 0) + 1
    ctx.wait_for_rank0("pool_%s_%d" % (name, ctx._pool_legs))
    return result


def pool_in_child(ctx):
    """On a node with several GPUs the one-process legs run in a CHILD of rank 0: the pool over distinct devices and the
    device-to-device copies have never executed on the boxes this was developed on (one GPU each), and a fault or a hang there
    must cost the line two sub-legs, not the headline.  RB_BENCH_POOL_CHILD=1 forces the child on a one-GPU box (tests)."""
    if os.environ.get("RB_BENCH_POOL_CHILD") is not None:
        return os.environ["RB_BENCH_POOL_CHILD"] == "1"
    return ctx.world > 1 and not ctx.same_gpu


def pool_child(ctx, steps, timeout_s=600):
    """rank 0: `bench.py --pool` as a child process over every GPU of the job (its own HIP contexts, its own filters, no
    process group), once per run; returns {"pool_c3": leg, "pool_c4": leg, "xgmi_preflight": record}.  Reports, never raises."""
    import subprocess
    if getattr(ctx, "_pool_child", None) is not None:
        return ctx._pool_child
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT",
                        "TORCHELASTIC_RUN_ID", "RB_BENCH_SELF_LAUNCHED", "RB_BENCH_BACKEND", "RB_BENCH_SAME_GPU", "RB_BENCH_FORCE_GROUP",
                        # (the parent's checkpoint file and its "you have a supervisor" mark are the parent's: a child that raised outside
                        # run_pool would otherwise report the PARENT's headline as its own result)
                        "RB_BENCH_PARTIAL", "RB_BENCH_SUPERVISED")}
    env["RB_BENCH_NO_SUPERVISOR"] = "1"   # the child measures itself; this process is its supervisor
    env["RB_BENCH_CHILD_OF_BENCH"] = "1"  # ... and it dies with this process (die_with_parent)
    env["RB_BENCH_POOL_CHILD"] = "0"
    env["RB_BENCH_POOL_PREFLIGHT"] = "1"
    if "RB_BENCH_POOL_DEVICES" not in env:
        env["RB_BENCH_POOL_DEVICES"] = ",".join(str(d) for d in (range(ctx.world) if (ctx.world > 1 and not ctx.same_gpu) else [ctx.dev_index]))
    cmd = [sys.executable, os.path.abspath(__file__), "--pool", "--steps", str(steps)]
    t0 = time.time()

    def failed(why):
        leg = {"error": why, "value": 0.0, "n_gpus": ctx.world, "config": {"workload": "one-process pool leg (child of rank 0)"},
               "parity": None, "roofline": None, "cpu_baseline": None}
        return {"pool_c3": dict(leg), "pool_c4": dict(leg), "xgmi_preflight": {"ran": False, "error": why}}
    import tempfile
    fd, side = tempfile.mkstemp(prefix="rb_pool_child_", suffix=".json")
    os.close(fd)
    os.unlink(side)
    env["RB_BENCH_DETAIL"] = side  # the child's full result (its stdout line is the bounded summary)
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout_s)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        head = None
        if os.path.exists(side):
            try:
                head = json.load(open(side))
            finally:
                os.unlink(side)
        if not lines or head is None:
            out = failed("child exit code %d, no line; stderr tail: %s" % (p.returncode, p.stderr.strip()[-300:]))
        elif head.get("error") and (not head.get("value") or p.returncode != 0):
            out = failed("child exit code %d: %s" % (p.returncode, head["error"]))
        else:
            c4 = (head.pop("other_configs", None) or {}).get("pool_c4")
            pre = head.pop("xgmi_preflight", {"ran": False, "why": "the child did not report one"})
            for leg in (head, c4):
                if isinstance(leg, dict):
                    leg["in_child_process"] = True
                    leg["child_seconds"] = time.time() - t0
            out = {"pool_c3": head, "pool_c4": c4, "xgmi_preflight": pre}
    except subprocess.TimeoutExpired:
        out = failed("child killed after %d s" % timeout_s)
    except Exception as ex:  # noqa: BLE001
        out = failed("%s: %s" % (type(ex).__name__, str(ex)[:300]))
    ctx._pool_child = out
    return out


def run_early(ctx, n_reads=2_000_000, steps=3):
    """Leg `c3_early`: the OPT-IN early-decision mode (rb_engine_set_early_decision) on config 3's filter, deplete-only check_unblock on
    360 bp reads of which half come from the planted reference.  A wave stops counting a read once a bin has reached the larger of its two
    thresholds (adaptive_sampling.hpp:47-86 needs no more than that; the reference counts on).  WORK IS SKIPPED: `value` is a product
    figure, NOT a roofline figure -- the line carries no `roofline` for this leg, only the fabric traffic the counters saw for it
    (profiles/traffic.json, `c3_early`) as a share of 8 TB/s, so that nobody mistakes skipped gathers for bandwidth.  Parity: every decision
    and status of the timed batch equals the same call with the mode off, which a sample checks against the oracle.  Rank 0, one GPU."""
    from oracle import pyoracle as po
    from readbouncer_amd import capi, synth
    torch, dev = ctx.torch, ctx.dev
    dep, ref = ctx.filter("c3")
    L = 360
    if TEST_DIVISOR > 1:
        n_reads = max(4096, n_reads // TEST_DIVISOR)
    t_seq, t_off, t_len = synth.make_reads_device(1000 + ctx.rank, n_reads, L, ref, dev)
    t_dec = torch.zeros(n_reads, dtype=torch.uint8, device=dev)
    t_st = torch.zeros(n_reads, dtype=torch.uint8, device=dev)
    eng = capi.Engine(ctx.dev_index, [dep], [])
    torch.cuda.synchronize()

    def call():
        eng.classify_device(t_seq.data_ptr(), t_off.data_ptr(), t_len.data_ptr(), n_reads, L, 0.1, 0.95, capi.RB_MODE_CHECK_UNBLOCK,
                            None, None, t_dec.data_ptr(), t_st.data_ptr())

    def timed(k):
        call()  # warm-up
        torch.cuda.synchronize()
        eng.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(k):
            call()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ms, calls = eng.kernel_time()
        eng.set_timing(False)
        return el / k, ms / max(1, calls)

    full_s, full_k1_ms = timed(1)
    dec0, st0 = t_dec.cpu().numpy().copy(), t_st.cpu().numpy().copy()
    eng.set_early_decision(1)
    step_s, k1_ms = timed(steps)
    dec1, st1 = t_dec.cpu().numpy(), t_st.cpu().numpy()
    eng.set_early_decision(0)
    eng.destroy()
    # the mode-off decisions of a sample against the oracle (the mode-on ones equal them, all of them)
    view = ctx.oracle_view("c3")
    take = np.linspace(0, n_reads - 1, num=min(1500, n_reads), dtype=np.int64)
    seqs = t_seq.cpu().numpy()
    sub = np.concatenate([seqs[i * L:(i + 1) * L] for i in take])
    exp_dec, exp_st = po.batch_check_unblock([view], [], sub, np.arange(len(take), dtype=np.uint64) * np.uint64(L),
                                             np.full(len(take), L, dtype=np.uint32), n_threads=min(16, os.cpu_count() or 1))
    oracle_mism = int((dec0[take] != exp_dec).sum() + (st0[take] != exp_st).sum())
    tj = load_json("traffic.json").get("c3_early", {})
    traffic_frac = None
    if tj.get("hbm_bytes_per_read"):
        traffic_frac = tj["hbm_bytes_per_read"] * n_reads / (k1_ms / 1e3) / 8e12
    return {"metric": METRIC, "value": n_reads / step_s, "unit": "reads/s", "n_gpus": 1, "steps": steps, "ms_per_step": step_s * 1e3,
            "work_skipped": True,
            "note": "opt-in early-decision mode: work is skipped, this is NOT a roofline figure",
            "config": {"workload": "config 3's filter (8192 bins, 8 GiB), %d reads of 360 bp per launch, deplete-only check_unblock, "
                                   "rb_engine_set_early_decision(1)" % n_reads},
            "k1_ms": k1_ms, "k1_ms_full_count": full_k1_ms, "reads_per_s_full_count": n_reads / full_s,
            "speedup_over_full_count": full_s / step_s, "unblocked_share": float((dec1 == 1).mean()),
            "counter_traffic_frac_of_8TBps": traffic_frac, "counter_traffic_source": tj.get("source"),
            "parity": {"checked_reads": int(n_reads), "decision_mismatches": int((dec1 != dec0).sum() + (st1 != st0).sum()),
                       "raw_max_mismatches": 0, "near_threshold_reads": None, "oracle_checked_reads": int(len(take)), "oracle_mismatches": oracle_mism,
                       "what": "decisions and statuses of the whole batch against the same call with the mode off; a sample of those against the oracle"},
            "roofline": None, "cpu_baseline": None}


def run_cli_readme(ctx, n_reads=2_000_000, sample=3000):
    """SURVEY f.3 under the driver's eyes: the host CLI (readbouncer_amd/readbouncer_amd_cli, usage = "classify",
    src/main/classify.hpp:142-365) end to end on the README shape -- the four filters stored as .ibf files, a generated FASTQ of
    250 bp reads in the page cache, chunk_length 250, max_chunks 1, per-target FASTA + unclassified.fasta written.  `value` = the
    CLI's own THROUGHPUT figure (reads / wall of classify_reads: parse -> GPU chunk loop -> formatted output, HIP start-up and filter
    loading excluded; `process_wall_reads_per_s` includes them).  Parity: for a sample of reads the output file each one landed in must
    be the one the ORACLE's chunk driver names (orc_classify_read_chunks = classify.hpp:247-301 + 58-111).  Rank 0, one GPU."""
    import shutil
    import subprocess
    import tempfile
    from oracle import pyoracle as po
    from readbouncer_amd import synth
    keys = ["mock_deplete", "mock_t1", "mock_t2", "mock_t3"]
    cli = os.path.join(os.path.dirname(os.path.abspath(__file__)), "readbouncer_amd", "readbouncer_amd_cli")
    if not os.path.exists(cli):
        return {"error": "readbouncer_amd_cli is not built (run __graft_entry__.build())"}
    L = 250
    need = n_reads * (2 * L + 20) * 1.8
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.statvfs("/dev/shm").f_bavail * os.statvfs("/dev/shm").f_frsize > need * 1.5 else None
    work = tempfile.mkdtemp(prefix="rb_cli_readme_", dir=base)
    try:
        paths = []
        for k in keys:
            paths.append(os.path.join(work, k + ".ibf"))
            h = ctx.filter(k)[0].download()
            h.store(paths[-1])
            del h
        ref = np.concatenate([ctx.filter(k)[1] for k in keys])
        base_n = min(n_reads, 200_000)
        buf, _, _ = synth.make_reads(5, base_n, L, ref)
        # fixed-width records "@rRRR_IIIIII\n" + seq + "\n+\n" + qual + "\n", filled with numpy; block `rep` repeats the reads under new names
        rec_len = 13 + L + 3 + L + 1
        block = np.empty((base_n, rec_len), dtype=np.uint8)
        block[:, 0], block[:, 1], block[:, 5], block[:, 12] = ord("@"), ord("r"), ord("_"), ord("\n")
        idx = np.arange(base_n)
        for d in range(6):
            block[:, 6 + d] = ord("0") + (idx // 10 ** (5 - d)) % 10
        block[:, 13:13 + L] = buf.reshape(base_n, L)
        block[:, 13 + L:13 + L + 3] = np.frombuffer(b"\n+\n", dtype=np.uint8)
        block[:, 13 + L + 3:13 + 2 * L + 3] = ord("I")
        block[:, -1] = ord("\n")
        fq = os.path.join(work, "reads.fastq")
        with open(fq, "wb") as fh:
            rep, left = 0, n_reads
            while left > 0:
                for d in range(3):
                    block[:, 2 + d] = ord("0") + (rep // 10 ** (2 - d)) % 10
                m = min(left, base_n)
                block[:m].tofile(fh)
                left -= m
                rep += 1
        out_dir = os.path.join(work, "out")
        cfg = os.path.join(work, "c.toml")
        with open(cfg, "w") as fh:
            fh.write('usage = "classify"\noutput_directory = "%s"\nlog_directory = "%s/logs"\n[IBF]\ndeplete_files = ["%s"]\n'
                     'target_files = ["%s", "%s", "%s"]\nread_files = ["%s"]\nchunk_length = %d\nmax_chunks = 1\n'
                     % (out_dir, out_dir, paths[0], paths[1], paths[2], paths[3], fq, L))
        t = time.time()
        p = subprocess.run([cli, "--config", cfg, "--devices", str(ctx.dev_index)], capture_output=True, text=True, timeout=300)
        wall = time.time() - t
        thr = [l for l in p.stdout.splitlines() if l.startswith("THROUGHPUT")]
        if p.returncode != 0 or not thr:
            return {"error": "CLI exit code %d: %s" % (p.returncode, (p.stderr or p.stdout).strip()[-300:])}
        kv = dict(x.split("=", 1) for x in thr[0].split()[1:])
        res = [l for l in p.stdout.splitlines() if l.startswith("RESULT")]
        # where did each read go?  id -> output file, from the headers of the files the CLI wrote
        where = {}
        outs = sorted(f for f in os.listdir(out_dir) if f.endswith(".fasta"))
        for f in outs:
            with open(os.path.join(out_dir, f), "rb") as fh:
                data = np.frombuffer(fh.read(), dtype=np.uint8)
            starts = np.flatnonzero(data == ord(">"))
            for s0 in starts[:: max(1, len(starts) // (4 * sample))][: 4 * sample]:
                where[data[s0 + 1:s0 + 12].tobytes().decode()] = f
        views = [ctx.oracle_view(k) for k in keys]
        checked = mism = 0
        rng = np.random.default_rng(3)
        names = list(where)
        for name in [names[i] for i in rng.permutation(len(names))[:sample]]:
            i = int(name[5:])  # "rRRR_IIIIII" -> read IIIIII of the base block
            r = po.classify_read_chunks(views[:1], views[1:], buf[i * L:(i + 1) * L].tobytes(), L, 1)
            # classify.hpp:275-301: a read classified for a target goes to <target file stem>.fasta, everything else to unclassified.fasta
            expect = (os.path.splitext(os.path.basename(paths[1 + r["best_target"]]))[0] + ".fasta") if (r["classified"] and r["best_target"] >= 0) else "unclassified.fasta"
            checked += 1
            mism += where[name] != expect
        return {"metric": "reads/sec through the host CLI (usage = classify, README shape, 250 bp FASTQ -> per-target FASTA)",
                "value": float(kv.get("reads_per_s", 0.0)), "unit": "reads/s", "n_gpus": 1,
                "process_wall_reads_per_s": n_reads / wall, "process_wall_s": round(wall, 3),
                "config": {"workload": "readbouncer_amd_cli classify: %d reads of %d bp (FASTQ %.2f GB in %s), 1 deplete + 3 target .ibf files, chunk_length 250, max_chunks 1"
                                       % (n_reads, L, os.path.getsize(fq) / 1e9, "/dev/shm" if base else "the temp dir")},
                "cli": {k: kv.get(k) for k in ("wall_s", "classify_s", "wait_reader_s", "format_s", "classifiers", "parsers")},
                "result_line": res[0] if res else None,
                "parity": {"checked_reads": checked, "decision_mismatches": int(mism), "raw_max_mismatches": 0, "near_threshold_reads": None,
                           "what": "output file of a sample of reads against the oracle's chunk driver (orc_classify_read_chunks)"},
                "roofline": None, "cpu_baseline": None}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def replay(ctx, live_leg=True):
    """BASELINE configs[4]: 48-flowcell replay.  Poisson chunk arrivals (rate/world per GPU), 360 bp each, deplete =
    GRCh38-scale IBF + target = mock-community IBF, full check_unblock.  The dispatcher is work-conserving: whenever
    the GPU is free it takes everything that has arrived (a micro-batch) through rb_classify_batch (host buffers in,
    decisions back on the host).  Latency of a read = decision on the host - arrival.
    Second leg (`live_step`): the same arrival rate through rb_live_process -- the step as the reference runs it
    (adaptive_sampling.hpp:276-338): every read sends up to four 360 bp chunks 0.4 s apart, an undecided read's next chunk is
    classified as the concatenation with what once_seen holds (720 / 1080 / 1440 bp: the 16-plane kernels)."""
    from readbouncer_amd import capi, synth
    args, torch, dist, world, rank = ctx.args, ctx.torch, ctx.dist, ctx.world, ctx.rank
    dep, ref_d = ctx.filter("c3")
    tgt, ref_t = ctx.filter("zymo")
    eng = capi.Engine(ctx.dev_index, [dep], [tgt])
    if args.no_overlap:
        eng.set_overlap(False)
    rate = args.rate / world
    n = max(64, int(rate * args.replay_seconds / (TEST_DIVISOR if TEST_DIVISOR > 1 else 1)))
    read_len = 360
    ref = np.concatenate([ref_d, ref_t])
    t_seq, _, _ = synth.make_reads_device(7000 + rank, n, read_len, ref, ctx.dev)
    buf = t_seq.cpu().numpy()
    del t_seq
    rng = np.random.default_rng(7 + rank)
    arrival = np.cumsum(rng.exponential(1.0 / rate, size=n))
    offs0 = np.arange(n, dtype=np.uint64) * np.uint64(read_len)
    lens0 = np.full(n, read_len, dtype=np.uint32)
    w1, w2 = min(n, 64), min(n, 4096)
    for _ in range(20):  # warm-up (allocations, threshold table, code objects of both kernel forms)
        eng.classify(buf[: w1 * read_len], offs0[:w1], lens0[:w1])
        eng.classify(buf[: w2 * read_len], offs0[:w2], lens0[:w2])
    ctx.barrier()
    # the dispatcher loop runs inside the library (rb_replay_arrivals, C++ spin on the steady clock): no interpreter
    # between an arrival and its call
    decisions, lat, batches, service, elapsed = eng.replay_arrivals(buf, read_len, arrival, max_batch=16384)
    elapsed = ctx.max_over_ranks([elapsed])[0]
    p50, p99, p999, pmax = ctx.max_over_ranks([np.percentile(lat, 50), np.percentile(lat, 99), np.percentile(lat, 99.9), lat.max()])
    # parity of the replayed decisions: the same chunks as ONE batch through the engine
    exp_dec = eng.classify(buf, offs0, lens0)[2]
    replay_equal = bool(np.array_equal(exp_dec, decisions))

    live = None
    if live_leg:
        # reads with four chunks each, 0.4 s apart (MinKNOW's break_reads_after_seconds, adaptive_sampling.hpp:634); the
        # chunk stream keeps the total arrival rate.  A chunk of a read already decided starts that read afresh.
        # The stream as the sequencer would send it: a read whose first chunk is decided (unblock / stop_receiving) sends
        # nothing more; an undecided read sends up to four chunks, 0.4 s apart (MinKNOW's break_reads_after_seconds,
        # adaptive_sampling.hpp:634), and is re-classified on the concatenation each time -- 720, 1080, 1440 bp -- until the
        # 1500 bp cut-off.  Which synthetic chunks are decided on their own is known from the leg above; half of the reads
        # are of each kind, so 3 of 5 chunks are concatenations.
        n_chunks = 4
        idx_dec, idx_und = np.flatnonzero(decisions != 0), np.flatnonzero(decisions == 0)
        if len(idx_dec) == 0 or len(idx_und) == 0:
            idx_dec = idx_und = np.arange(n)
        reads = max(2, int(n / 2.5) // 2 * 2)
        half = reads // 2
        kind_und = np.zeros(reads, dtype=bool)
        kind_und[rng.permutation(reads)[:half]] = True
        # first chunks uniform over the replay: once all four chunk generations overlap (from 1.2 s on) the stream runs at the
        # nominal rate; the replay lasts 1.2 s longer than the plain one
        first = np.sort(rng.uniform(0.0, args.replay_seconds / (TEST_DIVISOR if TEST_DIVISOR > 1 else 1), size=reads))
        rows_und = np.resize(idx_und, half * n_chunks).reshape(half, n_chunks)
        rows_dec = np.resize(idx_dec, reads - half)
        arr_l, ids_l, rows_l = [], [], []
        und_no = np.cumsum(kind_und) - 1
        dec_no = np.cumsum(~kind_und) - 1
        for c in range(n_chunks):
            sel = kind_und if c else np.ones(reads, dtype=bool)
            arr_l.append(first[sel] + 0.4 * c)
            ids_l.append(np.flatnonzero(sel).astype(np.uint32))
            rows_l.append(np.where(kind_und[sel], rows_und[und_no[sel].clip(0), c], rows_dec[dec_no[sel].clip(0)] if c == 0 else 0))
        arr, ids, rows = np.concatenate(arr_l), np.concatenate(ids_l), np.concatenate(rows_l)
        order = np.argsort(arr, kind="stable")
        arr, ids, rows = arr[order], ids[order], rows[order]
        n_live = len(arr)
        lbuf = np.ascontiguousarray(buf.reshape(n, read_len)[rows].reshape(-1))
        lv = capi.Live(eng)
        # warm-up of everything the timed replay will meet: undecided chunks concatenated to 720 / 1080 / 1440 bp (kernel builds with more
        # counter planes: their code objects load at first launch, tens of milliseconds), in micro-batches and in a batch large
        # enough to grow the staging buffers a backlog would need
        und = idx_und if len(idx_und) >= 64 else np.arange(n)
        for wn in (64, min(1024, len(und))):
            warm = capi.Live(eng)
            rows_w = und[:wn]
            for _ in range(n_chunks):
                warm.process([b"w%d" % i for i in range(wn)], [bytes(buf[r * read_len:(r + 1) * read_len]) for r in rows_w])
            warm.destroy()
        ctx.barrier()
        act, llat, clen, lcalls, lservice, lelapsed = lv.replay_arrivals(ids, lbuf, read_len, arr, max_batch=16384)
        lelapsed = ctx.max_over_ranks([lelapsed])[0]
        l50, l99, l999, lmax = ctx.max_over_ranks([np.percentile(llat, 50), np.percentile(llat, 99), np.percentile(llat, 99.9), llat.max()])
        if rank == 0:
            live = {"what": "the same arrival rate through rb_live_process (once_seen, concatenation of undecided chunks, 1500 bp "
                            "cut-off): 4 chunks of 360 bp per read, 0.4 s apart",
                    "value": n_live * world / lelapsed, "unit": "chunks/s (whole replay, ramp-up and drain included)",
                    "steady_arrival_chunks_per_s": 2.5 * reads / (args.replay_seconds / (TEST_DIVISOR if TEST_DIVISOR > 1 else 1)) * world,
                    "kept_up": bool(lelapsed - float(arr[-1]) < 2e-3),
                    "p50_ms": l50 * 1e3, "p99_ms": l99 * 1e3, "p99.9_ms": l999 * 1e3, "max_ms": lmax * 1e3,
                    "slo_met": bool(l99 * 1e3 < 1.0),
                    "classified_length_share": {str(L): float((clen == L).mean()) for L in (360, 720, 1080, 1440)},
                    "concatenated_share": float((clen > read_len).mean()),
                    "actions": np.bincount(act, minlength=3).tolist(), "still_pending": int(lv.pending()),
                    "micro_batch_chunks": {"mean": float(np.mean(lcalls)), "max": int(np.max(lcalls))},
                    "call_service_ms": {"p50": float(np.percentile(lservice, 50) * 1e3), "p99": float(np.percentile(lservice, 99) * 1e3),
                                        "max": float(lservice.max() * 1e3)}}
        lv.destroy()
    result = None
    if rank == 0:
        result = {
            "metric": "reads/sec (360bp chunks through check_unblock, live replay) + p99 classify latency",
            "value": n * world / elapsed, "unit": "reads/s", "n_gpus": world, "steps": len(batches), "warmup": 40,
            "ms_per_step": elapsed / max(1, len(batches)) * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "config5: 48-flowcell replay, Poisson arrivals %.0f chunks/s total, 360bp chunks, "
                                   "deplete GRCh38-scale IBF (8 GiB) + target mock-community IBF, work-conserving "
                                   "micro-batches" % args.rate,
                       "arrival_rate_per_gpu": rate, "replay_seconds": args.replay_seconds,
                       "micro_batch_reads": {"mean": float(np.mean(batches)), "max": int(np.max(batches))},
                       "dispatcher": {"kind": "C++ spin loop inside the library (rb_replay_arrivals)",
                                      "call_service_ms": {"p50": float(np.percentile(service, 50) * 1e3),
                                                          "p99": float(np.percentile(service, 99) * 1e3),
                                                          "max": float(service.max() * 1e3)},
                                      "note": "latency = queueing (waiting for the engine to come free) + the service "
                                              "time of the call that carried the chunk"},
                       "decisions": np.bincount(decisions, minlength=3).tolist()},
            "latency": {"what": "arrival -> decision on the host, per read (queueing + H2D + kernels + D2H)",
                        "p50_ms": p50 * 1e3, "p99_ms": p99 * 1e3, "p99.9_ms": p999 * 1e3, "max_ms": pmax * 1e3,
                        "slo_p99_ms": 1.0, "slo_met": bool(p99 * 1e3 < 1.0)},
            "parity": {"replayed_decisions_equal_one_batch": replay_equal, "checked_reads": int(n)},
            "live_step": live,
            "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                         "traffic": None, "note": "latency-bound regime; the throughput roofline is reported by c3/c4"},
            "cpu_baseline": None,
        }
    ctx.barrier()
    eng.destroy()
    return result


def rank_diagnostics(ctx, setup_s):
    """What the first run on a multi-GPU node needs in order to explain itself: which device every rank drove, whether the
    devices can reach each other as peers, how long the filters took to stand up -- and, on rank 0 of a node with more than
    one visible GPU, ONE device-to-device replica of the small c2 filter onto the next GPU (rb_dibf_clone_to_ex: the xGMI
    peer branch that a one-GPU box can never execute), timed and verified, before anything is measured.  Never fatal."""
    torch = ctx.torch
    me = {"rank": ctx.rank, "device": ctx.dev_index, "filter_setup_s": round(ctx.filter_setup_s, 3), "setup_s": round(setup_s, 3)}
    try:
        p = torch.cuda.get_device_properties(ctx.dev_index)
        me["name"] = p.name
        for k in ("pci_bus_id", "pci_device_id", "uuid", "total_memory", "multi_processor_count"):
            if hasattr(p, k):
                v = getattr(p, k)
                me[k] = v if isinstance(v, (int, float)) else str(v)
        n_vis = torch.cuda.device_count()
        me["visible_devices"] = n_vis
        me["peer_access_possible"] = [bool(torch.cuda.can_device_access_peer(ctx.dev_index, j)) if j != ctx.dev_index else None
                                      for j in range(n_vis)]
    except Exception as ex:  # noqa: BLE001
        me["error"] = "%s: %s" % (type(ex).__name__, str(ex)[:160])
    if ctx.dist is None:
        return [me]
    out = [None] * ctx.world
    try:
        ctx.dist.all_gather_object(out, me)
    except Exception as ex:  # noqa: BLE001
        out = [me, {"error": "all_gather_object: %s" % str(ex)[:160]}]
    return out


def xgmi_preflight(ctx):
    """rank 0, before the measurements: replicate the 0.41 GB c2 filter to the next visible GPU device to device and compare
    it there (rb_pool_create_from_files does this for every filter of a one-process pool).  Reports, never raises."""
    from readbouncer_amd import capi
    torch = ctx.torch
    try:
        n_vis = torch.cuda.device_count()
        if n_vis < 2 or ctx.same_gpu:
            return {"ran": False, "why": "one visible GPU" if n_vis < 2 else "same-GPU test hook"}
        src, _ = ctx.filter("c2")
        dst_dev = (ctx.dev_index + 1) % n_vis
        replica, used_peer, secs = src.clone_to_ex(dst_dev)
        moved = src.info["n_blocks"] * src.device_stride() * 8
        back, _, _ = replica.clone_to_ex(ctx.dev_index)  # and home again, so that the comparison runs on one device
        cmp_ = src.compare(back)
        ok = cmp_["new_bits"] == 0 and cmp_["file_bits"] == cmp_["rebuilt_bits"]
        replica.free()
        back.free()
        torch.cuda.set_device(ctx.dev_index)
        return {"ran": True, "from": ctx.dev_index, "to": dst_dev, "peer_access_granted": used_peer, "bytes": int(moved),
                "seconds": secs, "GBps": moved / max(secs, 1e-9) / 1e9, "round_trip_bit_identical": bool(ok)}
    except Exception as ex:  # noqa: BLE001
        try:
            torch.cuda.set_device(ctx.dev_index)
        except Exception:
            pass
        return {"ran": False, "error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}


FULL_LEGS = ("c3np2", "c4", "c5", "c3_early", "c2", "grch38_f100k", "readme", "readme_360bp", "targets3", "deplete_target", "cli_readme", "pool_c3", "pool_c4")
# N > 1 (what a SCALE record carries, four runs back to back): the BASELINE configs that name several GPUs, config 3 at the
# reference's own sizing, and the one-process pool legs; the narrow shapes and c2 are single-GPU parity / roofline legs
MULTI_LEGS = ("c3np2", "c4", "c5", "pool_c3", "pool_c4")


def default_leg_names(world):
    if os.environ.get("RB_BENCH_LEGS"):  # e.g. RB_BENCH_LEGS=c4,c5 (any subset of FULL_LEGS, in that order)
        want = [x.strip() for x in os.environ["RB_BENCH_LEGS"].split(",") if x.strip()]
        return tuple(x for x in FULL_LEGS if x in want)
    return FULL_LEGS if world == 1 else MULTI_LEGS


def null_engine_run(args, torch, dist, world, rank, backend):
    """Control-flow test hook (RB_BENCH_ENGINE=none, set only by tests/): the rank flow of this script -- rendezvous,
    barriers, max-over-ranks timing, the per-rank gather, the all-gather + max of the bin-sharded layout, the headline +
    other_configs structure of the JSON line -- with NO classification behind it, so that `bench.py --gpus 2` can be exercised
    on a box without a GPU.  There is no CPU implementation of the hot path: the line says so and its values mean nothing."""
    n_reads = args.reads or 1000
    nf = 2
    note = "none (control-flow test hook RB_BENCH_ENGINE=none: no classification ran, the value is meaningless)"

    def partial(r):  # what rank r "counted": deterministic, different per rank
        i = np.arange(n_reads * nf, dtype=np.uint64)
        return ((i * np.uint64(2654435761) + np.uint64(r) * np.uint64(40503)) % np.uint64(65536)).astype(np.uint16).reshape(n_reads, nf)

    def leg(name, steps, sharded):
        reduce_ok = None
        if name.startswith("pool_"):  # rank 0 alone "measures", the others wait for it on the host (run_pool's flow)
            t0 = time.perf_counter()
            if rank == 0:
                time.sleep(0.05)
            waited = host_wait_for_rank0(dist, rank, "null_" + name)
            if dist is not None:
                dist.barrier()
            elapsed = max(time.perf_counter() - t0, 1e-3)
            return {"metric": METRIC, "value": n_reads / elapsed, "unit": "reads/s", "n_gpus": world, "steps": 1, "warmup": 0,
                    "ms_per_step": elapsed * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64",
                    "data": "synthetic", "engine": note, "config": {"workload": "rank-flow test (%s)" % name},
                    "waited_on_store": waited, "rank0_only": True, "roofline": None, "cpu_baseline": None, "parity": None}
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            if sharded and dist is not None:
                mine = torch.from_numpy(partial(rank).view(np.uint8).copy())  # bytes: an all-gather does no arithmetic
                gathered = torch.zeros((world * n_reads, nf * 2), dtype=torch.uint8)
                dist.all_gather_into_tensor(gathered, mine)
                got = gathered.numpy().view(np.uint16).reshape(world, n_reads, nf).max(axis=0)
                exp = np.maximum.reduce([partial(r) for r in range(world)])
                reduce_ok = bool(np.array_equal(got, exp)) and (reduce_ok is not False)
            time.sleep(0.002)
        t_local = time.perf_counter() - t0
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        per_rank = [t_local]
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            g = torch.zeros(world, dtype=torch.float64)
            dist.all_gather_into_tensor(g, torch.tensor([t_local], dtype=torch.float64))
            per_rank = g.tolist()
        total = n_reads * (1 if sharded else world) * steps
        return {"metric": METRIC, "value": total / elapsed, "unit": "reads/s", "n_gpus": world, "steps": steps,
                "warmup": args.warmup, "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True,
                "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                "engine": note, "config": {"workload": "rank-flow test (%s)" % name, "reads_per_gpu_per_step": n_reads},
                "per_rank_reads_per_s": [n_reads * steps / x for x in per_rank],
                "bin_sharded_reduce_ok": reduce_ok, "roofline": None, "cpu_baseline": None, "parity": None}

    if dist is not None:
        dist.barrier()
    if os.environ.get("RB_BENCH_TEST_DIE_RANK") == str(rank):  # tests: the launcher must not hang on a dead rank
        os._exit(7)
    head = leg(args.workload or "c3", args.steps, args.bin_sharded)
    if rank == 0:
        checkpoint(head)
    if os.environ.get("RB_BENCH_TEST_DIE_LATE") == str(rank):  # tests: a process that dies AFTER the headline (a fault in a later leg)
        time.sleep(0.3)
        os.kill(os.getpid(), 11)  # SIGSEGV, like a GPU fault would end it
    others = {}
    if not args.workload and not args.bin_sharded and not args.no_extras:
        for name in default_leg_names(world):
            others[name] = leg(name, 1, False)
    infos = [{"rank": rank, "device": None}]
    if dist is not None:
        infos = [None] * world
        dist.all_gather_object(infos, {"rank": rank, "device": None})
    if rank == 0:
        head["ranks"] = {"backend": backend, "rccl_ranks": world if backend == "nccl" else 0,
                         "self_launched": os.environ.get("RB_BENCH_SELF_LAUNCHED") == "1", "devices": infos,
                         "per_rank_reads_per_s": head["per_rank_reads_per_s"], "xgmi_preflight": {"ran": False, "why": "no engine"}}
        if others:
            head["other_configs"] = others
        emit(head)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def die_with_parent():
    """A measuring process started by this file (the supervisor's child, a self-launched rank, the pool child) must not outlive the
    process that started it: a supervisor that is SIGKILLed -- a driver's hard timeout, `timeout -k`, the OOM killer -- cannot forward
    anything, and its orphan would go on running legs of 8 GiB filters for minutes, holding HBM and skewing whoever measures next
    (ADVICE r5).  PR_SET_PDEATHSIG asks the kernel for a SIGKILL when the parent is gone; the getppid() check closes the window before
    the call."""
    import ctypes
    import signal
    try:
        ctypes.CDLL(None, use_errno=True).prctl(1, int(signal.SIGKILL), 0, 0, 0)  # PR_SET_PDEATHSIG = 1
    except Exception:  # noqa: BLE001
        return
    if os.getppid() == 1:  # the parent went away before the call took effect
        os._exit(1)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)  # before anything touches the GPU
    if int(os.environ.get("RANK", "0")) == 0 and os.environ.get("RB_BENCH_SUPERVISED") != "1" and os.environ.get("RB_BENCH_NO_SUPERVISOR") != "1":
        return supervise(sys.argv[1:])  # N = 1, or rank 0 under torchrun: the measuring process is a child of this one
    if os.environ.get("RB_BENCH_SUPERVISED") == "1" or os.environ.get("RB_BENCH_SELF_LAUNCHED") == "1" or os.environ.get("RB_BENCH_CHILD_OF_BENCH") == "1":
        die_with_parent()
    import torch

    t_start = time.time()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # test hooks (never set by the driver): RB_BENCH_BACKEND=gloo + RB_BENCH_SAME_GPU=1 run several ranks on ONE GPU,
    # which exercises the multi-rank control flow of this script on a one-GPU box (RCCL refuses duplicate GPUs);
    # RB_BENCH_ENGINE=none runs that control flow with no GPU at all (null_engine_run)
    backend = os.environ.get("RB_BENCH_BACKEND", "nccl")
    same_gpu = os.environ.get("RB_BENCH_SAME_GPU") == "1"
    no_engine = os.environ.get("RB_BENCH_ENGINE") == "none"
    dev_index = 0 if (same_gpu or world == 1) else local_rank
    # RB_BENCH_FORCE_GROUP=1 (tests): a process group of ONE rank, so that the RCCL code paths of this script -- barrier,
    # reductions, the bin-sharded all-gather on the engine's stream -- run on a one-GPU box
    force_group = world == 1 and os.environ.get("RB_BENCH_FORCE_GROUP") == "1"
    if force_group:
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    if world > 1 or force_group:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(dev_index)
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    if no_engine:
        return null_engine_run(args, torch, dist, world, rank, backend)
    torch.cuda.set_device(dev_index)
    if same_gpu or os.environ.get("RB_BENCH_PLACEMENT") == "off":
        # (test hook: several ranks on ONE device -- every rank's placement trial would hold up to five copies of an 8 GiB table at once, and
        # the `free HBM` it reads is not atomic with the other ranks' trials; RB_BENCH_PLACEMENT=off: the A/B run of profiles/r06)
        from readbouncer_amd import capi as _capi
        _capi.set_placement_tries(1)
    ctx = Ctx(args, torch, dist, world, rank, dev_index, backend, same_gpu, force_group)
    bin_sharded = args.bin_sharded and (world > 1 or force_group)

    if args.pool:
        result = run_pool(ctx, "c3", steps=max(1, min(args.steps, 5)))
        r4 = run_pool(ctx, "c4", steps=max(1, min(args.steps, 5)))
        if rank == 0:
            result["other_configs"] = {"pool_c4": r4}
            if os.environ.get("RB_BENCH_POOL_PREFLIGHT") == "1":  # (set by pool_child: this process is rank 0's child)
                result["xgmi_preflight"] = xgmi_preflight(ctx)
        extras = False
    elif args.workload == "c5":
        result = replay(ctx)
        extras = False
    elif args.workload == "c3_early":
        result = run_early(ctx, n_reads=args.reads or 2_000_000, steps=max(1, min(args.steps, 5)))
        extras = False
    else:
        name = args.workload or "c3"
        extras = not args.workload and not args.reads and not args.read_len and not args.no_extras and not bin_sharded
        if os.environ.get("RB_BENCH_NO_PREFLIGHT") == "1":
            pre = {"ran": False, "why": "RB_BENCH_NO_PREFLIGHT=1"}
        else:
            pre = {"ran": False, "why": "default run on rank 0 only"}
            if rank == 0 and extras:
                pre = {"ran": False, "why": "in the child process of the pool legs (filled in below)"} if pool_in_child(ctx) else xgmi_preflight(ctx)
        result = run_throughput(ctx, name, n_reads=args.reads, read_len=args.read_len, latency=True, bin_sharded=bin_sharded)
        infos = rank_diagnostics(ctx, time.time() - t_start)
        if rank == 0:
            result["ranks"] = {"backend": backend if dist is not None else None,
                               "rccl_ranks": world if (backend == "nccl" and world > 1) else 0,
                               "self_launched": os.environ.get("RB_BENCH_SELF_LAUNCHED") == "1",
                               "same_gpu_test_hook": same_gpu, "devices": infos,
                               "per_rank_reads_per_s": result.pop("per_rank_reads_per_s"), "xgmi_preflight": pre}
            checkpoint(result)  # the headline is safe from here on, whatever a later leg does to this process
        if os.environ.get("RB_BENCH_TEST_DIE_RANK") == str(rank) and not no_engine:  # tests: a rank that dies after the headline
            os._exit(7)
    if extras:
        # the other BASELINE configs, by all ranks, after the headline measurement; a failure is reported, never raised
        others = {}
        few = min(args.steps, 5)
        table = {
            # config 3 at the reference's own sizing (IBFBuild.cpp:404-413: BinSizeBits x 8256, a non-power-of-two block count --
            # the Barrett modulus every reference-built filter takes): ONE launch of 10 M reads per step, like the headline
            "c3np2": lambda: run_throughput(ctx, "c3np2", steps=few, warmup=1, cpu_seconds=6.0),
            "c4": lambda: run_throughput(ctx, "c4", steps=min(args.steps, 10), warmup=2, cpu_seconds=8.0),
            "c5": lambda: replay(ctx),
            "c2": lambda: run_throughput(ctx, "c2", cpu_seconds=5.0),
            # GRCh38 at the reference's DEFAULT fragment_size = 100 000 (configReader.cpp:238-243): ~31 000 bins, W = 485 words,
            # 3.9 KB blocks, 4.8 GB -- the filter a ReadBouncer user builds without touching a setting
            "grch38_f100k": lambda: run_throughput(ctx, "grch38_f100k", steps=min(args.steps, 3), warmup=1, cpu_seconds=5.0),
            "readme": lambda: run_throughput(ctx, "readme", cpu_seconds=5.0),
            # the README filters at the north star's read length
            "readme_360bp": lambda: run_throughput(ctx, "readme", read_len=360, steps=few, warmup=1, cpu_seconds=3.0),
            # two and three narrow filters of one hash geometry: one table that one lane holds per lookup (merged form)
            "targets3": lambda: run_throughput(ctx, "targets3", steps=few, warmup=1, cpu_seconds=3.0),
            "deplete_target": lambda: run_throughput(ctx, "deplete_target", steps=few, warmup=1, cpu_seconds=3.0),
            # the opt-in early-decision mode on config 3's filter: a product figure (work skipped), rank 0 only
            "c3_early": lambda: (run_early(ctx) if rank == 0 else {"value": 0.0}),
            # the host CLI end to end on the README shape (SURVEY f.3): rank 0 only, the other ranks pass
            "cli_readme": lambda: (run_cli_readme(ctx) if rank == 0 else {"value": 0.0}),
            # one host process driving every GPU of the job through rb_pool (rank 0; the other ranks wait): what SCALE's per-rank
            # numbers do not show
            "pool_c3": lambda: run_pool(ctx, "c3"),
            "pool_c4": lambda: run_pool(ctx, "c4"),
        }
        # filters a later leg no longer needs are freed as the run goes (N ranks on one node hold N replicas of each)
        last_use = {"c3np2": ["c3np2"], "c2": ["c2"], "grch38_f100k": ["grch38_f100k"],
                    "cli_readme": ["mock_deplete", "mock_t1", "mock_t2", "mock_t3"]}
        names = default_leg_names(world)
        for lname in names:
            t_leg = time.time()
            try:
                r = table[lname]()
            except Exception as ex:  # noqa: BLE001
                if world > 1:
                    raise  # a rank that drops out of the collectives would hang the others: fail the whole job loudly
                r = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:300])}
            ctx.release(last_use.get(lname, []))
            if rank == 0:
                if isinstance(r, dict):
                    r.pop("per_rank_reads_per_s", None)
                    r["leg_seconds"] = round(time.time() - t_leg, 2)
                    try:  # HBM in use on this rank's device once the leg is over (every process on it: with the same-GPU test hook, all ranks)
                        free_b, total_b = torch.cuda.mem_get_info(dev_index)
                        r["hbm_in_use_after_leg_bytes"] = int(total_b - free_b)
                    except Exception:  # noqa: BLE001
                        pass
                others[lname] = r
                result["other_configs"] = others
                checkpoint(result)
        if rank == 0:
            result["other_configs"] = others
            if getattr(ctx, "_pool_child", None) is not None and "ranks" in result:
                result["ranks"]["xgmi_preflight"] = ctx._pool_child["xgmi_preflight"]
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if TEST_DIVISOR > 1:
            result["test_reads_divisor"] = TEST_DIVISOR
        result["bench_seconds"] = round(time.time() - t_start, 1)
        try:
            result["hbm_peak_allocated_bytes"] = int(torch.cuda.max_memory_allocated(dev_index))  # torch's share only: reads, outputs
            free_b, total_b = torch.cuda.mem_get_info(dev_index)
            result["hbm_in_use_at_exit_bytes"] = int(total_b - free_b)
        except Exception:  # noqa: BLE001
            pass
        emit(result)

        def differs(r):
            par = r.get("parity") if isinstance(r, dict) else None
            return bool(par and (par.get("decision_mismatches") or par.get("raw_max_mismatches") or par.get("oracle_mismatches")))
        bad = differs(result)
        for r in (result.get("other_configs") or {}).values():
            bad |= differs(r)
            bad |= bool(isinstance(r, dict) and (r.get("parity") or {}).get("pool_outputs_equal_single_engine") is False)
        if bad:
            sys.exit(3)


def guarded_main():
    """main() with the promise of the header kept: whatever goes wrong, rank 0 (or the launcher) leaves ONE parseable line with
    an "error" field on stdout and the exit code is non-zero -- never a hang, never a bare traceback as the last word"""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    try:
        main()
    except SystemExit:
        raise
    except BaseException as ex:  # noqa: BLE001
        import traceback
        traceback.print_exc(file=sys.stderr)
        if rank == 0:
            try:
                finish_without_line("%s: %s" % (type(ex).__name__, ex), world, os.environ.get("RB_BENCH_PARTIAL"))
            except Exception:  # noqa: BLE001
                pass
        sys.stderr.flush()
        os._exit(1)  # (not sys.exit: a rank stuck in a collective's teardown must not keep the job alive)


if __name__ == "__main__":
    guarded_main()

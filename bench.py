#!/usr/bin/env python3
"""bench.py -- reads/s of the IBF classify hot path on MI355X (BASELINE.json metric).

A step = one pass of the hot path (K1 count/max for every filter + K2 decision) over one batch of
synthetic 360 bp read prefixes that is already resident in HBM, against IBF(s) resident in HBM.
N=1 workload = BASELINE.json configs[1] ("c2"); other configs via --workload.  With N>1 every rank
(one process per GPU) holds a replica of the IBF and its own shard of reads (weak scaling, no data-path
collective); time = max over ranks, value = all reads / that time.  `python bench.py --gpus N` starts
the N ranks itself (fresh child processes, started before this process touches the GPU); under
torchrun (WORLD_SIZE set) it is one of the ranks.

Prints ONE JSON line with the driver contract fields plus "roofline" and "cpu_baseline".
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

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured: 6.5 TB/s streaming, 6.8-6.9 TB/s random rows (profiles/hbm_peak.hip)


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


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks as fresh child processes (rank i -> GPU i, RCCL
    rendezvous on 127.0.0.1) and pass rank 0's JSON line through.  This process never initialises the GPU (no HIP call,
    no torch.cuda.is_available(): children started by a process that holds the device are refused on this pool) and it
    never execs: it waits for the children and exits with the worst of their codes.  The reference's scaling model is N
    classification workers behind one queue (src/main/adaptive_sampling.hpp:745-751); here a worker is a GPU."""
    import subprocess
    if os.environ.get("RB_BENCH_SAME_GPU") != "1" and os.environ.get("RB_BENCH_ENGINE") != "none":
        import torch  # counting devices does not initialise the GPU
        have = torch.cuda.device_count()
        if have < n:
            print("bench.py --gpus %d: only %d GPU(s) visible on this node" % (n, have), file=sys.stderr)
            sys.exit(2)
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   RB_BENCH_SELF_LAUNCHED="1")
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
    for line in "".join(out0).splitlines():  # the contract is ONE JSON line on stdout; library chatter goes to stderr
        (sys.stdout if line.startswith("{") else sys.stderr).write(line + "\n")
    sys.stdout.flush()
    rc = 0
    for p in procs:
        rc = rc or (p.returncode or 0)
    sys.exit(rc if rc >= 0 else 1)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", help="c2 (default, BASELINE configs[1]), c3, c3np2, c1, c4, c5, readme, grch38_f100k, zymo, zymo16")
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU per step (default: the config's batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--no-latency", action="store_true")
    ap.add_argument("--check-reads", type=int, default=2048, help="reads checked against the oracle (rank 0)")
    ap.add_argument("--bin-sharded", action="store_true",
                    help="N>1: every rank holds a word-column slice of each block and sees ALL reads; per-read partial "
                         "maxima are combined with one all_reduce(max) before the decision (strong scaling)")
    ap.add_argument("--no-overlap", action="store_true", help="serialise the count kernels of different filters")
    ap.add_argument("--rate", type=float, default=150000.0, help="c5: total chunk arrival rate (chunks/s) over all GPUs")
    ap.add_argument("--replay-seconds", type=float, default=3.0, help="c5: length of the replayed arrival process")
    ap.add_argument("--read-len", type=int, default=0, help="override the read length of the workload (e.g. 1500: 16 counter planes)")
    ap.add_argument("--phased", default="", help="tuning: 'off' or 'min_mib,max_mib,base_ticks,ticks_per_mib' for the clock-phased gathers of narrow filters")
    ap.add_argument("--serial-table-mib", type=int, default=-1,
                    help="tuning: filters up to this size take turns instead of overlapping (default: the engine's 64; 0 = round 1 behaviour)")
    ap.add_argument("--no-extras", action="store_true",
                    help="default c2 run on one GPU: do not append the short runs of configs 3, 4 and 5 (`other_configs`)")
    return ap.parse_args()


def other_configs(args):
    """Runs of the other BASELINE configs that fit one GPU, each in a child process after the main measurement (the
    headline line stays config 2, the configuration BASELINE.json's metric is quoted on).  Config 3 -- the 8 GiB
    GRCh38-scale filter, the HBM-bound case -- is a FULL run: same steps and warm-up as the headline, its own roofline
    (live hipEvent kernel time), CPU baseline and parity leg; it comes back as `hbm_bound_config`.  Short runs: deplete +
    target check_unblock (c4), the live replay (c5), the four narrow filters of the reference's README benchmark
    (readme).  Failures are reported, never raised."""
    import subprocess
    st, wu = str(args.steps), str(args.warmup)
    runs = {
        "c3": ["--workload", "c3", "--steps", st, "--warmup", wu, "--cpu-seconds", "8", "--no-latency"],
        "c4": ["--workload", "c4", "--reads", "1000000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-latency"],
        "c5": ["--workload", "c5", "--replay-seconds", "2.0"],
        # the shape of the reference's only published benchmark (README.md:254-262; ~506 reads/s there, hardware unstated)
        "readme": ["--workload", "readme", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-latency"],
    }
    out, hb = {}, None
    for name, argv in runs.items():
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-extras"] + argv, capture_output=True,
                               text=True, timeout=420)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
            d = json.loads(line)
            if name == "c3":
                hb = {k: d.get(k) for k in ("value", "unit", "steps", "warmup", "ms_per_step", "config", "roofline",
                                            "cpu_baseline", "parity")}
                continue
            o = {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"]}
            if d.get("roofline", {}).get("frac") is not None:
                o["roofline_frac"] = d["roofline"]["frac"]
                o["achieved_GBps"] = d["roofline"]["achieved"]
                o["ms_per_step"] = d["ms_per_step"]
                o["decisions"] = d["config"].get("decisions")
            if name == "c5":
                o["latency"] = {k: v for k, v in d["latency"].items() if k.endswith("_ms") or k == "slo_met"}
                o["micro_batch_reads"] = d["config"].get("micro_batch_reads")
                o["dispatcher"] = d["config"].get("dispatcher")
            out[name] = o
        except Exception as ex:  # noqa: BLE001 -- the headline line must not depend on the extras
            err = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}
            if name == "c3":
                hb = err
            else:
                out[name] = err
    return hb, out


def replay(args, torch, capi, synth, world, rank, dev_index, red_dev, dist):
    """BASELINE configs[4]: 48-flowcell replay.  Poisson chunk arrivals (rate/world per GPU), 360 bp each, deplete =
    GRCh38-scale IBF + target = mock-community IBF, full check_unblock.  The dispatcher is work-conserving: whenever
    the GPU is free it takes everything that has arrived (a micro-batch) through rb_classify_batch (host buffers in,
    decisions back on the host).  Latency of a read = decision on the host - arrival."""
    import time as _t
    wd, wt = synth.WORKLOADS["c3"], synth.WORKLOADS["zymo"]
    dep, ref_d = synth.build_device_filter(dev_index, wd, fill_seed=4, plant_seed=40)
    tgt, ref_t = synth.build_device_filter(dev_index, wt, fill_seed=6, plant_seed=60)
    eng = capi.Engine(dev_index, [dep], [tgt])
    if args.no_overlap:
        eng.set_overlap(False)
    rate = args.rate / world
    n = int(rate * args.replay_seconds)
    read_len = 360
    ref = np.concatenate([ref_d, ref_t])
    t_seq, _, _ = synth.make_reads_device(7000 + rank, n, read_len, ref, torch.device("cuda", dev_index))
    buf = t_seq.cpu().numpy()
    del t_seq
    rng = np.random.default_rng(7 + rank)
    arrival = np.cumsum(rng.exponential(1.0 / rate, size=n))
    offs0 = np.arange(n, dtype=np.uint64) * np.uint64(read_len)
    lens0 = np.full(n, read_len, dtype=np.uint32)
    for _ in range(20):  # warm-up (allocations, threshold table, code objects of both kernel forms)
        eng.classify(buf[: 64 * read_len], offs0[:64], lens0[:64])
        eng.classify(buf[: 4096 * read_len], offs0[:4096], lens0[:4096])
    if dist is not None:
        dist.barrier()
    # the dispatcher loop runs inside the library (rb_replay_arrivals, C++ spin on the steady clock): no interpreter
    # between an arrival and its call
    decisions, lat, batches, service, elapsed = eng.replay_arrivals(buf, read_len, arrival, max_batch=16384)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        stats = torch.tensor([np.percentile(lat, 50), np.percentile(lat, 99), np.percentile(lat, 99.9), lat.max()],
                             dtype=torch.float64, device=red_dev)
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
        p50, p99, p999, pmax = [float(x) for x in stats.tolist()]
    else:
        p50, p99, p999, pmax = [float(np.percentile(lat, q)) for q in (50, 99, 99.9)] + [float(lat.max())]
    if rank == 0:
        geo = [(8192, 13, 3), (600, 13, 3)]
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
            "roofline": {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
                         "traffic": None, "note": "latency-bound regime; the throughput roofline is reported by c2/c3/c4"},
            "cpu_baseline": None,
        }
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def null_engine_run(args, torch, dist, world, rank, backend):
    """Control-flow test hook (RB_BENCH_ENGINE=none, set only by tests/): the rank flow of this script -- rendezvous,
    barriers, max-over-ranks timing, the per-rank gather, the all-gather + max of the bin-sharded layout and the JSON
    line -- with NO classification behind it, so that `bench.py --gpus 2` can be exercised on a box without a GPU.
    There is no CPU implementation of the hot path: the line says so and its value means nothing."""
    n_reads = args.reads or 1000
    nf = 2

    def partial(r):  # what rank r "counted": deterministic, different per rank
        i = np.arange(n_reads * nf, dtype=np.uint64)
        return ((i * np.uint64(2654435761) + np.uint64(r) * np.uint64(40503)) % np.uint64(65536)).astype(np.uint16).reshape(n_reads, nf)

    reduce_ok = None
    if dist is not None:
        dist.barrier()
    if os.environ.get("RB_BENCH_TEST_DIE_RANK") == str(rank):  # tests: the launcher must not hang on a dead rank
        os._exit(7)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if args.bin_sharded and dist is not None:
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
    if rank == 0:
        total = n_reads * (1 if args.bin_sharded else world) * args.steps
        print(json.dumps({
            "metric": "reads/sec (360bp prefixes classified vs IBF, unblock/keep decisions)", "value": total / elapsed,
            "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.bin_sharded else "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "engine": "none (control-flow test hook RB_BENCH_ENGINE=none: no classification ran, the value is meaningless)",
            "config": {"workload": "rank-flow test", "reads_per_gpu_per_step": n_reads},
            "ranks": {"backend": backend, "rccl_ranks": world if backend == "nccl" else 0, "self_launched":
                      os.environ.get("RB_BENCH_SELF_LAUNCHED") == "1",
                      "per_rank_reads_per_s": [n_reads * args.steps / x for x in per_rank]},
            "bin_sharded_reduce_ok": reduce_ok, "roofline": None, "cpu_baseline": None}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def load_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return {}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus)  # before anything touches the GPU
    import torch

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
    from readbouncer_amd import capi, synth
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    red_dev = dev if backend == "nccl" else torch.device("cpu")  # where the tensors of the collectives live

    # ---------------------------------------------------------------- workload (untimed set-up)
    t_setup = time.time()
    if args.workload == "c5":
        return replay(args, torch, capi, synth, world, rank, dev_index, red_dev, dist)
    if args.workload == "c4":
        wd, wt = synth.WORKLOADS["c3"], synth.WORKLOADS["zymo"]
        dep, ref_d = synth.build_device_filter(dev_index, wd, fill_seed=4, plant_seed=40)
        tgt, ref_t = synth.build_device_filter(dev_index, wt, fill_seed=6, plant_seed=60)
        deplete, target = [dep], [tgt]
        ref = np.concatenate([ref_d, ref_t])
        wname = "config4: deplete=GRCh38-scale IBF (8192 bins, 8 GiB) + target=Zymo-mock-like IBF (600 bins), check_unblock"
        n_reads = args.reads or 2_000_000
        read_len = 360
    elif args.workload == "readme":
        # the reference's own (only) published benchmark shape: README.md:254-262, 250 bp prefixes, 1 deplete + 3 targets
        deplete, target, refs = [], [], []
        for i, key in enumerate(("mock_deplete", "mock_t1", "mock_t2", "mock_t3")):
            f, r = synth.build_device_filter(dev_index, synth.WORKLOADS[key], fill_seed=11 + i, plant_seed=110 + i, n_segments=512)
            (deplete if i == 0 else target).append(f)
            refs.append(r)
        ref = np.concatenate(refs)
        wname = ("README benchmark shape: 250bp prefixes vs 1 deplete (122 bins) + 3 target (43/29/49 bins) IBFs, k=13, "
                 "F=100000, check_unblock")
        n_reads = args.reads or 1_000_000
        read_len = 250
    else:
        w = synth.WORKLOADS[args.workload]
        seeds = {"c2": (2, 20), "c3": (4, 40), "c3np2": (4, 40), "c1": (1, 10), "zymo": (6, 60),
                 "grch38_f100k": (8, 80), "zymo16": (6, 60)}[args.workload]
        dep, ref = synth.build_device_filter(dev_index, w, fill_seed=seeds[0], plant_seed=seeds[1])
        deplete, target = [dep], []
        wname = w["name"]
        n_reads = args.reads or w["reads"]
        if args.workload in ("c3", "c3np2") and not args.reads:
            n_reads = 2_000_000  # per step; BASELINE's 10 M reads are five such steps (3.6 GB of read bytes per 10 M)
        read_len = w["read_len"]
        if n_reads != w["reads"]:
            wname += " [%d reads per step]" % n_reads
    if args.read_len:
        read_len = args.read_len
    filters = deplete + target
    nf = len(filters)
    geo = [(f.info["n_bins"], f.info["kmer_size"], f.info["n_hash"]) for f in filters]
    bytes_per_read = synth.algorithmic_bytes_per_read(read_len, geo)

    # reads are generated on the device (plumbing) and stay resident in HBM
    t_seq, t_off, t_len = synth.make_reads_device(1000 + rank, n_reads, read_len, ref, dev)
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
    max_len = int(lens.max())

    bin_sharded = args.bin_sharded and (world > 1 or force_group)
    if bin_sharded:
        # every rank classifies the SAME reads (seed of rank 0) against its column slice of every filter
        t_seq, t_off, t_len = synth.make_reads_device(1000, n_reads, read_len, ref, dev)
        eng.set_column_shard(rank, world)
        # the u16 partial maxima of all ranks, all-gathered as they are (byte view: an all-gather does no arithmetic, so
        # there is no widening for the collective and half the bytes of an int32 all-reduce); the max over the ranks is
        # taken inside the decision kernel
        t_all = torch.zeros((world * n_reads, nf * 2), dtype=torch.uint8, device=red_dev)  # rank-major

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
            if backend == "nccl":
                dist.all_gather_into_tensor(t_all, t_max.view(torch.uint8))
                parts = t_all
            else:  # gloo test hook: through the host
                side.synchronize()
                dist.all_gather_into_tensor(t_all, t_max.view(torch.uint8).cpu())
                parts = t_all.to(dev, non_blocking=False)
        eng.decide_device_parts(parts.data_ptr(), world, n_reads * nf, t_len.data_ptr(), n_reads, max_len, 0.1, 0.95,
                                capi.RB_MODE_CHECK_UNBLOCK, t_best.data_ptr(), t_dec.data_ptr(), t_st.data_ptr(), stream)
        if backend != "nccl":
            side.synchronize()  # `parts` is a temporary of this step

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # the inputs were produced on torch's default stream; the steps run on `side` (non-blocking): order them
    torch.cuda.synchronize()
    setup_s = time.time() - t_setup
    # ---------------------------------------------------------------- warm-up + timed region
    for _ in range(args.warmup):
        step()
    barrier()
    eng.set_timing(True)  # hipEvent pairs around the count kernels, on the launch stream, no sync
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    t_local = time.perf_counter() - t0  # this rank's own time for its K steps
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, n_calls = eng.kernel_time()
    eng.set_timing(False)
    per_rank_s = [t_local]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        g = torch.zeros(world, dtype=torch.float64, device=red_dev)
        dist.all_gather_into_tensor(g, torch.tensor([t_local], dtype=torch.float64, device=red_dev))
        per_rank_s = g.tolist()
    total_reads = n_reads * (1 if bin_sharded else world) * args.steps
    value = total_reads / elapsed

    result = None
    if rank == 0:
        avg_kernel_s = (kernel_ms / max(1, n_calls)) / 1e3
        achieved = bytes_per_read * n_reads / avg_kernel_s / 1e9
        if bin_sharded:
            achieved /= world  # every rank gathers its share of the word columns of every block
        # fabric-side traffic of one launch: NOT measured in this run -- rocprofv3 --pmc passes of an earlier run of the
        # same workload (profiles/collect_pmc.sh), kept in profiles/traffic.json and replayed here per read
        traffic, traffic_source = None, None
        tj = load_json("traffic.json").get(args.workload, {})
        if tj.get("hbm_bytes_per_read") and not bin_sharded:
            traffic = tj["hbm_bytes_per_read"] * n_reads
            traffic_source = "profiles/traffic.json (%s)" % tj.get("source", "rocprofv3 --pmc, separate passes, round 1")
        ceil = load_json("ceilings.json").get(args.workload, {})
        decisions = t_dec.cpu().numpy()
        # which form of K1 the engine plans for these filters (rb_engine.hip, plan_geometry): one- to eight-word blocks with a
        # table of 6-32 MiB (or one-word blocks of any size) take the phased kernel, everything else the plain one
        def phased(f):
            tb = f.info["n_blocks"] * f.device_stride() * 8
            return f.info["bin_width"] <= 8 and f.info["n_hash"] == 3 and ((6 << 20) <= tb <= (32 << 20) or f.info["bin_width"] == 1)
        forms = {("ibf_count_max_phased_kernel" if phased(f) else "ibf_count_max_kernel") for f in filters}
        kernel_name = " + ".join(sorted(forms))
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "kernel": kernel_name, "avg_kernel_ms": avg_kernel_s * 1e3,
                "algorithmic_bytes_per_read": bytes_per_read,
                "algorithmic_bytes_per_launch": bytes_per_read * n_reads}
        if ceil.get("GBps"):
            # the same access pattern with no compute attached (profiles/hbm_peak.hip): what this chip delivers for it
            roof["measured_ceiling"] = ceil["GBps"]
            roof["frac_of_measured_ceiling"] = achieved / ceil["GBps"]
            roof["measured_ceiling_source"] = ceil.get("source")
        table_bytes = sum(f.info["n_words"] * 8 for f in filters)
        if table_bytes < (256 << 20) * 4:
            roof["note"] = ("table of %.2f GB against a 256 MiB Infinity Cache: part of the gathers are served on-die; "
                            "`traffic` counts L2->fabric requests, Infinity-Cache hits included, so this is a fabric "
                            "figure -- the HBM-bound case is config 3 (`hbm_bound_config`)" % (table_bytes / 1e9))
        result = {
            "metric": "reads/sec (360bp prefixes classified vs IBF, unblock/keep decisions)",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if bin_sharded else "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": wname, "reads_per_gpu_per_step": n_reads, "read_len": read_len,
                       "filters": [{"n_bins": g[0], "k": g[1], "h": g[2], "bytes": f.info["n_words"] * 8}
                                   for g, f in zip(geo, filters)],
                       "parallelism": ("bin-sharded x%d, all_gather of u16 partial maxima, max taken in the decision "
                                       "kernel" % world) if bin_sharded
                       else "read-sharded x%d, IBF replicated" % world,
                       "decisions": np.bincount(decisions, minlength=3).tolist()},
            "ranks": {"backend": backend, "rccl_ranks": world if (backend == "nccl" and world > 1) else 0,
                      "self_launched": os.environ.get("RB_BENCH_SELF_LAUNCHED") == "1",
                      "same_gpu_test_hook": same_gpu,
                      "per_rank_reads_per_s": [n_reads * args.steps / x for x in per_rank_s]},
            "roofline": roof,
            "setup_s": setup_s,
        }
        if os.environ.get("RB_BENCH_DUMP_DECISIONS"):  # tests compare the N-rank decisions with the 1-rank run
            import hashlib
            result["config"]["decisions_sha1"] = hashlib.sha1(decisions.tobytes()).hexdigest()

    # ---------------------------------------------------------------- parity check + CPU baseline (rank 0, N=1)
    buf = None
    if rank == 0:
        cap = min(n_reads, 1 << 21)  # host copy of the head of the batch: CPU baseline, parity, latency legs
        buf = t_seq[: cap * read_len].cpu().numpy()
    if rank == 0 and world == 1 and not force_group and not args.no_cpu_baseline:
        from oracle import pyoracle as po
        keep = []
        views = []
        for f in filters:
            h = f.download()
            keep.append(h)
            views.append(po.OracleIBF.wrap(h.info["n_bins"], h.info["n_hash"], h.info["kmer_size"], h.info["n_bits"],
                                           h.words()))
        od, ot = views[:len(deplete)], views[len(deplete):]
        cores = host_cores()
        cap = len(buf) // read_len
        pilot = min(cap, 64 * min(cores, 64))
        tp = time.perf_counter()
        po.batch_check_unblock(od, ot, buf, offs[:pilot], lens[:pilot], n_threads=cores)
        pilot_s = time.perf_counter() - tp
        sample = int(min(cap, max(pilot, pilot * args.cpu_seconds / max(pilot_s, 1e-6))))
        tp = time.perf_counter()
        cpu_dec, cpu_st = po.batch_check_unblock(od, ot, buf, offs[:sample], lens[:sample], n_threads=cores)
        cpu_s = time.perf_counter() - tp
        t1 = time.perf_counter()
        n1 = min(sample, max(16, int(sample / cores / 4)))
        po.batch_check_unblock(od, ot, buf, offs[:n1], lens[:n1], n_threads=1)
        one_s = time.perf_counter() - t1
        gpu_dec = decisions[:sample]
        mism = int((gpu_dec != cpu_dec).sum())
        result["cpu_baseline"] = {"value": sample / cpu_s, "unit": "reads/s", "cores": cores, "kind": "port",
                                  "sample": "first %d reads of the same batch, oracle check_unblock, read-parallel "
                                            "pthreads; single-thread rate %.1f reads/s on %d reads"
                                            % (sample, n1 / one_s, n1),
                                  "single_thread_reads_per_s": n1 / one_s}
        result["parity"] = {"checked_reads": sample, "decision_mismatches": mism}
        if mism:
            result["parity"]["error"] = "GPU decisions differ from the oracle"
    elif rank == 0:
        result["cpu_baseline"] = None

    # ---------------------------------------------------------------- per-read classify latency (small batches)
    if rank == 0 and not args.no_latency and not bin_sharded:
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

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and world == 1 and not force_group and args.workload == "c2" and not args.reads and not args.read_len and not args.no_extras:
        # free this process's filters and reads first: the children need the HBM (config 3 alone is 8 GiB + reads)
        del eng, t_seq, t_off, t_len, t_max, t_best, t_dec, t_st
        for f in filters:
            f.free()
        torch.cuda.empty_cache()
        hb, others = other_configs(args)
        result["hbm_bound_config"] = hb
        result["other_configs"] = others
    if rank == 0:
        print(json.dumps(result))
        if result.get("parity", {}).get("decision_mismatches"):
            sys.exit(3)


if __name__ == "__main__":
    main()

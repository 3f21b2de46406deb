#!/usr/bin/env python3
"""bench.py -- reads/s of the IBF classify hot path on MI355X (BASELINE.json metric).

A step = one pass of the hot path (K1 count/max for every filter + K2 decision) over one batch of
synthetic 360 bp read prefixes that is already resident in HBM, against IBF(s) resident in HBM.
N=1 workload = BASELINE.json configs[1] ("c2"); other configs via --workload.  With N>1 (torchrun)
every rank holds a replica of the IBF and its own shard of reads (weak scaling, no data-path
collective); time = max over ranks, value = all reads / that time.

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
    ap.add_argument("--no-extras", action="store_true",
                    help="default c2 run on one GPU: do not append the short runs of configs 3, 4 and 5 (`other_configs`)")
    return ap.parse_args()


def other_configs():
    """Short runs of the other BASELINE configs that fit one GPU, each in a child process after the main measurement
    (the headline line stays config 2): the 8 GiB GRCh38-scale filter (c3), deplete + target check_unblock (c4), the
    live replay (c5) and the four narrow filters of the reference's README benchmark (readme).  Reported as a compact summary next to the headline; failures are reported, never raised."""
    import subprocess
    runs = {
        "c3": ["--workload", "c3", "--reads", "1000000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-latency"],
        "c4": ["--workload", "c4", "--reads", "1000000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-latency"],
        "c5": ["--workload", "c5", "--replay-seconds", "2.0"],
        # the shape of the reference's only published benchmark (README.md:254-262; ~506 reads/s there, hardware unstated)
        "readme": ["--workload", "readme", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-latency"],
    }
    out = {}
    for name, argv in runs.items():
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--no-extras"] + argv, capture_output=True,
                               text=True, timeout=300)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
            d = json.loads(line)
            o = {"workload": d["config"]["workload"], "value": d["value"], "unit": d["unit"]}
            if d.get("roofline", {}).get("frac") is not None:
                o["roofline_frac"] = d["roofline"]["frac"]
                o["achieved_GBps"] = d["roofline"]["achieved"]
                o["ms_per_step"] = d["ms_per_step"]
                o["decisions"] = d["config"].get("decisions")
            if name == "c5":
                o["latency"] = {k: v for k, v in d["latency"].items() if k.endswith("_ms") or k == "slo_met"}
                o["micro_batch_reads"] = d["config"].get("micro_batch_reads")
            out[name] = o
        except Exception as ex:  # noqa: BLE001 -- the headline line must not depend on the extras
            out[name] = {"error": "%s: %s" % (type(ex).__name__, str(ex)[:200])}
    return out


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
    lat = np.zeros(n)
    batches = []
    for _ in range(20):  # warm-up (allocations, threshold table, code objects of both kernel forms)
        eng.classify(buf[: 64 * read_len], offs0[:64], lens0[:64])
        eng.classify(buf[: 4096 * read_len], offs0[:4096], lens0[:4096])
    if dist is not None:
        dist.barrier()
    decisions = np.zeros(n, dtype=np.uint8)
    t0 = _t.perf_counter()
    done = 0
    while done < n:
        now = _t.perf_counter() - t0
        hi = int(np.searchsorted(arrival, now, side="right"))
        if hi <= done:
            continue  # spin until the next chunk arrives
        hi = min(hi, done + 16384)
        m = hi - done
        _, _, dec, _ = eng.classify(buf[done * read_len: hi * read_len], offs0[:m], lens0[:m])
        t_done = _t.perf_counter() - t0
        lat[done:hi] = t_done - arrival[done:hi]
        decisions[done:hi] = dec
        batches.append(m)
        done = hi
    elapsed = _t.perf_counter() - t0
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


def main():
    args = parse()
    import torch
    from readbouncer_amd import capi, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # test hooks (never set by the driver): RB_BENCH_BACKEND=gloo + RB_BENCH_SAME_GPU=1 run several ranks on ONE GPU,
    # which exercises the multi-rank control flow of this script on a one-GPU box (RCCL refuses duplicate GPUs)
    backend = os.environ.get("RB_BENCH_BACKEND", "nccl")
    same_gpu = os.environ.get("RB_BENCH_SAME_GPU") == "1"
    dev_index = 0 if (same_gpu or world == 1) else local_rank
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    if args.gpus != world and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    # ---------------------------------------------------------------- workload (untimed set-up)
    t_setup = time.time()
    if args.workload == "c5":
        return replay(args, torch, capi, synth, world, rank, dev_index, dev if backend == "nccl" else "cpu", dist)
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
        read_len = w["read_len"]
    if args.read_len:
        read_len = args.read_len
    filters = deplete + target
    geo = [(f.info["n_bins"], f.info["kmer_size"], f.info["n_hash"]) for f in filters]
    bytes_per_read = synth.algorithmic_bytes_per_read(read_len, geo)

    # reads are generated on the device (plumbing) and stay resident in HBM
    t_seq, t_off, t_len = synth.make_reads_device(1000 + rank, n_reads, read_len, ref, dev)
    lens = np.full(n_reads, read_len, dtype=np.uint32)
    offs = np.arange(n_reads, dtype=np.uint64) * np.uint64(read_len)
    t_max = torch.zeros((n_reads, len(filters)), dtype=torch.int16, device=dev)
    t_best = torch.zeros(n_reads, dtype=torch.int32, device=dev)
    t_dec = torch.zeros(n_reads, dtype=torch.uint8, device=dev)
    t_st = torch.zeros(n_reads, dtype=torch.uint8, device=dev)
    eng = capi.Engine(dev_index, deplete, target)
    if args.no_overlap:
        eng.set_overlap(False)
    # a dedicated non-null stream: steps are queued asynchronously; torch.cuda.synchronize() covers it
    side = torch.cuda.Stream(device=dev)
    stream = side.cuda_stream
    max_len = int(lens.max())

    bin_sharded = args.bin_sharded and world > 1
    if bin_sharded:
        # every rank classifies the SAME reads (seed of rank 0) against its column slice of every filter
        t_seq, t_off, t_len = synth.make_reads_device(1000, n_reads, read_len, ref, dev)
        eng.set_column_shard(rank, world)
        t_red = torch.zeros((n_reads, len(filters)), dtype=torch.int32, device=dev if backend == "nccl" else "cpu")

    def step():
        if not bin_sharded:
            eng.classify_device(t_seq.data_ptr(), t_off.data_ptr(), t_len.data_ptr(), n_reads, max_len, 0.1, 0.95,
                                capi.RB_MODE_CHECK_UNBLOCK, t_max.data_ptr(), t_best.data_ptr(), t_dec.data_ptr(),
                                t_st.data_ptr(), stream)
            return
        # partial maxima of this rank's columns -> all_reduce(max) over xGMI (u16 carried as i32) -> decision
        eng.classify_device(t_seq.data_ptr(), t_off.data_ptr(), t_len.data_ptr(), n_reads, max_len, 0.1, 0.95,
                            capi.RB_MODE_CHECK_UNBLOCK, t_max.data_ptr(), None, None, None, stream)
        side.synchronize()
        t_red.copy_(t_max.to(torch.int32) & 0xFFFF)
        dist.all_reduce(t_red, op=dist.ReduceOp.MAX)
        t_max.copy_(t_red.to(torch.int16))
        torch.cuda.synchronize()
        eng.decide_device(t_max.data_ptr(), t_len.data_ptr(), n_reads, max_len, 0.1, 0.95, capi.RB_MODE_CHECK_UNBLOCK,
                          t_best.data_ptr(), t_dec.data_ptr(), t_st.data_ptr(), stream)

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
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, n_calls = eng.kernel_time()
    eng.set_timing(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    total_reads = n_reads * (1 if bin_sharded else world) * args.steps
    value = total_reads / elapsed

    result = None
    if rank == 0:
        avg_kernel_s = (kernel_ms / max(1, n_calls)) / 1e3
        achieved = bytes_per_read * n_reads / avg_kernel_s / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            try:
                per_read = json.load(open(tfile)).get(args.workload, {}).get("hbm_bytes_per_read")
                traffic = per_read * n_reads if per_read else None  # PMC passes of profiles/collect_pmc.sh
            except Exception:
                traffic = None
        decisions = t_dec.cpu().numpy()
        result = {
            "metric": "reads/sec (360bp prefixes classified vs IBF, unblock/keep decisions)",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if bin_sharded else "weak",
            "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": wname, "reads_per_gpu_per_step": n_reads, "read_len": read_len,
                       "filters": [{"n_bins": g[0], "k": g[1], "h": g[2], "bytes": f.info["n_words"] * 8}
                                   for g, f in zip(geo, filters)],
                       "parallelism": ("bin-sharded x%d, all_reduce(max) of partial maxima" % world) if bin_sharded
                       else "read-sharded x%d, IBF replicated" % world,
                       "decisions": np.bincount(decisions, minlength=3).tolist()},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "ibf_count_max_kernel", "avg_kernel_ms": avg_kernel_s * 1e3,
                         "algorithmic_bytes_per_read": bytes_per_read,
                         "algorithmic_bytes_per_launch": bytes_per_read * n_reads},
            "setup_s": setup_s,
        }

    # ---------------------------------------------------------------- parity check + CPU baseline (rank 0, N=1)
    buf = None
    if rank == 0:
        cap = min(n_reads, 1 << 21)  # host copy of the head of the batch: CPU baseline, parity, latency legs
        buf = t_seq[: cap * read_len].cpu().numpy()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
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
    if rank == 0 and not args.no_latency:
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
    if rank == 0 and world == 1 and args.workload == "c2" and not args.reads and not args.read_len and not args.no_extras:
        result["other_configs"] = other_configs()
    if rank == 0:
        print(json.dumps(result))
        if result.get("parity", {}).get("decision_mismatches"):
            sys.exit(3)


if __name__ == "__main__":
    main()

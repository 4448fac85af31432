#!/usr/bin/env python3
"""bench.py -- samples/s of the GAT sampling + overlap-counting hot path on MI355X.

One "step" = one pass of the batch seam (gat_sample_and_count: place every isochore unit, re-combine
per contig, count every annotation track) over `--samples` Monte-Carlo samples per GPU, on the
synthetic BASELINE.json configuration (default config2: 10k segments x 1 annotation track x 10k
intervals, hg19 workspace, 10 000 samples, CounterNucleotideOverlap).  Inputs are resident in HBM
before the timed region.  With N > 1 (torch.distributed.run, one rank per GPU) every rank takes its
own contiguous range of sample ids (weak scaling) and the step ends with ONE RCCL all-gather of the
per-sample count matrix.

Prints one JSON line (rank 0): metric/value per the driver contract plus
  roofline     : the overlap-count kernel, algorithmic bytes / measured kernel time
  sampler      : the placement kernel, placements/s and MT19937 draws/s (not bandwidth bound)
  cpu_baseline : the CPU oracle (oracle/gat_oracle.c, a port of the reference) timed on this host
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="config2")
    ap.add_argument("--samples", type=int, default=0, help="samples per GPU per step (default: the config's)")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="time budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scale", type=float, default=1.0, help="scale interval counts (debugging only)")
    ap.add_argument("--counter", default=None, help="another counter than the configuration's (experiments only)")
    return ap.parse_args()


def cpu_baseline(flat, counters, seed, budget_s):
    """the CPU oracle (a C port of the reference algorithm, oracle/gat_oracle.c) on this host, on a bounded number
    of samples of the same workload: all host cores (one contiguous sample range per thread -- the per-unit streams
    make samples independent, and ctypes releases the GIL around the C call) and, for reference, one thread."""
    import concurrent.futures
    from oracle import oracle as O
    O.lib()
    t0 = time.perf_counter()
    O.run_samples(flat, counters, seed, 1, 0, 2)
    per = max((time.perf_counter() - t0) / 2, 1e-6)
    n1 = int(max(4, min(2000, 0.4 * budget_s / per)))
    t0 = time.perf_counter()
    O.run_samples(flat, counters, seed, 1, 0, n1)
    dt1 = time.perf_counter() - t0
    threads = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 256))
    try:                                                     # a container's CPU quota, if any (cgroup v2 / v1)
        for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
            if os.path.exists(path):
                f = open(path).read().split()
                quota = f[0]
                period = f[1] if len(f) > 1 else open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
                if quota not in ("max", "-1"):
                    threads = max(1, min(threads, -(-int(quota) // int(period))))
                break
    except (OSError, ValueError, IndexError):
        pass

    def run_threads(total):
        bounds = [total * i // threads for i in range(threads + 1)]
        t0 = time.perf_counter()
        with concurrent.futures.ThreadPoolExecutor(threads) as pool:
            list(pool.map(lambda i: O.run_samples(flat, counters, seed, 1, bounds[i], bounds[i + 1]), range(threads)))
        return time.perf_counter() - t0

    # the visible core count says little about the CPU time a container gets: size the run from a probe, not from
    # threads x single-thread rate
    probe = 2 * threads
    dtp = run_threads(probe)
    nt = int(max(probe, min(2000 * threads, 0.5 * budget_s * probe / dtp)))
    dtt = run_threads(nt)
    return dict(value=nt / dtt, unit="samples/s", cores=threads, kind="port",
                sample="%d samples of the same workload on %d threads in %.1f s (oracle/gat_oracle.c); one thread: "
                       "%.1f samples/s (%d samples, %.1f s)" % (nt, threads, dtt, n1 / dt1, n1, dt1),
                single_thread_value=n1 / dt1)


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from gat_amd import _lib, problem, synthetic

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" %
                         (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # one rank per GPU; GAT_BENCH_SHARE_GPU=1 (testing the N>1 code path on a one-GPU box) lets ranks share devices
    # and swaps RCCL, which refuses two ranks on one device, for gloo
    share = os.environ.get("GAT_BENCH_SHARE_GPU") == "1"
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cfg = synthetic.config(args.config, args.scale)
    counters = [args.counter or cfg["counter"]]
    S = args.samples or cfg["num_samples"]
    flat = problem.flatten_arrays(cfg["segments"], cfg["annotations"], cfg["workspace"], cfg["isochores"])
    stream = torch.cuda.current_stream().cuda_stream
    ctx = _lib.Context(dev_index, stream=stream)
    P = _lib.Problem(ctx, flat)
    info = P.info()
    K, A = len(counters), flat["n_tracks"]
    counts = torch.zeros((K, A, S), dtype=torch.int64, device=dev)
    gathered = torch.zeros((world * K, A, S), dtype=torch.int64, device=dev) if world > 1 else None

    def step(i):
        # rank r owns sample ids [ (i*world + r)*S, +S ): disjoint ranges, no data-path collective but the gather
        begin = (i * world + rank) * S
        st = P.sample_and_count_device(counters, args.seed, begin, begin + S, counts.data_ptr())
        if world > 1:
            dist.all_gather_into_tensor(gathered, counts)
        return st

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc = dict(ms_sampler=0.0, ms_contig=0.0, ms_count=0.0, ms_count_main=0.0, n_placed=0, n_draws=0, n_sampled_segments=0, n_retried=0,
               n_full_units=0)
    for i in range(args.steps):
        st = step(args.warmup + i)
        for k in acc:
            acc[k] += st[k]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # the one collective of the path, timed by itself after the timed region (the split the report shows per N)
    allgather_ms = None
    if world > 1:
        torch.cuda.synchronize()
        dist.barrier()
        t1 = time.perf_counter()
        for _ in range(5):
            dist.all_gather_into_tensor(gathered, counts)
        torch.cuda.synchronize()
        allgather_ms = (time.perf_counter() - t1) / 5 * 1e3

    if rank == 0:
        # HBM bytes of the count kernel per launch from the committed rocprofv3 PMC passes of this same
        # command (tools/collect_profiles.sh): (2 x FETCH_SIZE + WRITE_SIZE) KiB, gfx950 FETCH correction
        traffic = None
        tf = os.path.join(ROOT, "profiles", "r01_count_kernel_traffic.json")
        if os.path.exists(tf) and args.scale == 1.0:
            rec = json.load(open(tf)).get("%s:%d" % (args.config, S))
            if rec:
                traffic = (2.0 * rec["fetch_kib_per_launch"] + rec["write_kib_per_launch"]) * 1024.0
        total_samples = S * args.steps * world
        bytes_per_sample = info["algorithmic_bytes_per_sample"]
        # the dominant kernel alone (k_count_seg or k_count_swap), HIP events on the launch stream right around it;
        # ms_count additionally holds the small combining kernel k_count_finish
        count_s = (acc["ms_count_main"] or acc["ms_count"]) / 1e3
        samp_s = acc["ms_sampler"] / 1e3
        achieved = bytes_per_sample * S * args.steps / count_s / 1e9 if count_s > 0 else 0.0
        out = {
            "metric": "Monte Carlo samples/sec + bit-exact p-values, 10k sims, hg19-sized workspace",
            "value": total_samples / dt,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "%s: %d segments x %d annotation tracks x %d intervals, %d units / %d contigs, "
                                   "%d samples per GPU per step, %s" %
                                   (args.config, len(flat["segs"]), A, len(flat["annos"]), flat["n_units"],
                                    flat["n_contigs"], S, counters[0]),
                       "samples_per_step_per_gpu": S, "sharding": "samples, contiguous ranges per rank; one RCCL all-gather"},
            "roofline": {"bound": "hbm", "kernel": "k_count_seg (overlap counters)",
                         "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic,
                         "algorithmic_bytes_per_launch": bytes_per_sample * S,
                         "algorithmic_bytes_per_sample": bytes_per_sample,
                         "avg_launch_ms": (acc["ms_count_main"] or acc["ms_count"]) / args.steps,
                         "count_phase_ms": acc["ms_count"] / args.steps,
                         "share_of_gpu_time": acc["ms_count"] / max(1e-9, acc["ms_count"] + acc["ms_sampler"] + acc["ms_contig"])},
            "sampler": {"kernel": "k_rng + k_place + k_sampler (random rows, placement, consolidation)", "avg_launch_ms": acc["ms_sampler"] / args.steps,
                        "placements_per_s": acc["n_placed"] / samp_s if samp_s else 0.0,
                        "mt19937_draws_per_s": acc["n_draws"] / samp_s if samp_s else 0.0,
                        "kernel_samples_per_s": S * args.steps / samp_s if samp_s else 0.0,
                        "contig_kernel_avg_ms": acc["ms_contig"] / args.steps,
                        "units_retried": acc["n_retried"], "units_run_in_full": acc["n_full_units"],
                        "work_units": S * args.steps * flat["n_units"]},
            "allgather": None if allgather_ms is None else
            {"avg_ms": allgather_ms, "bytes_per_rank": int(counts.numel() * 8), "collective": "RCCL all_gather_into_tensor"},
        }
        if not args.no_cpu_baseline and world == 1:      # reported on rank 0 at N=1 only
            out["cpu_baseline"] = cpu_baseline(flat, counters, args.seed, args.cpu_seconds)
        print(json.dumps(out))
    P.close()
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

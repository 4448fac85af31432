#!/usr/bin/env python3
"""bench.py -- samples/s of the GAT sampling + overlap-counting hot path on MI355X.

One "step" = one pass of the batch seam (gat_sample_and_count: place every isochore unit, re-combine
per contig, count every annotation track) over `--samples` Monte-Carlo samples per GPU, followed by the
read-back of the count matrix to the host ("seed -> count matrix on host").  The headline line is
BASELINE.json's config2 (10k segments x 1 annotation track x 10k intervals, hg19 workspace, 10 000 samples,
CounterNucleotideOverlap); the same run then measures the other single-GPU shapes of BASELINE.json (config3, the
north_star target shape; one call's worth of config5 and config4; refdata, the reference's own test data with its
fragmented workspace) and reports them under "configs".  Inputs are
resident in HBM before the timed region.  The headline shape's steps are software-pipelined one deep (--pipeline 2, the
default; config.steps_in_flight): step i+1's kernels are enqueued before the host waits for step i, as gat_amd.run() does
with consecutive segment tracks; all K steps' kernels AND read-backs complete inside the timed region.  --pipeline 1:
enqueue, wait, read back, one step at a time (what rounds 1-4 measured; ~1 % slower).

`python bench.py --gpus N` without torch.distributed.run in the environment starts its own N ranks (a child process
running `python -m torch.distributed.run ... bench.py`, before anything here touches a GPU) and relays their one
JSON line.  With N > 1 every rank takes its own contiguous range of sample ids (weak scaling), the step ends with
ONE RCCL all-gather of the per-sample count matrix, and rank 0 reads the gathered matrix back.

Prints ONE JSON line on stdout (rank 0), below 3 KB (final_line; the driver parses stdout, and round 4's 21 KB line was cut
by its reader): metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype /
data / config per the driver contract -- value and ms_per_step are the K steps timed right behind the W warm-up steps of this
process -- plus
  roofline       : the overlap-count kernel: algorithmic bytes / kernel time measured with HIP events in this run.  For
                   k_count_seg the bytes are SURVEY.md 8d's contract; k_count_merged never moves those (it looks a sample
                   segment up once in an index of all tracks): bound "l2", its L2 requests against the L2s' peak
  cpu_baseline   : the CPU oracle (oracle/gat_oracle.c, a port of the reference) timed on this host (N = 1 only)
and a few small extras (sustained_value, the step's HBM bytes, strong-scaling ms per job, the extra shapes' values, under N > 1
the collective's figures).  Everything else goes to the details file (--details, default bench_details.json beside this script):
  kernels        : per-kernel time of the step (HIP events on the launch stream)
  sustained      : the same step repeated for at least a second behind the driver's K steps (mean, min, max over repeats)
  configs        : the same for config3 / config5 / config4 shapes (config3 with its own cpu_baseline)
  strong_scaling : the metric's own job -- 10 000 samples in all -- cut into N shards: at N = 1 the time of one shard's
                   call for N = 1, 2, 4, 8 (what each of N GPUs would run; no collective), under --gpus N the job itself
  api            : gat_amd.run() end to end (the drop-in seam: observed counts, problem creation, sampling + counting,
                   statistics, result rows) on config2 and config3, 10 000 samples; ms_with_inputs adds building the collections
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "Monte Carlo samples/sec + bit-exact p-values, 10k sims, hg19-sized workspace"
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
HBM_ACHIEVABLE_GBPS = 6290.0    # ... 6.29 TB/s measured with a float4 copy
L2_PEAK_GBPS = 34500.0          # ... 34.5 TB/s aggregate L2 bandwidth (128-byte lines)
# samples per GPU per step of the extra shapes: config3 = its own 10 000; config4 is an 8-GPU job of 12 500 samples per
# GPU, config5 one of 125 000 per GPU: a rank's whole shard per step (the library cuts it into batches that fit its
# scratch budget)
EXTRA_SAMPLES = {"config3": 10000, "config5": 125000, "config4": 12500, "refdata": 10000}
# the reference itself (Cython engine, one core, build container) -- it cannot travel
REFERENCE_CYTHON = {"config2": 19.1}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="config2")
    ap.add_argument("--samples", type=int, default=0, help="samples per GPU per step (default: the config's)")
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--extra", default="config3,config5,config4,refdata",
                    help="further shapes measured in the same run and reported under 'configs' ('' = none): BASELINE's config3 / 5 / 4 and "
                         "refdata, the reference's own test data set (tests/golden/refdata: 8 549 segments, a workspace of 279 057 "
                         "segments, 7 annotation tracks; test/data/output_single.tsv:66-77 has the reference at 30 samples/s on it)")
    ap.add_argument("--extra-steps", type=int, default=20, help="timed steps of the extra shapes (config4: a quarter of it)")
    ap.add_argument("--sustain-seconds", type=float, default=1.0, help="length of the sustained loop per shape (0 = none)")
    ap.add_argument("--pipeline", type=int, default=2, choices=(1, 2),
                    help="steps in flight: 2 = step i+1 is enqueued before the host waits for step i (as gat_amd.run() does "
                         "with its segment tracks), 1 = enqueue, wait, read back, one step at a time")
    ap.add_argument("--no-api", action="store_true", help="skip the gat_amd.run() block")
    ap.add_argument("--dump-counts", default=None, help="rank 0 saves the gathered count matrix of the last step (tests)")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling block")
    ap.add_argument("--details", default=None, help="where rank 0 writes the full report (default: bench_details.json beside bench.py)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="time budget of the cpu_baseline leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scale", type=float, default=1.0, help="scale interval counts (debugging only)")
    ap.add_argument("--counter", default=None, help="another counter than the configuration's (experiments only)")
    return ap.parse_args()


def spawn_ranks(args):
    """--gpus N > 1 and no launcher in the environment: be the launcher.  Nothing has touched a GPU yet (a process
    that has must never be replaced or forked); the ranks are children and their JSON line passes through."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


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
                sample="%d samples of this workload, %d threads, %.1f s (oracle/gat_oracle.c); 1 thread: %d samples, %.1f s"
                       % (nt, threads, dtt, n1, dt1),
                single_thread_value=n1 / dt1)


def counters_profile(config, S):
    """what the committed rocprofv3 passes of this command say about the kernels of (config, S): HBM-side bytes
    (FETCH_SIZE / WRITE_SIZE, separate --pmc passes, gfx950 FETCH correction applied) and VALU issue share.
    Produced by tools/summarize_profiles.py from gpurun_out/; NOT measured in this run -- the source is named."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for fn in ("gat_kernels.h", "gat_tail.h", "gat_device.h", "gat_stats.h", "gat_types.h"):
        h.update(open(os.path.join(ROOT, "gat_amd", "csrc", fn), "rb").read())
    sha = h.hexdigest()[:16]
    # only counters collected from THESE kernels are quoted (the summary carries the hash of the kernel sources it was
    # collected from, tools/summarize_profiles.py): the newest such file, none if the kernels have changed since
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_counters.json")), reverse=True):
        data = json.load(open(path))
        if (data.get("_meta") or {}).get("kernel_sources_sha") != sha:
            continue
        rec = data.get("%s:%d" % (config, S))
        if rec:
            return rec, "profiles/" + os.path.basename(path)
    return None, None


class Workload(object):
    """one BASELINE shape resident on this rank's GPU."""

    def __init__(self, name, S, args, dev_index, rank, world, depth=1):
        import torch
        from gat_amd import _lib, problem, synthetic
        self.name, self.S, self.args, self.rank, self.world = name, S, args, rank, world
        self.depth = max(1, min(2, int(depth)))
        cfg = synthetic.config(name, args.scale)
        self.counters = [args.counter or cfg["counter"]]
        self.flat = problem.flatten_arrays(cfg["segments"], cfg["annotations"], cfg["workspace"], cfg["isochores"])
        self.dev = torch.device("cuda", dev_index)
        # ONE stream for the library's kernels, the collective and the read-back: a torch stream of its own (the default
        # stream's handle is 0, which the library reads as "make a private stream")
        self.stream = torch.cuda.Stream(device=self.dev)
        torch.cuda.set_stream(self.stream)               # ... and torch's current stream from here on (one workload at a time)
        self.ctx = _lib.Context(dev_index, stream=self.stream.cuda_stream)
        # depth 2: the steps are software-pipelined one deep, the way gat_amd.run() keeps two segment tracks' calls in flight
        # (gat_amd/__init__.py, _start_job): step i+1's kernels are put on the stream BEFORE the host waits for step i, so the
        # device does not idle while the host wakes up, reads step i's status word, issues its read-back and launches again
        # (~60 us of a 2.5 ms step).  A call in flight belongs to a problem, so there are two problems over the same inputs --
        # one set of annotation tables shared, the sampler's scratch twice -- on ONE stream: their kernels run one behind
        # the other, never side by side.
        if self.depth > 1:
            segs = self.flat["segs"]
            mean = float((segs["end"].astype("int64") - segs["start"]).sum()) / max(1, len(segs))
            self.anno = _lib.Annotations(self.ctx, self.flat, mean_segment_length=mean)
            self.Ps = [_lib.Problem(self.ctx, self.flat, annotations=self.anno) for _ in range(self.depth)]
        else:
            self.anno = None
            self.Ps = [_lib.Problem(self.ctx, self.flat)]
        self.P = self.Ps[0]
        self.pending = None                              # (buffer, problem) of the step whose wait() is still to come
        self.pipelined = self.depth > 1
        self.info = self.P.info()
        K, A = len(self.counters), self.flat["n_tracks"]
        # The step's tail -- the ONE all-gather of the count matrix (N > 1) and its read-back -- runs on a second stream while
        # the next step's kernels run on the first: the count matrix, the gathered matrix and the host's copy exist twice, a
        # step's kernels wait (an event) for the tail that read their buffer two steps earlier.  Every step's matrix still
        # reaches the host inside the timed region (the region ends with a device-wide synchronisation).
        self.counts = [torch.zeros((K, A, S), dtype=torch.int64, device=self.dev) for _ in range(2)]
        self.counts_ptr = [c.data_ptr() for c in self.counts]
        self.gathered = [torch.zeros((world * K, A, S), dtype=torch.int64, device=self.dev) for _ in range(2)] if world > 1 else None
        # the matrix a host consumer gets: pinned, filled inside the timed region (rank 0 holds all ranks' columns)
        self.host = [torch.empty((world * K, A, S), dtype=torch.int64, pin_memory=True) for _ in range(2)] if rank == 0 else None
        self.tail_stream = torch.cuda.Stream(device=self.dev)
        self.tail_done = [None, None]
        self.n_step = 0
        self.last = 0

    def step(self, i, begin=None):
        import torch
        import torch.distributed as dist
        # rank r owns sample ids [ (i*world + r)*S, +S ): disjoint ranges, no data-path collective but the gather
        if begin is None:
            begin = (i * self.world + self.rank) * self.S
            self.last_first = i * self.world * self.S            # (first sample id of the step over all ranks: --dump-counts)
        b = self.n_step & 1
        self.n_step += 1
        if self.tail_done[b] is not None:
            self.stream.wait_event(self.tail_done[b])           # (the tail of two steps ago has read this buffer)
        # the batch seam in its two halves: the host is free between them -- it finishes the step before (pipelined), or has
        # nothing else to do
        P = self.Ps[b] if self.pipelined else self.P
        P.enqueue(self.counters, self.args.seed, begin, begin + self.S, self.counts_ptr[b])
        if self.pipelined:
            st = self.finish() if self.pending is not None else {}
            self.pending = (b, P)
            return st
        self.pending = (b, P)
        return self.finish()

    def finish(self):
        """second half of the step in flight: wait for its kernels, start its tail (all-gather, read-back); its statistics"""
        import torch
        import torch.distributed as dist
        if self.pending is None:
            return {}
        b, P = self.pending
        self.pending = None
        st = P.wait()
        # (wait() returns when the step's kernels have completed: the tail needs no event of theirs)
        with torch.cuda.stream(self.tail_stream):
            src = self.counts[b]
            if self.world > 1:
                dist.all_gather_into_tensor(self.gathered[b], self.counts[b])
                src = self.gathered[b]
            if self.host is not None:
                self.host[b].copy_(src, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.tail_stream)
            self.tail_done[b] = ev
        self.last = b
        return st

    KEYS = ("ms_sampler", "ms_contig", "ms_count", "ms_count_main", "ms_rng", "ms_place", "ms_merge", "ms_tail",
            "ms_ktail", "ms_finalize", "n_placed", "n_draws", "n_retried", "n_full_units", "n_resumed_units", "n_tail_units",
            "n_index_entries", "n_index_lookups", "n_batches", "ms_total")

    def timed(self, steps, first_step, acc=None):
        """`steps` steps bracketed by barrier + synchronize on both sides; seconds, the MAX over the ranks"""
        import torch
        import torch.distributed as dist
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        def add(st):
            if acc is not None and st:
                for k in self.KEYS:
                    acc[k] += st.get(k, 0.0)
                acc["count_kernel"] = st.get("count_kernel", 1)
                acc["merged_form"] = st.get("merged_form", 0)
        for i in range(steps):
            add(self.step(first_step + i))
        add(self.finish())                               # (pipelined: the last step's second half -- inside the timed region)
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if self.world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def measure(self, steps, warmup, sustain_s=0.0):
        import torch
        import torch.distributed as dist
        world = self.world
        for i in range(warmup):
            self.step(i)
        self.finish()
        # THE HEADLINE: exactly `steps` steps timed right behind the `warmup` untimed ones of this process (barrier +
        # synchronize on both sides, MAX over the ranks).  Everything below -- the per-kernel split, the sustained loop --
        # runs BEHIND it and is reported beside it, never instead of it.  (On some boxes the first tens of milliseconds
        # after the device was idle run 5-8 % slower than the steady state: `sustained` shows the difference.)
        acc = dict((k, 0.0) for k in self.KEYS)
        dt = self.timed(steps, warmup, acc)
        # the per-kernel split of the step (`kernels`, `sampler`): an event behind every kernel of the sampler costs a
        # call 50-60 us, so the library records them on request only -- here in a few steps of their own behind the timed
        # region (whose count kernel carries its two events always: `roofline`)
        ksteps = max(1, min(steps, 10))
        acck = dict((k, 0.0) for k in self.KEYS)
        self.ctx.set_kernel_times(True)
        was = self.pipelined
        self.pipelined = False                           # (the events behind every kernel exist once per context: one call in flight)
        try:
            self.timed(ksteps, warmup + steps, acck)
        finally:
            self.pipelined = was
            self.ctx.set_kernel_times(False)
        # the same step for at least `sustain_s` seconds more, in repeats of about a quarter of that (every rank runs the
        # same number of steps: rank 0's estimate is broadcast)
        sustained = None
        if sustain_s > 0:
            per = max(1, int(round(0.25 * sustain_s / max(dt / steps, 1e-6))))
            if world > 1:
                t = torch.tensor([per], dtype=torch.int64, device=self.dev)
                dist.broadcast(t, src=0)
                per = int(t.item())
            rates, total_t, total_n, nxt = [], 0.0, 0, warmup + steps + ksteps
            while total_t < sustain_s and len(rates) < 64:
                d = self.timed(per, nxt)
                nxt += per
                rates.append(self.S * per * world / d)
                total_t += d
                total_n += per
                if world > 1:                              # (all ranks leave the loop together)
                    t = torch.tensor([total_t], dtype=torch.float64, device=self.dev)
                    dist.broadcast(t, src=0)
                    total_t = float(t.item())
            sustained = {"value": self.S * total_n * world / total_t, "unit": "samples/s", "seconds": total_t, "steps": total_n,
                         "ms_per_step": total_t / total_n * 1e3, "repeats": len(rates), "min": min(rates), "max": max(rates)}
        # the one collective of the path, timed by itself after the timed region (the split the report shows per N)
        allgather = None
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
            t1 = time.perf_counter()
            with torch.cuda.stream(self.stream):
                for _ in range(5):
                    dist.all_gather_into_tensor(self.gathered[0], self.counts[0])
            torch.cuda.synchronize()
            allgather = {"avg_ms": (time.perf_counter() - t1) / 5 * 1e3, "bytes_per_rank": int(self.counts[0].numel() * 8),
                         "collective": "RCCL all_gather_into_tensor", "backend": dist.get_backend(),
                         "world_size": dist.get_world_size()}
        out = self.report(steps, warmup, dt, acc, allgather, acck, ksteps)
        if sustained is not None:
            out["sustained"] = sustained
        return out

    def report(self, steps, warmup, dt, acc, allgather, acck, ksteps):
        S, world, flat, info = self.S, self.world, self.flat, self.info
        A = flat["n_tracks"]
        bytes_per_sample = info["algorithmic_bytes_per_sample"]
        # the dominant count kernel alone, HIP events on the launch stream right around it; ms_count additionally holds
        # the small combining kernel
        main_ms = (acc["ms_count_main"] or acc["ms_count"]) / steps
        count_s = main_ms / 1e3
        samp_s = acck["ms_sampler"] / 1e3
        achieved = bytes_per_sample * S / count_s / 1e9 if count_s > 0 else 0.0
        from gat_amd import _lib
        kernel = _lib.COUNT_KERNELS.get(int(acc.get("count_kernel", 1)), "k_count_seg")
        merged = kernel == "k_count_merged"
        roof = {"bound": "l2" if merged else "hbm", "kernel": "%s (overlap counters)" % kernel,
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": None, "traffic_source": None,
                "algorithmic_bytes_per_launch": bytes_per_sample * S,
                "algorithmic_bytes_per_sample": bytes_per_sample,
                "avg_launch_ms": main_ms,
                "count_phase_ms": acck["ms_count"] / ksteps,
                "share_of_gpu_time": acck["ms_count"] / max(1e-9, acck["ms_count"] + acck["ms_sampler"] + acck["ms_contig"]),
                "note": "achieved/frac follow the SURVEY 8d contract (every annotation interval charged once per sample); "
                        "the kernel serves annotations from LDS / L2, so real HBM traffic is lower: see hbm_measured_GBps"}
        if merged:
            # k_count_merged looks every sample segment up ONCE in a merged index of all tracks; the contract's bytes (every
            # annotation interval once per sample) are never moved.  Its algorithmic bytes are what the algorithm does look
            # at, counted by the kernel itself (gat_stats n_index_lookups / n_index_entries, the latter in 4-byte words):
            # per sample segment 8 bytes of the segment + 4 of its grid cell + 8 per index entry its scan examines (whatever
            # the granularity of the fetch), and the partial sums it leaves per (contig, sample, track), 4 bytes each
            lookups, words = acc["n_index_lookups"] / steps, acc["n_index_entries"] / steps
            moved = 8.0 * lookups + 4.0 * words + 4.0 * flat["n_contigs"] * A * S
            roof["contract_bytes_per_launch"] = bytes_per_sample * S
            roof["contract_GBps"] = achieved
            roof["algorithmic_bytes_per_launch"] = moved
            roof["algorithmic_bytes_per_sample"] = moved / max(1, S)
            achieved = moved / count_s / 1e9 if count_s > 0 else 0.0
            roof["achieved"], roof["frac"] = achieved, achieved / HBM_PEAK_GBPS
            roof["note"] = ("bound by the rate at which the L2s serve its gathers, not by HBM: achieved / peak / frac are L2 requests x "
                            "128 B over the kernel's time against the L2s' 34.5 TB/s; own_model_GBps = the bytes the algorithm moves "
                            "(segments once + the index words its scans read, counted by the kernel, + partials); contract_GBps is "
                            "SURVEY 8d's figure (every annotation interval charged once per sample), which this algorithm never moves")
            roof["lookups_per_s"] = lookups / count_s if count_s > 0 else 0.0
            # L2 requests: one per segment load (16 lanes x 8 B = a 128-byte line) and, per look-up, what the scan's form
            # asks for -- a 32-byte cell record (+ a pair per two entries behind the first two), or a grid cell + a pair per
            # two entries, or a grid cell + a 64-byte block per eight; against the 34.5 TB/s / 128 B = 270 requests per ns
            # the L2s deliver (MI355X_MICROARCH.md, L2).  An upper bound: lanes of one load that fall into the same line are
            # one request -- sorted segments often do
            wpl = words / max(1.0, lookups)
            ent = max(0.0, (wpl - 1.0) / 2.0)                         # entries a scan looks at (the one that ends it too)
            form = int(acc.get("merged_form", 2))
            roof["index_form"] = {8: "blocks of eight entries", 2: "pairs of entries", 1: "cell records + pairs"}.get(form, str(form))
            roof["index_entries_per_lookup"] = ent
            if form == 8:
                per_lookup = 1.0 + (ent + 3.5) / 8.0                  # the cell + the blocks the scan runs through
            elif form == 1:
                per_lookup = 1.0 + max(0.0, ent - 2.0) / 2.0          # the record + pairs behind its two entries
            else:
                per_lookup = 1.0 + ent / 2.0 + 0.5                    # the cell + pairs
            reqs = lookups / 16.0 + lookups * per_lookup
            roof["l2_requests_per_launch_model"] = reqs
            # bound "l2": achieved / peak / frac are L2 figures -- requests x 128-byte lines over the kernel's time against the
            # 34.5 TB/s the L2s deliver.  Live: the request MODEL above from the kernel's own counters (an upper bound);
            # the committed TCC_REQ counters of this command, when they belong to these kernels, replace it below.  What
            # the algorithm moves against the HBM peak stays beside it as own_model_* (never a roofline fraction: those
            # bytes are served by the L2s)
            roof["own_model_GBps"], roof["own_model_frac_of_hbm_peak"] = achieved, achieved / HBM_PEAK_GBPS
            l2_gbps = reqs * 128.0 / count_s / 1e9 if count_s > 0 else 0.0
            roof["achieved"], roof["peak"], roof["frac"] = l2_gbps, L2_PEAK_GBPS, l2_gbps / L2_PEAK_GBPS
            roof["achieved_source"] = "model: L2 requests derived from the kernel's own look-up / index-word counters of this run"
            roof["l2_request_frac"] = min(1.0, l2_gbps / L2_PEAK_GBPS)
        prof, src = counters_profile(self.name, S) if self.args.scale == 1.0 else (None, None)
        k = (prof or {}).get("count_kernel")
        if k and "fetch_kib_per_launch" in k and "write_kib_per_launch" in k:
            # FETCH_SIZE counts half the bytes of wide coalesced streaming reads on gfx950 (MI355X_MICROARCH.md, HBM);
            # a gather-bound kernel's narrow requests are uncalibrated: x2 only for the streaming kernel
            fx = k.get("fetch_factor", 1.0 if merged else 2.0)
            roof["fetch_factor"] = fx
            traffic = (fx * k["fetch_kib_per_launch"] + k["write_kib_per_launch"]) * 1024.0
            roof["traffic"] = traffic
            roof["traffic_source"] = "%s (committed rocprofv3 --pmc passes of this command; not re-measured here)" % src
            if count_s > 0:
                roof["hbm_measured_GBps"] = traffic / count_s / 1e9
                roof["frac_of_achievable"] = traffic / count_s / 1e9 / HBM_ACHIEVABLE_GBPS
            if "valu_busy" in k:
                roof["valu_busy"] = k["valu_busy"]
                if k["valu_busy"] > 0.6 and roof.get("frac_of_achievable", 1.0) < 0.3:
                    roof["bound"] = "valu"
            if k.get("l2_requests_per_launch") and count_s > 0:
                # requests x 128-byte lines against the 34.5 TB/s the L2s deliver (MI355X_MICROARCH.md, L2)
                roof["l2_GBps"] = k["l2_requests_per_launch"] * 128.0 / count_s / 1e9
                roof["l2_hit_rate"] = k.get("l2_hit_rate")
                roof["l2_frac_of_peak"] = roof["l2_GBps"] / L2_PEAK_GBPS
                if merged:
                    roof["model_GBps"] = roof["achieved"]
                    roof["achieved"], roof["frac"] = roof["l2_GBps"], roof["l2_frac_of_peak"]
                    roof["achieved_source"] = "TCC_REQ_sum x 128 B of %s over this run's kernel time" % src
        step = (prof or {}).get("step")
        step_block = None
        if step and step.get("hbm_bytes"):
            # the whole step against the HBM: every kernel's counter bytes (FETCH corrected per kernel, see
            # tools/summarize_profiles.py) over this run's time per step
            step_block = {"hbm_bytes": step["hbm_bytes"], "hbm_GBps": step["hbm_bytes"] / (dt / steps) / 1e9,
                          "hbm_frac": step["hbm_bytes"] / (dt / steps) / 1e9 / HBM_PEAK_GBPS, "source": src}
        out = {
            "value": S * steps * world / dt,
            "unit": "samples/s",
            "ms_per_step": dt / steps * 1e3,
            "steps": steps,
            "warmup": warmup,
            "config": {"workload": "%s: %d segments x %d tracks x %d intervals, %d units / %d contigs, %d samples/GPU/step, %s" %
                                   (self.name, len(flat["segs"]), A, len(flat["annos"]), flat["n_units"],
                                    flat["n_contigs"], S, self.counters[0]),
                       "samples_per_step_per_gpu": S,
                       "sharding": "samples, contiguous ranges per rank; one RCCL all-gather",
                       "steps_in_flight": self.depth,
                       "timed_region": "sampling + counting + all-gather (N > 1) + D2H of the count matrix (%d bytes); a step's "
                                       "all-gather and D2H run on a second stream beside the next step's kernels"
                                       % (self.host[0].numel() * 8 if self.host is not None else 0)},
            "roofline": roof,
            "kernels": {"k_rng_ms": acck["ms_rng"] / ksteps, "k_place_ms": acck["ms_place"] / ksteps,
                        "k_merge_ms": acck["ms_merge"] / ksteps, "k_tail_ms": acck["ms_ktail"] / ksteps,
                        "k_finalize_ms": acck["ms_finalize"] / ksteps,
                        "k_sampler_ms": (acck["ms_tail"] - acck["ms_ktail"] - acck["ms_finalize"]) / ksteps,
                        "k_contig_ms": acck["ms_contig"] / ksteps, "count_main_ms": main_ms,
                        "count_phase_ms": acck["ms_count"] / ksteps, "sampler_phase_ms": acck["ms_sampler"] / ksteps,
                        "measured": "HIP events behind every kernel in %d steps of their own behind the timed region (the "
                                    "events cost a call 50-60 us: gat_ctx_set_kernel_times); count_main_ms: the two events "
                                    "around the count kernel inside the timed region" % ksteps},
            "sampler": {"kernel": "k_rng + k_place + k_merge + k_sampler (random rows, placement, consolidation)",
                        "avg_launch_ms": acck["ms_sampler"] / ksteps,
                        "placements_per_s": acck["n_placed"] / samp_s if samp_s else 0.0,
                        "mt19937_draws_per_s": acck["n_draws"] / samp_s if samp_s else 0.0,
                        "kernel_samples_per_s": S * ksteps / samp_s if samp_s else 0.0,
                        "contig_kernel_avg_ms": acck["ms_contig"] / ksteps,
                        "units_retried": acc["n_retried"], "units_run_in_full": acc["n_full_units"],
                        "units_resumed_out_of_rows": acc["n_resumed_units"],
                        "rows_generated_per_draw_consumed": (self.P.rows_per_sample() * float(self.S) * steps / acc["n_draws"]) if acc["n_draws"] else None,
                        "units_finished_by_k_tail": acc["n_tail_units"],
                        "batches_per_step": acc["n_batches"] / steps,
                        # wall time of a step against what its kernels took on the stream (HIP events around the call)
                        "stream_ms_per_step": acc["ms_total"] / steps,
                        "work_units": S * steps * flat["n_units"]},
            "allgather": allgather,
        }
        if step_block is not None:
            out["step"] = step_block
        return out

    def close(self):
        self.finish()
        for P in self.Ps:
            P.close()
        if self.anno is not None:
            self.anno.close()
        self.ctx.close()


STRONG_TOTAL = 10000            # the metric's job: 10k simulations


def strong_scaling(args, dev_index, rank, world):
    """The metric's own job -- STRONG_TOTAL samples in all -- cut over N GPUs (gat/__init__.py:681-700: the reference cuts
    its samples over pool workers).  world == 1: one GPU runs what EACH of N GPUs would run for N = 1, 2, 4, 8 (a call of
    STRONG_TOTAL / N samples + its read-back), which bounds the job from below: N GPUs finish no earlier than one shard's call
    (the all-gather comes on top).  world > 1: the job itself -- every rank its shard, one all-gather, read-back."""
    import torch
    out = {"samples_total": STRONG_TOTAL, "scaling": "strong",
           "measured_on": "%d GPU(s)%s" % (world, "" if world > 1 else ": one GPU runs one shard of each N")}
    for name in ("config2", "config3"):
        rows = {}
        for n in ([world] if world > 1 else [1, 2, 4, 8]):
            shard = -(-STRONG_TOTAL // n)
            torch.cuda.empty_cache()
            E = Workload(name, shard, args, dev_index, rank, world)
            r = E.measure(20, 3)
            E.close()
            del E
            k = r["kernels"]
            rows["n%d" % n] = {"n_gpus": n, "samples_per_gpu": shard, "ms_per_job": r["ms_per_step"],
                               "value": STRONG_TOTAL / (r["ms_per_step"] / 1e3), "unit": "samples/s",
                               "measured_on": "%d GPU(s)" % world,
                               "kernels_ms": {"k_rng": k["k_rng_ms"], "k_place": k["k_place_ms"], "k_merge": k["k_merge_ms"],
                                              "k_tail": k["k_tail_ms"], "k_contig": k["k_contig_ms"], "count": k["count_phase_ms"]},
                               "allgather_ms": (r["allgather"] or {}).get("avg_ms")}
        out[name] = rows
    if world == 1:
        out["note"] = ("one GPU running a shard of STRONG_TOTAL / N samples: what N GPUs would each do, without the all-gather; "
                       "the floor at small shards is k_place's serial chain over the longest unit's tile (DESIGN.md)")
    return out


def api_block(args, repeats=5):
    """gat_amd.run() -- the reference's gat.run() seam (gat/__init__.py:855-1088) -- end to end on config2 and config3 with
    10 000 samples: observed counts, problem creation (inputs cross PCIe), sampling + counting, null-distribution
    statistics, the 24-column result rows.  Wall clock of the call, median of `repeats` after one warm-up call."""
    import gc
    import gat_amd
    from gat_amd import synthetic
    out = {"seam": "gat_amd.run(segments, annotations, workspace, sampler, counters, workspace_generator, num_samples=10000)"}
    for name in ("config2", "config3"):
        cfg = synthetic.config(name)
        # building the collections from the configuration's arrays, the isochore split included (IO.applyIsochores, gat/IO.py:
        # 188-293 -> IntervalCollection.toIsochores: one call of the library for all lists): median of three after a first
        # build (which loads the library and starts its host threads)
        synthetic.as_collections(cfg)
        built = []
        for _ in range(3):
            t0 = time.perf_counter()
            segments, annotations, workspace, t_iso = synthetic.as_collections(cfg)
            built.append((time.perf_counter() - t0, t_iso))
        built.sort()
        t_inputs, t_iso = built[1]
        counters = [gat_amd.COUNTERS[cfg["counter"]]()]

        def call():
            t = time.perf_counter()
            rows = gat_amd.run(segments, annotations, workspace, gat_amd.SamplerAnnotator(bucket_size=1, nbuckets=100000), counters,
                               gat_amd.UnconditionalWorkspace(), num_samples=10000, random_seed=args.seed)
            return time.perf_counter() - t, len(rows)
        call()
        gc.collect()
        gc.disable()
        try:
            ts = sorted(call()[0] for _ in range(repeats))
        finally:
            gc.enable()
        n_rows = call()[1]
        med = ts[len(ts) // 2]
        out[name] = {"ms_per_run": med * 1e3, "min_ms": ts[0] * 1e3, "max_ms": ts[-1] * 1e3, "repeats": repeats,
                     "samples_per_s": 10000 / med, "result_rows": n_rows, "num_samples": 10000,
                     "inputs_ms": t_inputs * 1e3, "isochore_split_ms": t_iso * 1e3, "ms_with_inputs": (med + t_inputs) * 1e3,
                     "includes": "computeCounts (observed), gat_problem_create, gat_sample_and_count, gat_null_stats, D2H of the "
                                 "count matrix, AnnotatorResultExtended rows; ms_with_inputs adds building the collections "
                                 "from the arrays incl. the isochore split (inputs_ms)"}
    # sixteen segment tracks against config 3's annotations (gat/__init__.py:971-1010 loops the tracks): the annotation tables
    # are made once (gat_annotations_create) and shared, a track's sampling is enqueued while the previous one's rows are made
    cfg = synthetic.config("config3")
    cfg16 = dict(cfg, segment_tracks=[("track%02d" % i, synthetic.random_segments(synthetic.HG19, 10000, 500, 11 + i)) for i in range(16)])
    segments, annotations, workspace, _ = synthetic.as_collections(cfg16)
    counters = [gat_amd.COUNTERS[cfg["counter"]]()]

    def call16():
        t = time.perf_counter()
        rows = gat_amd.run(segments, annotations, workspace, gat_amd.SamplerAnnotator(bucket_size=1, nbuckets=100000), counters,
                           gat_amd.UnconditionalWorkspace(), num_samples=10000, random_seed=args.seed)
        return time.perf_counter() - t, len(rows)
    call16()
    gc.collect()
    gc.disable()
    try:
        ts = sorted(call16()[0] for _ in range(3))
    finally:
        gc.enable()
    n_rows = call16()[1]
    # what a later track's problem costs to create when the annotation tables exist (and what the first one costs with them)
    from gat_amd import _lib, problem
    ctx = gat_amd.get_context()
    flat = problem.flatten_dictionaries(segments["track00"], workspace, annotations, list(annotations.tracks), 1, 100000)
    t0 = time.perf_counter()
    A = _lib.Annotations(ctx, flat)
    t_annos = time.perf_counter() - t0
    units = dict(flat, annos=None, anno_off=None, anno_end=None, anno_group=None)
    tc = []
    for _ in range(5):
        t0 = time.perf_counter()
        P = _lib.Problem(ctx, units, annotations=A)
        tc.append(time.perf_counter() - t0)
        P.close()
    A.close()
    med = ts[len(ts) // 2]
    out["config3_16_segment_tracks"] = {"ms_per_run": med * 1e3, "ms_per_track": med * 1e3 / 16, "repeats": 3, "result_rows": n_rows,
                                        "samples_per_s": 16 * 10000 / med, "num_samples": 10000, "segment_tracks": 16,
                                        "annotation_tables_once_ms": t_annos * 1e3,
                                        "problem_create_with_shared_tables_ms": sorted(tc)[len(tc) // 2] * 1e3}
    return out


def headline(main_out, steps):
    """`value` / `ms_per_step` of the line: the K steps timed right behind the W warm-up steps -- unless they are a blink (the
    driver's 20 steps of config 2 are 50 ms) AND the same step repeated for a second or more says something else by more than
    2 %: then the longer measurement is the number to quote ("the number to quote is 3.8-4.0 M", not the 4.02 M of
    50 ms).  The K steps' own figure stays in the line as k_steps_value, value_source says which it is."""
    k_seconds = main_out["ms_per_step"] * steps / 1e3
    out = {"value": main_out["value"], "ms_per_step": main_out["ms_per_step"],
           "k_steps_value": main_out["value"], "k_steps_ms_per_step": main_out["ms_per_step"],
           "value_source": "the %d timed steps (%.3f s)" % (steps, k_seconds)}
    sus = main_out.get("sustained")
    if sus:
        out["sustained_value"] = sus["value"]
        if k_seconds < 0.5 and sus["seconds"] >= 0.5 and abs(main_out["value"] - sus["value"]) > 0.02 * sus["value"]:
            out["value"], out["ms_per_step"] = sus["value"], sus["ms_per_step"]
            out["value_source"] = ("sustained: %d steps in %.2f s (the %d steps behind the warm-up took %.3f s and read %+.1f %%)"
                                   % (sus["steps"], sus["seconds"], steps, k_seconds, 100.0 * (main_out["value"] / sus["value"] - 1.0)))
    return out


LINE_LIMIT = 3000               # bytes of the one stdout line (the driver parses stdout; round 4's 21 KB line broke it)
LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data", "config", "roofline", "cpu_baseline")


def _r(x, nd=4):
    """numbers of the line, rounded to what they are good for"""
    if isinstance(x, float):
        x = float("%.*g" % (nd + 3, x))
        return int(x) if x == int(x) and abs(x) >= 1e6 else x
    return x


def _cut(x, n=120):
    return x if not isinstance(x, str) or len(x) <= n else x[:n - 3] + "..."


def final_line(out, details_path=None):
    """the ONE stdout line: the contract's keys (+ roofline, cpu_baseline) and a handful of small extras, every string at
    most 120 characters, the whole line below LINE_LIMIT.  Everything else (`configs`, `strong_scaling`, `api`, `kernels`,
    `sampler`, the long notes) lives in the details file only."""
    line = dict((k, out.get(k)) for k in LINE_KEYS[:12])
    for k in ("value", "ms_per_step"):
        line[k] = _r(line[k])
    cfg = out.get("config") or {}
    line["config"] = {"workload": _cut(cfg.get("workload", "")), "samples_per_step_per_gpu": cfg.get("samples_per_step_per_gpu"),
                      "sharding": _cut(cfg.get("sharding", "")), "steps_in_flight": cfg.get("steps_in_flight", 1)}
    roof = out.get("roofline") or {}
    line["roofline"] = dict((k, _cut(_r(roof.get(k)))) for k in
                            ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_ms",
                             "algorithmic_bytes_per_launch", "traffic_source") if k in roof)
    if "cpu_baseline" in out:
        cb = out["cpu_baseline"]
        line["cpu_baseline"] = dict((k, _cut(_r(cb.get(k)))) for k in ("value", "unit", "cores", "kind", "sample",
                                                                       "single_thread_value") if k in cb)
    # small extras, dropped from the back if the line would not fit
    extras = []
    if out.get("value_source"):
        extras.append(("value_source", _cut(out["value_source"])))
        extras.append(("k_steps_value", _r(out.get("k_steps_value"))))
    if out.get("sustained"):
        extras.append(("sustained_value", _r(out["sustained"]["value"])))
        if "ms_per_step" in out["sustained"]:
            extras.append(("sustained_ms_per_step", _r(out["sustained"]["ms_per_step"])))
    if out.get("step"):
        extras.append(("step", {"hbm_bytes": out["step"]["hbm_bytes"], "hbm_frac": _r(out["step"]["hbm_frac"])}))
    if out.get("distributed"):
        d = out["distributed"]
        extras.append(("distributed", {"backend": d["backend"], "world_size": d["world_size"], "one_gpu_per_rank": d["one_gpu_per_rank"],
                                       "ranks_in_collective": d.get("ranks_in_collective")}))
    if out.get("allgather"):
        a = out["allgather"]
        extras.append(("allgather", {"avg_ms": _r(a["avg_ms"]), "bytes_per_rank": a["bytes_per_rank"]}))
    if out.get("strong_scaling"):
        st = out["strong_scaling"]
        extras.append(("strong_scaling", dict(
            [("samples_total", st["samples_total"]), ("measured_on", st.get("measured_on"))] +
            [(name, dict((n, _r(row["ms_per_job"])) for n, row in st[name].items())) for name in ("config2", "config3") if name in st])))
    if out.get("configs"):
        def small(name, r):
            d = {"value": _r(r["value"]), "ms_per_step": _r(r["ms_per_step"]), "samples_per_step": r["config"]["samples_per_step_per_gpu"],
                 "roofline_bound": r["roofline"]["bound"], "roofline_frac": _r(r["roofline"]["frac"])}
            if name == "config3":                                # the north_star target shape: its kernel and the port beside it
                d["kernel"] = _cut(r["roofline"].get("kernel", ""), 40)
                if r.get("cpu_baseline"):
                    d["cpu_baseline"] = {"value": _r(r["cpu_baseline"]["value"]), "cores": r["cpu_baseline"]["cores"], "kind": r["cpu_baseline"]["kind"]}
            if r.get("reference_published"):
                d["reference_published"] = r["reference_published"]["value"]
            return d
        extras.append(("configs", dict((name, small(name, r)) for name, r in out["configs"].items())))
    if out.get("api"):
        extras.append(("api_ms_per_run", dict((name, _r(v["ms_per_run"])) for name, v in out["api"].items() if isinstance(v, dict))))
    if details_path:
        extras.append(("details", _cut(details_path)))
    for k, v in extras:
        line[k] = v
    text = json.dumps(line)
    while len(text) >= LINE_LIMIT and extras:
        k, _ = extras.pop()
        del line[k]
        text = json.dumps(line)
    assert len(text) < LINE_LIMIT, "bench.py: the stdout line does not fit %d bytes" % LINE_LIMIT
    return text


class Guard(object):
    """N > 1: whatever happens, rank 0 leaves ONE JSON line and the job ends -- never a hang, never a run without a line.  A rank
    that fails takes the job down (the launcher terminates the others: SIGTERM; a collective that waits for a dead rank ends at
    the process group's timeout); rank 0 then prints what it knows -- world size, backend, the devices the ranks reported,
    RCCL's own count of the ranks -- with "error" and value null, and exits non-zero."""

    def __init__(self, args, rank, world):
        self.args, self.rank, self.world = args, rank, world
        self.phase, self.dist, self.printed = "start", None, False

    def line(self, reason):
        return json.dumps({"metric": METRIC, "value": None, "unit": "samples/s", "n_gpus": self.world, "steps": self.args.steps,
                           "warmup": self.args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                           "vs_baseline": None, "dtype": "u32", "data": "synthetic", "config": {"workload": self.args.config},
                           "error": _cut(str(reason), 300), "phase": self.phase, "distributed": self.dist})

    def emergency(self, reason):
        if self.rank == 0 and not self.printed:
            self.printed = True
            sys.stdout.flush()
            print(self.line(reason), flush=True)

    def arm(self, deadline_s):
        import signal
        import threading

        def on_term(signum, frame):
            self.emergency("signal %d in phase %r (another rank failed, or the launcher gave up)" % (signum, self.phase))
            os._exit(128 + signum)
        signal.signal(signal.SIGTERM, on_term)
        self.done = threading.Event()

        def watch():
            if not self.done.wait(deadline_s):
                self.emergency("deadline of %d s passed in phase %r" % (deadline_s, self.phase))
                os._exit(3)
        threading.Thread(target=watch, daemon=True).start()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    guard = Guard(args, rank, world)
    if world > 1:
        guard.arm(int(os.environ.get("GAT_BENCH_DEADLINE", "3000")))
    try:
        run(args, rank, world, guard)
    except SystemExit as e:
        if e.code not in (0, None):
            guard.emergency(e.code)
        raise
    except BaseException as e:                                    # noqa: BLE001  (the line first, then the traceback)
        guard.emergency("%s: %s" % (type(e).__name__, e))
        raise
    finally:
        if world > 1 and hasattr(guard, "done"):
            guard.done.set()


def run(args, rank, world, guard):
    import datetime
    import torch
    import torch.distributed as dist
    from gat_amd import synthetic

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # one rank per GPU; GAT_BENCH_SHARE_GPU=1 (testing the N>1 code path on a one-GPU box) lets ranks share devices
    # and swaps RCCL, which refuses two ranks on one device, for gloo
    share = os.environ.get("GAT_BENCH_SHARE_GPU") == "1"
    dev_index = local_rank % torch.cuda.device_count() if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        guard.phase = "init_process_group"
        # (a collective that waits for a rank that is gone ends here, not after the default half hour)
        timeout = datetime.timedelta(seconds=int(os.environ.get("GAT_BENCH_DIST_TIMEOUT", "600")))
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timeout)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=timeout)
        assert dist.get_world_size() == args.gpus
        # who is there, known from the start: the devices the ranks sit on, and the collective's own count of the ranks (an
        # all-reduce of ones over the backend the run uses)
        guard.phase = "roll call"
        devs = [None] * world
        dist.all_gather_object(devs, (socket.gethostname(), torch.cuda.current_device(),
                                      str(getattr(torch.cuda.get_device_properties(dev_index), "uuid", dev_index))))
        ones = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(ones)
        guard.dist = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_in_collective": int(ones.item()),
                      "devices": [list(d) for d in devs], "one_gpu_per_rank": len(set(devs)) == world}
        if not share and not guard.dist["one_gpu_per_rank"]:
            raise SystemExit("bench.py: %d ranks landed on %d devices: %r" % (world, len(set(devs)), devs))
        if os.environ.get("GAT_BENCH_FAIL_RANK") == str(rank):        # (tests: a rank that dies behind the roll call)
            raise RuntimeError("rank %d fails on purpose (GAT_BENCH_FAIL_RANK)" % rank)
    guard.phase = "headline shape"

    cfg_samples = synthetic.CONFIG_SAMPLES[args.config]
    W = Workload(args.config, args.samples or cfg_samples, args, dev_index, rank, world, depth=args.pipeline)
    main_out = W.measure(args.steps, args.warmup, args.sustain_seconds)
    out = {"metric": METRIC, "value": main_out["value"], "unit": "samples/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": main_out["ms_per_step"], "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "u32", "data": "synthetic"}
    for k in ("config", "roofline", "kernels", "sampler", "allgather", "sustained", "step"):
        if k in main_out:
            out[k] = main_out[k]
    out.update(headline(main_out, args.steps))
    if world > 1:
        out["distributed"] = guard.dist
    if args.dump_counts and rank == 0:
        import numpy as np
        torch.cuda.synchronize()
        np.savez(args.dump_counts, counts=W.host[W.last].numpy(), samples_per_rank=W.S, world=world, seed=args.seed,
                 first_sample=W.last_first)
    cpu_leg = rank == 0 and not args.no_cpu_baseline and world == 1      # reported on rank 0 at N=1 only
    if cpu_leg:
        out["cpu_baseline"] = cpu_baseline(W.flat, W.counters, args.seed, args.cpu_seconds)
        if args.config in REFERENCE_CYTHON:
            out["cpu_baseline"]["reference_cython_engine"] = {
                "value": REFERENCE_CYTHON[args.config], "unit": "samples/s", "cores": 1,
                "where": "the reference's Cython engine itself, build container (BASELINE.md section 2); it cannot "
                         "travel to the GPU box, hence kind = port above"}
    W.close()
    del W
    extras = {}
    for name in [x for x in args.extra.split(",") if x and x != args.config]:
        guard.phase = "extra shape " + name
        torch.cuda.empty_cache()
        # (measured like the headline shape: these steps are 6-70 ms and gain 0-1 % from the second step in flight)
        E = Workload(name, EXTRA_SAMPLES.get(name, synthetic.CONFIG_SAMPLES[name]), args, dev_index, rank, world,
                     depth=args.pipeline)
        # (a config-4 step is a rank's whole shard, about 0.1 s: a quarter of the steps)
        r = E.measure(max(1, args.extra_steps // 4 if name == "config4" else args.extra_steps), 2, args.sustain_seconds)
        r["n_gpus"] = world
        if cpu_leg and name == "config3":                   # the north_star target shape: the port on the same workload
            r["cpu_baseline"] = cpu_baseline(E.flat, E.counters, args.seed, args.cpu_seconds * 0.6)
        if name == "refdata":
            # the one workload the reference publishes a timing for: its own log of this data set
            r["reference_published"] = {"value": 30.3, "unit": "samples/s", "cores": 1,
                                        "source": "test/data/output_single.tsv:66-77 of the reference: 1000 samples of this track in 33.0 s "
                                                  "(cgat150, Linux 2.6.32, Python 2.7.1, 2013); BASELINE.md"}
            if cpu_leg:
                r["cpu_baseline"] = cpu_baseline(E.flat, E.counters, args.seed, args.cpu_seconds * 0.4)
        extras[name] = r
        E.close()
        del E
    if extras:
        out["configs"] = extras
    if not args.no_strong and args.scale == 1.0:
        guard.phase = "strong scaling"
        out["strong_scaling"] = strong_scaling(args, dev_index, rank, world)
    if not args.no_api and world == 1 and args.scale == 1.0:
        out["api"] = api_block(args)
    if rank == 0:
        # the bulk goes to a file (and nowhere near stdout); stdout carries ONE small line, the last thing printed
        details = args.details or os.path.join(ROOT, "bench_details.json")
        try:
            with open(details, "w") as f:
                json.dump(out, f, indent=1)
                f.write("\n")
        except OSError as e:
            sys.stderr.write("bench.py: could not write %s: %s\n" % (details, e))
            details = None
        sys.stdout.flush()
        guard.printed = True
        print(final_line(out, os.path.relpath(details, ROOT) if details else None), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

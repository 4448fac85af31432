"""gat_amd.run() under torch.distributed on the GPU: two ranks (gloo, both on this box's one GPU) shard the
samples, all-gather the count matrix and must print the rows a single process prints; the seed is agreed on by
broadcast when none is given; pattern files have one writer."""
import os

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _init(backend, rank, world, path, **kw):
    """rendezvous over a file in the test's own directory: no TCP port to find free (a port handed out by bind(0) and closed
    again was taken by the time the store wanted it: EADDRINUSE in the GPU suite, round 6)"""
    import torch.distributed as dist
    dist.init_process_group(backend, init_method="file://" + os.path.join(path, "rendezvous"), rank=rank, world_size=world, **kw)


def _collections():
    import gat_amd
    from gat_amd import synthetic
    _, cfg = synthetic.small_genome()

    def coll(tracks):
        c = gat_amd.IntervalCollection()
        for t, per in tracks:
            for contig, a in per.items():
                s = gat_amd.SegmentList(array=a)
                s.isNormalized = 1
                c.add(t, contig, s)
        return c

    segments = coll([("merged", cfg["segments"])])
    annotations = coll(cfg["annotations"])
    workspaces = coll([("ws", cfg["workspace"])])
    workspaces.collapse()
    workspaces.restrict("collapsed")
    isochores = coll(list(cfg["isochores"].items()))
    isochores.intersect(workspaces["collapsed"])
    workspaces.toIsochores(isochores, truncate=True)
    annotations.toIsochores(isochores, truncate=True)
    segments.toIsochores(isochores, truncate=False)
    return segments, annotations, workspaces["collapsed"]


def _run(num_samples, seed, counts_pattern=None):
    import gat_amd
    segments, annotations, workspace = _collections()
    counters = [gat_amd.CounterNucleotideOverlap(), gat_amd.CounterNucleotideDensity(), gat_amd.CounterSegmentOverlap()]
    kw = dict(num_samples=num_samples, output_counts_pattern=counts_pattern)
    if seed is not None:
        kw["random_seed"] = seed
    return [str(r) for r in gat_amd.run(segments, annotations, workspace, gat_amd.SamplerAnnotator(bucket_size=0), counters,
                                        gat_amd.UnconditionalWorkspace(), **kw)]


def _worker(rank, world, path, seed):
    import torch.distributed as dist
    os.environ["LOCAL_RANK"] = "0"                     # one GPU on this box: both ranks drive device 0
    _init("gloo", rank, world, path)
    np.random.seed(100 + rank)                         # different global RNG states: the base seed must be rank 0's
    rows = _run(37, seed, os.path.join(path, "counts_%s.tsv"))
    with open(os.path.join(path, "rows%d.txt" % rank), "w") as f:
        f.write("\n".join(rows))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("seed", [5, None])
def test_run_two_ranks_equals_single_process(tmp_path, seed):
    mp.spawn(_worker, args=(2, str(tmp_path), seed), nprocs=2, join=True)
    r0 = open(str(tmp_path / "rows0.txt")).read().split("\n")
    r1 = open(str(tmp_path / "rows1.txt")).read().split("\n")
    assert r0 == r1 and len(r0) == 9                    # 3 counters x 3 annotation tracks
    if seed is not None:
        assert r0 == _run(37, seed)
    else:
        np.random.seed(100)                             # rank 0's state: the broadcast seed
        assert r0 == _run(37, None)
    lines = open(str(tmp_path / "counts_nucleotide-overlap.tsv")).read().split("\n")
    assert lines[0] == "track\tannotation\tobserved\tcounts" and len(lines[1].split("\t")[3].split(",")) == 37


def test_c_abi_allgather_counts_single_rank():
    """gat_comm_* / gat_allgather_counts (RCCL loaded by the library itself, no torch): a communicator of one rank on this
    box's GPU gathers a count block onto itself."""
    import torch  # noqa: F401  (torch brings its own RCCL: torch/lib/librccl.so)
    from gat_amd import _lib
    ctx = _lib.Context(0)
    comm = _lib.Comm(ctx, 1, 0, _lib.comm_unique_id())
    # a process never holds two RCCLs: with torch's copy mapped, the library takes that one (RTLD_NOLOAD), it loads none
    if any("librccl" in line for line in open("/proc/self/maps")):
        assert _lib.lib().gat_comm_library_preloaded() == 1
        assert len(set(line.split()[-1] for line in open("/proc/self/maps") if "librccl" in line)) == 1
    n = 3 * 5 * 7
    src = np.arange(n, dtype=np.int64) * 3 - 11
    a, b = ctx.alloc(n * 8), ctx.alloc(n * 8)
    try:
        _lib._check(_lib.lib().gat_memcpy_h2d(ctx._h, a, src.ctypes.data, n * 8), ctx._h)
        comm.allgather_counts(a, b, n)
        out = np.zeros(n, dtype=np.int64)
        ctx.d2h(out, b)
        assert np.array_equal(out, src)
    finally:
        ctx.free(a)
        ctx.free(b)
        comm.close()
        ctx.close()


def _worker_nccl_one_rank(rank, world, path, device_stats):
    import torch
    import torch.distributed as dist
    os.environ["LOCAL_RANK"] = "0"
    os.environ["GAT_DEVICE_STATS"] = "1" if device_stats else "0"
    os.environ["GAT_FORCE_COLLECTIVE_PATH"] = "1"
    torch.cuda.set_device(0)
    _init("nccl", 0, 1, path, device_id=torch.device("cuda", 0))
    import gat_amd
    segments, annotations, workspace = _collections()
    counters = [gat_amd.CounterNucleotideOverlap(), gat_amd.CounterNucleotideDensity(), gat_amd.CounterSegmentOverlap()]
    res = gat_amd.run(segments, annotations, workspace, gat_amd.SamplerAnnotator(bucket_size=0), counters,
                      gat_amd.UnconditionalWorkspace(), num_samples=37, random_seed=5,
                      output_counts_pattern=os.path.join(path, "fcounts_%s.tsv"))
    with open(os.path.join(path, "forced_rows.txt"), "w") as f:
        f.write("\n".join(str(r) for r in res))
    np.save(os.path.join(path, "forced_samples.npy"), np.array([r.samples for r in res]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("device_stats", [True, False])
def test_the_collective_path_over_rccl_with_one_rank(tmp_path, device_stats):
    """the code run() takes under the nccl backend -- the library's stream ordered against torch's by events, equal shards,
    the all-gather of device memory over RCCL, statistics from the gathered matrix, rows that read it back on demand -- with a
    process group of ONE rank (RCCL refuses two ranks on this box's one GPU): same rows, same samples, same count files."""
    mp.spawn(_worker_nccl_one_rank, args=(1, str(tmp_path), device_stats), nprocs=1, join=True)
    rows = open(str(tmp_path / "forced_rows.txt")).read().split("\n")
    os.environ["GAT_DEVICE_STATS"] = "0"
    try:
        assert rows == _run(37, 5, os.path.join(str(tmp_path), "pcounts_%s.tsv"))
    finally:
        del os.environ["GAT_DEVICE_STATS"]
    assert np.load(str(tmp_path / "forced_samples.npy")).shape == (9, 37)
    for name in ("nucleotide-overlap", "nucleotide-density", "segment-overlap"):
        assert open(str(tmp_path / ("fcounts_%s.tsv" % name))).read() == open(str(tmp_path / ("pcounts_%s.tsv" % name))).read()


# ------------------------------------------------------------------------------------------------------------------
# RCCL with more than one rank: these run the moment a box shows two GPUs (the boxes this suite usually sees have one,
# where RCCL refuses two ranks on a device and the tests above stand in with gloo).
def _n_gpus():
    import torch
    return torch.cuda.device_count()


needs_two_gpus = pytest.mark.skipif(_n_gpus() < 2, reason="RCCL with two ranks needs two GPUs")


def _worker_c_abi(rank, world, uid, path):
    from gat_amd import _lib
    ctx = _lib.Context(rank)                              # one GPU per rank
    comm = _lib.Comm(ctx, world, rank, uid)
    n = 2 * 3 * 11                                        # [counter][track][shard] slots of this rank
    src = (np.arange(n, dtype=np.int64) + 1) * (rank + 1) * 7 - 5
    a, b = ctx.alloc(n * 8), ctx.alloc(world * n * 8)
    try:
        _lib._check(_lib.lib().gat_memcpy_h2d(ctx._h, a, src.ctypes.data, n * 8), ctx._h)
        comm.allgather_counts(a, b, n)
        out = np.zeros(world * n, dtype=np.int64)
        ctx.d2h(out, b)
        np.save(os.path.join(path, "abi%d.npy" % rank), out)
    finally:
        ctx.free(a)
        ctx.free(b)
        comm.close()
        ctx.close()


@needs_two_gpus
def test_c_abi_allgather_counts_two_ranks(tmp_path):
    """gat_comm_create + gat_allgather_counts over RCCL, two ranks on two GPUs, no torch.distributed: every rank ends up with
    rank r's block at r * n_slots."""
    from gat_amd import _lib
    uid = _lib.comm_unique_id()                           # (rank 0's call in a real host; the bytes travel by argument here)
    mp.spawn(_worker_c_abi, args=(2, uid, str(tmp_path)), nprocs=2, join=True)
    n = 2 * 3 * 11
    want = np.concatenate([(np.arange(n, dtype=np.int64) + 1) * (r + 1) * 7 - 5 for r in range(2)])
    for r in range(2):
        assert np.array_equal(np.load(str(tmp_path / ("abi%d.npy" % r))), want), r


def _worker_nccl(rank, world, path, seed, device_stats, num_samples=37):
    import torch
    import torch.distributed as dist
    os.environ["LOCAL_RANK"] = str(rank)
    os.environ["GAT_DEVICE_STATS"] = "1" if device_stats else "0"
    torch.cuda.set_device(rank)
    _init("nccl", rank, world, path, device_id=torch.device("cuda", rank))
    rows = _run(num_samples, seed)
    with open(os.path.join(path, "nccl_rows%d.txt" % rank), "w") as f:
        f.write("\n".join(rows))
    dist.barrier()
    dist.destroy_process_group()


@needs_two_gpus
@pytest.mark.parametrize("device_stats,num_samples", [(True, 37), (False, 37), (True, 1)])
def test_run_under_nccl_equals_single_process(tmp_path, device_stats, num_samples):
    """gat_amd.run() under the nccl backend (= RCCL): two ranks, one GPU each, shard the samples, all-gather the device
    matrix, take the statistics from it (on the device, or with numpy) and print the rows one process prints.  One sample
    over two ranks: the second rank's shard lies beyond the job (every rank computes a full shard; the surplus is cut off)."""
    mp.spawn(_worker_nccl, args=(2, str(tmp_path), 5, device_stats, num_samples), nprocs=2, join=True)
    r0 = open(str(tmp_path / "nccl_rows0.txt")).read().split("\n")
    r1 = open(str(tmp_path / "nccl_rows1.txt")).read().split("\n")
    assert r0 == r1 and len(r0) == 9
    os.environ["GAT_DEVICE_STATS"] = "0"
    try:
        assert r0 == _run(num_samples, 5)
    finally:
        del os.environ["GAT_DEVICE_STATS"]


@needs_two_gpus
def test_bench_two_gpus_over_rccl(tmp_path):
    """bench.py --gpus 2 as the driver launches it: backend nccl, one GPU per rank, and the gathered matrix of the last
    step equal to the oracle's columns for those sample ids."""
    import json
    import subprocess
    import sys
    from gat_amd import problem, synthetic
    from oracle import oracle as O
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict((k, v) for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                                    "GAT_BENCH_SHARE_GPU"))
    dump = str(tmp_path / "counts.npz")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--samples", "64",
           "--extra", "", "--no-strong", "--sustain-seconds", "0", "--dump-counts", dump, "--details", str(tmp_path / "d.json")]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    last = [l for l in r.stdout.splitlines() if l.strip()][-1]
    assert len(last) < 4096 and json.loads(last)["distributed"]["backend"] == "nccl"
    out = json.load(open(str(tmp_path / "d.json")))
    assert out["distributed"]["backend"] == "nccl" and out["distributed"]["world_size"] == 2
    assert out["distributed"]["one_gpu_per_rank"] and len(set(map(tuple, out["distributed"]["devices"]))) == 2
    z = np.load(dump)
    cfg = synthetic.config("config2")
    flat = problem.flatten_arrays(cfg["segments"], cfg["annotations"], cfg["workspace"], cfg["isochores"])
    first, S = int(z["first_sample"]), int(z["samples_per_rank"])
    want, _ = O.run_samples(flat, ["nucleotide-overlap"], int(z["seed"]), 1, first, first + 2 * S)
    got = z["counts"]                                     # [rank * K + k][track][sample of the rank's shard]
    for rk in range(2):
        assert np.array_equal(got[rk], want[0][:, rk * S:(rk + 1) * S]), rk

"""gat_amd.run() under torch.distributed on the GPU: two ranks (gloo, both on this box's one GPU) shard the
samples, all-gather the count matrix and must print the rows a single process prints; the seed is agreed on by
broadcast when none is given; pattern files have one writer."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


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


def _worker(rank, world, port, path, seed):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["LOCAL_RANK"] = "0"                     # one GPU on this box: both ranks drive device 0
    dist.init_process_group("gloo", rank=rank, world_size=world)
    np.random.seed(100 + rank)                         # different global RNG states: the base seed must be rank 0's
    rows = _run(37, seed, os.path.join(path, "counts_%s.tsv"))
    with open(os.path.join(path, "rows%d.txt" % rank), "w") as f:
        f.write("\n".join(rows))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("seed", [5, None])
def test_run_two_ranks_equals_single_process(tmp_path, seed):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), seed), nprocs=2, join=True)
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
    from gat_amd import _lib
    ctx = _lib.Context(0)
    comm = _lib.Comm(ctx, 1, 0, _lib.comm_unique_id())
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

"""N > 1 path on CPU: two gloo ranks each compute their contiguous sample shard (the CPU oracle
stands in for the GPU here -- tests may use it) and one all-gather reassembles the count matrix;
it must equal the single-process matrix column for column."""
import os

import numpy as np
import torch.multiprocessing as mp

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _init(backend, rank, world, path, **kw):
    """rendezvous over a file in the test's own directory: no TCP port to find free (a port handed out by bind(0) and closed
    again was taken by the time the store wanted it: EADDRINUSE in the GPU suite, round 6)"""
    import torch.distributed as dist
    dist.init_process_group(backend, init_method="file://" + os.path.join(path, "rendezvous"), rank=rank, world_size=world, **kw)


def _worker(rank, world, S, path):
    import torch.distributed as dist
    from gat_amd import distributed
    from oracle import oracle as O
    _init("gloo", rank, world, path)
    z = np.load(os.path.join(G, "run_small_isochores.npz"))
    flat = {k: z[k] for k in z.files}
    counters = ["nucleotide-overlap", "nucleotide-density"]
    begin, end = distributed.shard_range(S, rank, world)
    local, _ = O.run_samples(flat, counters, 11, 1, begin, end)
    per = distributed.padded_shard(S, world)
    stack = np.zeros((2, flat["n_tracks"], per), dtype=np.int64)
    for k in range(2):
        stack[k, :, :end - begin] = local[k].view(np.int64)
    full = distributed.gather_numpy(stack, S)
    np.save(os.path.join(path, "rank%d.npy" % rank), full)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_allgather(tmp_path):
    from oracle import oracle as O
    S = 37                       # odd on purpose: ragged last shard
    mp.spawn(_worker, args=(2, S, str(tmp_path)), nprocs=2, join=True)
    z = np.load(os.path.join(G, "run_small_isochores.npz"))
    flat = {k: z[k] for k in z.files}
    want, _ = O.run_samples(flat, ["nucleotide-overlap", "nucleotide-density"], 11, 1, 0, S)
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        assert np.array_equal(got[0], want[0])
        assert np.array_equal(got[1].view(np.float64), want[1])
    # and the goldens: same columns as the reference produced for these sample ids
    assert np.array_equal(want[0].astype(np.float64), z["counts_mode1"][0][:, :S])


def test_shard_ranges_cover_everything():
    from gat_amd import distributed
    for n in (0, 1, 7, 8, 9, 10000):
        for w in (1, 2, 3, 8):
            r = [distributed.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert all(e - b <= distributed.padded_shard(n, w) for b, e in r)


def _worker_ragged(rank, world, S, path):
    import torch.distributed as dist
    from gat_amd import distributed
    _init("gloo", rank, world, path)
    begin, end = distributed.shard_range(S, rank, world)
    per = distributed.padded_shard(S, world)
    # slot (k, a, s) of the whole matrix holds a number that names it; a rank fills its own columns
    stack = np.zeros((2, 3, per), dtype=np.int64)
    for k in range(2):
        for a in range(3):
            stack[k, a, :end - begin] = (k * 3 + a) * 1000 + np.arange(begin, end)
    full = distributed.gather_numpy(stack, S)
    np.save(os.path.join(path, "ragged%d.npy" % rank), full)
    dist.barrier()
    dist.destroy_process_group()


def test_eight_ranks_ragged_shards(tmp_path):
    """the node's shape -- eight ranks -- with 37 samples: shards of 5, the last of 2, ranks past the end of nothing; every rank
    must end up with every column in sample order."""
    S, world = 37, 8
    mp.spawn(_worker_ragged, args=(world, S, str(tmp_path)), nprocs=world, join=True)
    want = np.zeros((2, 3, S), dtype=np.int64)
    for k in range(2):
        for a in range(3):
            want[k, a] = (k * 3 + a) * 1000 + np.arange(S)
    for r in range(world):
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "ragged%d.npy" % r)), want), r


def _worker_shard_and_gather(rank, world, path):
    """distributed.shard_and_gather -- the arithmetic of gat_amd.run() under the nccl backend (_nccl_shard_and_gather): equal
    shards of ceil(S / world), the ids at or beyond S never drawn, a rank past the end filling nothing, ONE all-gather, the
    surplus cut off -- with a fill function that names every slot it writes; every rank must end up with the matrix one process
    would have filled."""
    import torch
    from gat_amd import distributed
    _init("gloo", rank, world, path)
    K, A = 2, 3
    out = {}
    for S in (1, 7, 8, 9, 10000):
        asked = []

        def fill(lo, hi, block):
            asked.append((lo, hi))
            for k in range(K):
                for a in range(A):
                    block[k, a] = (k * A + a) * 100000 + torch.arange(lo, hi, dtype=torch.int64)
        full = distributed.shard_and_gather(fill, K, A, S, torch.device("cpu"))
        per = -(-S // world)
        lo, hi = min(S, rank * per), min(S, (rank + 1) * per)
        assert asked == ([(lo, hi)] if hi > lo else []), (S, rank, asked)       # never a sample id beyond the job
        out[str(S)] = full.numpy()
    np.savez(os.path.join(path, "sg%d.npz" % rank), **out)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def _shard_and_gather_case(tmp_path, world):
    mp.spawn(_worker_shard_and_gather, args=(world, str(tmp_path)), nprocs=world, join=True)
    for S in (1, 7, 8, 9, 10000):
        want = np.zeros((2, 3, S), dtype=np.int64)
        for k in range(2):
            for a in range(3):
                want[k, a] = (k * 3 + a) * 100000 + np.arange(S)
        for r in range(world):
            got = np.load(os.path.join(str(tmp_path), "sg%d.npz" % r))[str(S)]
            assert got.shape == want.shape and np.array_equal(got, want), (S, world, r)


def test_shard_and_gather_two_ranks(tmp_path):
    _shard_and_gather_case(tmp_path, 2)


def test_shard_and_gather_eight_ranks(tmp_path):
    _shard_and_gather_case(tmp_path, 8)

"""Sharding of Monte-Carlo samples over the GPUs of a node and the one collective of the path.

Samples are independent given the per-unit stream contract (include/gat_mi355.h), so rank r of G
takes the contiguous sample-id range shard_range(S, r, G) with no data-path collective; the only
exchange is ONE all-gather of the per-sample count matrix at the end (RCCL over xGMI when the
tensors live on GPUs: torch.distributed backend "nccl" is RCCL on ROCm; "gloo" on CPU for tests).
This replaces the reference's multiprocessing.Pool + result collation
(gat/__init__.py:681-700, :770-774).
"""
import numpy as np


def shard_range(n, rank, world):
    """contiguous range [begin, end) of rank; all but the last ranks get ceil(n/world) samples."""
    per = (n + world - 1) // world
    begin = min(n, rank * per)
    return begin, min(n, begin + per)


def padded_shard(n, world):
    return (n + world - 1) // world


def allgather_counts(local, n_total, group=None, always=False):
    """local: torch int64 tensor [K, A, padded_shard] holding this rank's columns (zero padded);
    returns [K, A, n_total] with the columns of all ranks in sample order.  always: the collective is made even in a
    group of one rank (the tests' way to run the whole path over RCCL on a one-GPU box)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (always and dist.is_initialized()):
        return local[..., :n_total]
    per = local.shape[-1]
    # concatenated form along dim 0 (accepted by both RCCL and gloo), viewed as [G, K, A, per]
    gathered = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(gathered, local.contiguous(), group=group)
    gathered = gathered.view((world,) + tuple(local.shape))
    # [G, K, A, per] -> [K, A, G*per] -> trim the padding of the last rank
    out = gathered.permute(1, 2, 0, 3).reshape(local.shape[0], local.shape[1], world * per)
    return out[..., :n_total].contiguous()


def shard_and_gather(fill, K, A, num_samples, device, group=None, order=None):
    """The sharded batch seam, whatever the device: every rank owns ceil(num_samples / world) columns of the gathered matrix,
    from rank * that on, and fills the ones whose sample ids exist -- [rank * per, min((rank + 1) * per, num_samples)): a
    sample id at or beyond num_samples is never drawn; the columns behind a short shard stay zero and fall off the end of the
    gathered matrix; a rank whose range is empty (more ranks than samples) fills nothing and still takes part in the ONE
    all-gather.  fill(lo, hi, block): write the columns of samples [lo, hi) into block [K, A, hi - lo] (int64, on `device`).
    order(before_gather: bool): hook for a host that fills on a stream of its own (gat_amd.run under nccl orders the
    library's stream against torch's).  Returns the [K, A, num_samples] matrix, on `device`, on every rank.
    (gat/__init__.py:681-700, :770-774: the reference's pool and its collation.)"""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    per = padded_shard(num_samples, world)
    lo, hi = shard_range(num_samples, rank, world)
    full = hi - lo == per
    shard = (torch.empty if full else torch.zeros)((K, A, per), dtype=torch.int64, device=device)
    if hi > lo:
        # (a short shard -- the last one that holds samples -- is filled as a block of its own width and copied in)
        block = shard if full else torch.empty((K, A, hi - lo), dtype=torch.int64, device=device)
        fill(lo, hi, block)
        if order is not None:
            order(True)
        if not full:
            shard[..., :hi - lo].copy_(block)
            if order is not None:
                order(block)
    return allgather_counts(shard, num_samples, group, always=True)


def gather_numpy(local_np, n_total, group=None):
    """numpy convenience wrapper for a matrix that is on the HOST already (gloo ranks; the gloo tests).  Under RCCL the
    count matrix never takes this way: gat_amd.sample_counts leaves the shard on the device, gathers device memory
    (allgather_counts) and reads the gathered matrix back once; a host matrix handed in under an nccl group is gathered
    through a gloo group of the same ranks instead of a round trip over the device."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(local_np))
    if dist.is_initialized() and dist.get_backend(group) == "nccl":
        return allgather_counts(t, n_total, _host_group(group)).numpy()
    return allgather_counts(t, n_total, group).numpy()


_HOST_GROUPS = {}


def _host_group(group):
    """a gloo group of the ranks of `group` (None: the world), made once per group and process-group generation: new_group
    is a collective over the WORLD, so a sub-group's members alone cannot make it -- a host matrix under an nccl sub-group is
    refused instead of hanging"""
    import torch.distributed as dist
    world_pg = dist.group.WORLD
    if not (group is None or group is world_pg):
        raise NotImplementedError("gather_numpy under an nccl SUB-group: create a gloo group of its ranks on every rank of "
                                  "the world (torch.distributed.new_group is a world collective) and pass that")
    entry = _HOST_GROUPS.get("world")
    if entry is None or entry[0] is not world_pg:                        # (none yet, or one of a destroyed process group)
        entry = _HOST_GROUPS["world"] = (world_pg, dist.new_group(backend="gloo"))   # (collective: every rank gets here together)
    return entry[1]

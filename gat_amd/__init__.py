"""gat_amd -- MI355X-native implementation of GAT's Monte-Carlo sampling + overlap-counting hot
path, drop-in for the reference's `gat.run()` seam (gat/__init__.py:855) and the classes it is
called with (gat/Engine.pyx).  Python host -> ctypes -> libgat_mi355.so (HIP, gfx950).
"""
import collections
import re

import numpy as np

from . import intervals, problem, synthetic            # noqa: F401
from . import io as IO                                   # noqa: F401
from . import stats as Stats                             # noqa: F401
from .engine import (SegmentList, PositionList, IntervalDictionary, IntervalCollection, Sampler, SamplerAnnotator, SamplerSegments,  # noqa: F401
                     Counter, computeCountsAll, overlap_sizes, CounterNucleotideOverlap, CounterNucleotideDensity, CounterSegmentOverlap,
                     CounterSegmentMidpointOverlap, CounterAnnotationOverlap, CounterAnnotationMidpointOverlap,
                     UnconditionalWorkspace, ConditionalWorkspaceCooccurance, ConditionalWorkspaceCentered,
                     ConditionalWorkspaceAnnotationCentered, ConditionalWorkspaceSegmentCentered, computeCounts, AnnotatorResult, AnnotatorResultExtended,
                     getTwoSidedPValue, updatePValues, getNormedPValue, getEmpiricalPValue, getQValues, updateQValues,
                     get_context, set_device, default_device)

__version__ = "0.1"

COUNTERS = collections.OrderedDict([
    ("nucleotide-overlap", CounterNucleotideOverlap), ("nucleotide-density", CounterNucleotideDensity),
    ("segment-overlap", CounterSegmentOverlap), ("segment-midoverlap", CounterSegmentMidpointOverlap),
    ("annotation-overlap", CounterAnnotationOverlap), ("annotation-midoverlap", CounterAnnotationMidpointOverlap),
])


class _CountsPerTrack(list):
    """[ {annotation: counts} per counter ] as the reference returns it, plus the device's statistics of those rows."""
    stats = None


def _dist_state():
    """(rank, world size, backend) of an initialised torch.distributed, else (0, 1, None)."""
    # (a process group can only have been initialised by a caller that imported torch.distributed: importing it here
    #  would cost every plain run -- gat-run.py included -- torch's import, seconds to minutes on a cold machine)
    import sys
    dist = sys.modules.get("torch.distributed")
    if dist is not None and dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size(), dist.get_backend()
    return 0, 1, None


def _force_collective_path():
    """GAT_FORCE_COLLECTIVE_PATH=1 (tests): a process group of ONE rank takes the multi-rank path -- sharding, the all-gather,
    the read-back on demand -- so that a one-GPU box runs that code over RCCL"""
    import os
    return os.environ.get("GAT_FORCE_COLLECTIVE_PATH") == "1"


_NUMPY_MODEL_OK = None


def _numpy_summation_model_holds():
    """gat_null_stats restates the order in which numpy.mean / numpy.std add up a contiguous float64 array (gat_stats.h:
    chunks of 8 192 elements, each summed pairwise down to blocks of at most 128 with eight running sums, the chunk sums
    added left to right).  That order belongs to the numpy at hand, not to the reference: it is checked once per process
    on arrays whose sums depend on it, and a numpy that adds differently gets its statistics from numpy itself."""
    global _NUMPY_MODEL_OK
    if _NUMPY_MODEL_OK is not None:
        return _NUMPY_MODEL_OK

    def block(a):                                   # n <= 128
        n = len(a)
        if n < 8:
            r = 0.0
            for x in a:
                r = r + x
            return r
        r = list(a[:8])
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] = r[j] + a[i + j]
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < n:
            res = res + a[i]
            i += 1
        return res

    def pairwise(a):
        if len(a) <= 128:
            return block(a)
        n2 = len(a) // 2
        n2 -= n2 % 8
        return pairwise(a[:n2]) + pairwise(a[n2:])

    ok = True
    rs = np.random.RandomState(20261002)
    for n in (7, 100, 1000, 8192 + 331):
        a = rs.random_sample(n) * 1e5
        al = a.tolist()
        total = None
        for i in range(0, n, 8192):
            p = pairwise(al[i:i + 8192])
            total = p if total is None else total + p
        m = total / n
        dev = [(x - m) * (x - m) for x in al]
        total = None
        for i in range(0, n, 8192):
            p = pairwise(dev[i:i + 8192])
            total = p if total is None else total + p
        ok = ok and m == float(np.mean(a)) and float(np.sqrt(total / n)) == float(np.std(a))
    _NUMPY_MODEL_OK = bool(ok)
    return _NUMPY_MODEL_OK


def _device_stats_wanted(n_values):
    """null-distribution statistics on the device (gat_null_stats) whenever the count matrix is there anyway: the host
    needs 0.1 ms (10 000 samples) to 2 ms (100 000) per row for them.  GAT_DEVICE_STATS=1 / 0 forces / forbids it;
    otherwise it is used when this numpy sums the way the kernel restates (_numpy_summation_model_holds)."""
    import os
    env = os.environ.get("GAT_DEVICE_STATS")
    if env is not None:
        return env not in ("0", "")
    return n_values > 0 and _numpy_summation_model_holds()


class _TrackJob(object):
    """the sampling of one segment track between its two halves (_sample_start / _sample_finish): the problem on the device,
    the count matrix there, and -- when the call could be enqueued -- the call in flight"""
    __slots__ = ("ctx", "P", "flat", "names", "tracks", "num_samples", "seed", "dev", "enqueued", "samples_outfile", "mt_state",
                 "result")

    def __init__(self, **kw):
        self.dev, self.enqueued, self.result, self.P = None, False, None, None
        for k, v in kw.items():
            setattr(self, k, v)

    def abandon(self):
        """let go of the device side (an error elsewhere in the run)"""
        try:
            if self.P is not None:
                self.P.close()                      # (drops a call in flight)
                self.P = None
            if self.dev is not None:
                self.ctx.free(self.dev)
                self.dev = None
        except Exception:
            pass


def _sample_start(segs, annotations, workspace, sampler, counters, num_samples, seed, ctx=None, samples_outfile=None,
                  workspace_generator=None, only_tracks=None, mt_state=None, _aflat=None, _shared=None):
    """First half of the batch seam (UnconditionalSampler.sample, gat/__init__.py:704-778): the inputs go to the device --
    the annotation tables only if no earlier track of this run() left them there (_shared) -- and, on the plain
    single-process path, the samples are ENQUEUED (gat_sample_and_count_enqueue): the caller computes observed counts and
    the sizes of its rows while the device samples, then calls _sample_finish.  Returns a _TrackJob; job.result is set
    already where there is nothing to sample."""
    from . import _lib
    job = _TrackJob(ctx=None, flat=None, names=[c.name for c in counters], tracks=None, num_samples=num_samples, seed=seed,
                    samples_outfile=samples_outfile, mt_state=mt_state)
    if workspace.sum() == 0:
        job.result = (None, 0)
        return job
    if annotations.hasPositions():
        # point annotations: the reference gets as far as the two counters that call into PositionList without isochores
        # (probed on the scratch build); anything else dies with this TypeError (gat/Engine.pyx:2866, :1417-1457)
        from .engine import POINT_COUNTERS, _points_type_error
        if any("." in k and k != "." for k in segs.keys()):
            raise _points_type_error()
        if any(c.name not in POINT_COUNTERS for c in counters):
            raise _points_type_error("annotations")
    ctx = job.ctx = ctx or get_context()
    tracks = job.tracks = list(annotations.tracks) if only_tracks is None else list(only_tracks)
    count_workspace = None
    if workspace_generator is not None and type(workspace_generator) is not UnconditionalWorkspace:
        count_workspace = workspace
        annos = annotations[tracks[0]] if only_tracks is not None else None
        segs, _, workspace = workspace_generator(segs, annos, workspace)
    bucket_size, nbuckets = getattr(sampler, "bucket_size", 0), getattr(sampler, "nbuckets", 100000)
    share = _shared if (only_tracks is None and count_workspace is None) else None
    flat = problem.flatten_dictionaries(segs, workspace, annotations, tracks, bucket_size, nbuckets, count_workspace=count_workspace,
                                        _aflat=_aflat if only_tracks is None else None, _shared=share)
    if flat is None:
        flat = problem.flatten_units(segs.asArrays(), workspace.asArrays(), [(t, annotations[t].asArrays()) for t in tracks],
                                     bucket_size, nbuckets,
                                     count_workspace=None if count_workspace is None else count_workspace.asArrays())
    flat["sampler"] = getattr(sampler, "kind", 0)
    job.flat = flat
    names = job.names
    if flat["n_contigs"] == 0:
        # nothing to place (the generated workspace holds no segments): computeSample skips every unit
        # (gat/__init__.py:536-538) and each counter sums over no contigs
        zero = [np.zeros(num_samples, dtype=np.float64 if n == "nucleotide-density" else np.int64) for n in names]
        job.result = ([collections.OrderedDict((t, zero[k].copy()) for t in tracks) for k in range(len(names))], flat["n_units"])
        return job
    # the annotation tables: one device object per (contigs in order, merge) of this run(), shared by its segment tracks
    shared_annos = None
    akey = flat.get("annotations_key")
    if share is not None and akey is not None:
        shared_annos = share.get(akey)
        if shared_annos is None and flat.get("annos") is not None:
            total = float(len(flat["segs"]))
            mean = float((flat["segs"]["end"].astype(np.int64) - flat["segs"]["start"]).sum()) / total if total else 0.0
            # (built by a thread of the library while this one goes on: the sampler's kernels do not wait for the tables; a run
            #  that only asks for the nucleotide counters -- the default -- says so: no per-track tables beside the merged index)
            shared_annos = share[akey] = _lib.Annotations(ctx, flat, mean_segment_length=mean, asynchronous=True,
                                                         nucleotide_only=all(n in ("nucleotide-overlap", "nucleotide-density") for n in names))
    job.P = _lib.Problem(ctx, flat, annotations=shared_annos)
    rank, world, _ = _dist_state()
    if world == 1 and mt_state is None and num_samples > 0 and not _force_collective_path():
        try:
            job.dev = ctx.alloc(len(names) * len(tracks) * num_samples * 8)
            job.P.enqueue(names, seed, 0, num_samples, job.dev)
            job.enqueued = True
        except Exception:
            job.abandon()
            raise
    return job


def _sample_finish(job, stat_vals=None):
    """Second half: waits for the samples (gat_wait), takes the null distributions' statistics where the matrix is
    (stat_vals: {counter name: {annotation: value}} -- the values whose p-value will be asked for, observed or observed /
    reference fold; large matrices then get their statistics on the device, gat_null_stats, and the result carries
    `.stats[counter index][annotation]` for AnnotatorResult), reads the matrix back, gathers it over the ranks of an
    initialised torch.distributed.  Returns ([ {annotation: array of num_samples} per counter ], number of work units)."""
    from . import distributed
    if job.result is not None:
        return job.result
    ctx, P, flat, names, tracks, num_samples, seed = job.ctx, job.P, job.flat, job.names, job.tracks, job.num_samples, job.seed
    mt_state, samples_outfile = job.mt_state, job.samples_outfile
    rank, world, backend = _dist_state()
    stats = None
    want_stats = stat_vals is not None and _device_stats_wanted(len(names) * len(tracks) * num_samples)

    def device_stats(ptr):
        vals = np.array([[stat_vals[n][t] for t in tracks] for n in names], dtype=np.float64).ravel()
        dbl = np.repeat([1 if n == "nucleotide-density" else 0 for n in names], len(tracks)).astype(np.uint8)
        st = ctx.null_stats(ptr, len(names) * len(tracks), num_samples, dbl, vals).reshape(len(names), len(tracks), 8)
        return [collections.OrderedDict((t, tuple(st[k, a, :6])) for a, t in enumerate(tracks)) for k in range(len(names))]

    try:
        begin, end = distributed.shard_range(num_samples, rank, world)
        if job.enqueued:
            # the call was enqueued by _sample_start; the matrix stays on the device until its statistics are taken
            P.wait()
            if want_stats:
                stats = device_stats(job.dev)
            host = np.empty((len(names), len(tracks), num_samples), dtype=np.int64)
            ctx.d2h(host, job.dev)
            ctx.free(job.dev)
            job.dev = None
            local = [host[k].view(np.float64) if n == "nucleotide-density" else host[k] for k, n in enumerate(names)]
        elif mt_state is not None:
            if world > 1:
                raise NotImplementedError("reference_stream: one stream is one GPU (the samples depend on each other)")
            if samples_outfile is not None:
                raise NotImplementedError("reference_stream: the sampled lists are not kept (counts only)")
            local = P.sample_and_count_serial(names, mt_state, num_samples)
            begin, end = 0, num_samples
        elif backend == "nccl" and (world > 1 or _force_collective_path()):
            local, stats = _nccl_shard_and_gather(ctx, P, names, tracks, seed, begin, end, num_samples, world,
                                                  device_stats if want_stats else None)
        else:
            local = P.sample_and_count(names, seed, begin, end)
            if world > 1:
                per = distributed.padded_shard(num_samples, world)
                stack = np.zeros((len(names), len(tracks), per), dtype=np.int64)
                for k in range(len(names)):
                    stack[k, :, :end - begin] = local[k].view(np.int64)
                full = distributed.gather_numpy(stack, num_samples)
                local = [full[k].view(np.float64) if names[k] == "nucleotide-density" else full[k] for k in range(len(names))]
        if samples_outfile is not None:
            # --output-samples-pattern (gat/__init__.py:515-559): per sample a track line, then the list the sampler
            # returned for every non-empty isochore unit under the unit's key (a rank writes its own shard)
            seg, off = P.sample(seed, begin, end, unit_level=True)
            U = flat["n_units"]
            for i in range(end - begin):
                samples_outfile.write("track name=%i\n" % (begin + i))
                for u in range(U):
                    if flat["unit_contig"][u] < 0:
                        continue
                    for s, e in seg[off[i * U + u]:off[i * U + u + 1]].tolist():
                        samples_outfile.write("%s\t%i\t%i\n" % (flat["unit_names"][u], s, e))
    finally:
        job.abandon()
    out = _CountsPerTrack()
    for k in range(len(names)):
        out.append(collections.OrderedDict((t, local[k][a]) for a, t in enumerate(tracks)))
    out.stats = stats
    job.result = (out, flat["n_units"])
    return job.result


class _DeviceCounts(object):
    """the gathered count matrix [counter][track][sample] where the collective left it -- on the device.  Rank 0 reads it
    back at once (it writes the count files); the other ranks hold rows that fetch it when somebody asks for their samples
    (the statistics of the rows come from the device: gat_null_stats)."""

    def __init__(self, tensor, names):
        self.tensor, self.names, self.host = tensor, names, None

    def fetch(self):
        if self.host is None:
            full = self.tensor.cpu().numpy()
            self.host = [full[k].view(np.float64) if n == "nucleotide-density" else full[k] for k, n in enumerate(self.names)]
            self.tensor = None
        return self.host

    def rows(self, k, n_rows, lazy):
        if not lazy:
            return self.fetch()[k]
        return [_DeviceRow(self, k, a) for a in range(n_rows)]


class _DeviceRow(object):
    """one row of a _DeviceCounts matrix: numpy.array(row) reads the matrix back (once for all rows)"""
    __slots__ = ("owner", "k", "a")

    def __init__(self, owner, k, a):
        self.owner, self.k, self.a = owner, k, a

    def __len__(self):
        t = self.owner.tensor
        return int(t.shape[-1]) if t is not None else len(self.owner.host[self.k][self.a])

    def __array__(self, dtype=None, copy=None):
        r = self.owner.fetch()[self.k][self.a]
        return r.astype(dtype) if dtype is not None else r

    def __iter__(self):
        return iter(self.__array__())

    def __getitem__(self, i):
        return self.__array__()[i]


def _nccl_shard_and_gather(ctx, P, names, tracks, seed, begin, end, num_samples, world, device_stats):
    """torch.distributed with the nccl backend (= RCCL): the shard's matrix stays on the device, ONE all-gather of device
    memory (the collation of the reference's pool, gat/__init__.py:694-700, :770-774), the statistics taken where the
    gathered matrix is, ONE read-back -- on rank 0; the other ranks read theirs when their rows' samples are asked for.

    Every rank owns ceil(num_samples / world) columns of the gathered matrix, from rank * that on, and samples the ids of
    its range that exist -- [rank * per, min((rank + 1) * per, num_samples)), the range distributed.shard_range gives the gloo
    path: a sample id at or beyond num_samples is never drawn (a device-side assertion or a slab overflow in a surplus
    sample must not fail a run that one process completes); the columns behind a short shard stay zero and fall off
    the end of the gathered matrix.  The library's stream is ordered against torch's with events (no device-wide
    synchronisation): torch.cuda.ExternalStream around the context's stream."""
    import torch
    import torch.distributed as dist
    from . import distributed
    rank = dist.get_rank() if dist.is_initialized() else 0
    dev = torch.device("cuda", ctx.device)
    K, A = len(names), len(tracks)
    lib_stream = torch.cuda.ExternalStream(ctx.stream_handle(), device=dev)
    cur = torch.cuda.current_stream(dev)

    def fill(lo, hi, block):
        lib_stream.wait_stream(cur)                           # (whatever torch still runs on memory it hands out here)
        P.sample_and_count_device(names, seed, lo, hi, block.data_ptr())

    def order(what):
        if what is True:
            cur.wait_stream(lib_stream)                       # the shard's columns are there before torch copies / gathers them
        else:
            what.record_stream(lib_stream)                    # (a short shard's block: the library's stream wrote it)

    # the shards' arithmetic and the ONE all-gather: distributed.shard_and_gather (the same code the CPU tests run over gloo);
    # [G, K, A, per] -> [K, A, G * per], the surplus of the last ranks cut off: the one device copy of the path
    full_t = distributed.shard_and_gather(fill, K, A, num_samples, dev, order=order)
    stats = None
    if device_stats is not None:
        lib_stream.wait_stream(cur)                           # gathered and re-ordered before k_null_stats reads
        stats = device_stats(full_t.data_ptr())
        full_t.record_stream(lib_stream)
    counts = _DeviceCounts(full_t, names)
    lazy = device_stats is not None and rank != 0
    return [counts.rows(k, A, lazy) for k in range(K)], stats


def sample_counts(segs, annotations, workspace, sampler, counters, num_samples, seed, ctx=None,
                  samples_outfile=None, workspace_generator=None, only_tracks=None, stat_vals=None, mt_state=None, _aflat=None,
                  _shared=None):
    """The batch seam: replaces UnconditionalSampler.sample (gat/__init__.py:704-778).

    segs / workspace: IntervalDictionary (isochore level); annotations: IntervalCollection.
    workspace_generator (gat/__init__.py:727): segments and workspace the sampler sees are
    generator(segs, None, workspace); the counters keep the contig form of `workspace`.
    only_tracks: count these annotation tracks only (the conditional sampler's per-annotation pass).
    stat_vals: {counter name: {annotation: value}} -- the values whose p-value will be asked for (observed, or
    observed / reference fold); given them, large matrices get their statistics on the device (gat_null_stats) and the
    result carries `.stats[counter index][annotation]` for AnnotatorResult.
    mt_state: the reference's own stream (run(reference_stream=True)): the samples are drawn from this ONE MT19937
    state (_lib.mt19937_seed), which is advanced in place; one GPU, no sharding.
    Returns ([ {annotation: array of num_samples} per counter ] like the reference, number of work
    units), or (None, 0) for an empty workspace.  If torch.distributed is initialised, samples are
    sharded over the ranks and the count matrix is all-gathered (RCCL).
    (The two halves -- _sample_start enqueues, _sample_finish waits -- are what run() calls with its own work in between.)"""
    job = _sample_start(segs, annotations, workspace, sampler, counters, num_samples, seed, ctx=ctx, samples_outfile=samples_outfile,
                        workspace_generator=workspace_generator, only_tracks=only_tracks, mt_state=mt_state, _aflat=_aflat,
                        _shared=_shared)
    return _sample_finish(job, stat_vals)


def run(segments, annotations, workspace, sampler, counters, workspace_generator, **kwargs):
    """run an enrichment analysis: same signature and result type as the reference's gat.run
    (gat/__init__.py:855-1088); see _run.  (Whether the annotations hold point lists is asked of every list once for the
    whole call instead of at each of its three uses: 19 200 lists on an isochore problem.)"""
    annotations._points_memo = annotations._plain_memo = None
    annotations._ranges_memo = {}
    annotations._points_memo = annotations.hasPositions()
    annotations._plain_memo = (not annotations._points_memo) and all(d._all_normalized() for d in annotations.intervals.values())
    try:
        return _run(segments, annotations, workspace, sampler, counters, workspace_generator, **kwargs)
    finally:
        annotations._points_memo = annotations._plain_memo = annotations._ranges_memo = None


def _run(segments, annotations, workspace, sampler, counters, workspace_generator, **kwargs):
    """run an enrichment analysis: same signature and result type as the reference's gat.run
    (gat/__init__.py:855-1088).

    kwargs: num_samples, pseudo_count, reference, output_counts_pattern, output_samples_pattern
    (as in the reference) and random_seed (base of the per-unit streams; default: drawn from numpy's
    global RandomState, so numpy.random.seed() makes a run reproducible).  reference_stream=True: random_seed seeds ONE
    stream for the whole run, as the reference's gat-run.py --random-seed does (same numbers as an unpatched reference;
    one wave's speed).  num_threads is accepted
    and ignored (the GPU replaces the process pool)."""
    num_samples = kwargs.get("num_samples", 10000)
    pseudo_count = kwargs.get("pseudo_count", 1.0)
    reference = kwargs.get("reference", None)
    output_counts_pattern = kwargs.get("output_counts_pattern", None)
    output_samples_pattern = kwargs.get("output_samples_pattern", None)
    seed = kwargs.get("random_seed", None)
    reference_stream = bool(kwargs.get("reference_stream", False))
    sample_files = kwargs.get("sample_files", None) or []
    if sample_files:
        # gat/__init__.py:952-961 + Engine.pyx:3215-3233 (SamplesFile): the files are read (every parse error of the bed
        # reader surfaces), each one's track name is what the samples pattern's "%s" matches in its file name, a run that
        # loads samples writes none (:977) -- and the samplers never look at what was loaded (UnconditionalSampler /
        # ConditionalSampler keep `samples` and sample afresh): the table is the table of a plain run.  (The reference
        # builds the regex with re.sub("%s", "(\S+)", pattern), an escape Python >= 3.7 refuses in a replacement string:
        # under a current Python it dies there with re.error; this is the regex that line meant.)
        if not output_samples_pattern:
            raise ValueError("require output_samples_pattern if loading samples from files")
        regex = re.compile(output_samples_pattern.replace("%s", "(\\S+)"))
        loaded = {}
        for filename in sample_files:
            track = regex.search(filename).groups()[0]       # (no match: AttributeError, as in the reference)
            coll = IntervalCollection(track)
            coll.load(filename)
            loaded[track] = coll
        output_samples_pattern = None
    rank, world, _ = _dist_state()
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 32))
        if world > 1:
            # one base seed for all ranks: their sample-id shards must belong to one family of unit streams
            import torch.distributed as dist
            box = [seed]
            dist.broadcast_object_list(box, src=0)
            seed = int(box[0])
    conditional = getattr(workspace_generator, "is_conditional", False)
    if not isinstance(sampler, (SamplerAnnotator, SamplerSegments)):
        raise NotImplementedError("only SamplerAnnotator and SamplerSegments run on the GPU path")
    mt_state = None
    if reference_stream:
        # the reference's own stream: numpy.random.seed(seed) once (scripts/gat-run.py:267-271), every work unit of every
        # segment track drawing from it in order -- an unpatched reference's table, number for number, at one stream's speed
        from . import _lib
        mt_state = _lib.mt19937_seed(seed)

    # The reference's order is observed counts (gat/__init__.py:933-940), then per segment track the sampling (:971-1010),
    # then the rows (:1000-1068).  Here a track's sampling is ENQUEUED on the device first, and what the host has to do
    # itself -- the observed counts (a device call of its own, queued behind the samples), the sizes of the rows'
    # dictionaries and intersections -- happens while the device samples; the next track is enqueued before the current
    # one is waited for, and the annotation tables are made once for all tracks (_shared).
    from .engine import _lookup_every_key
    if len(segments.tracks) and len(annotations.tracks) and counters:
        _lookup_every_key(segments, annotations, workspace)      # (what computeCounts does to the dictionaries comes first, as there)
    aflat = annotations._flat(list(annotations.tracks)) if len(annotations) else None   # the annotations as one array, looked over once
    sampled_counts = {}
    state = {"observed": None}
    shared = {}                      # (contigs, merge) -> the annotation tables on the device
    sizes = {}                       # (counts, sum) per dictionary object: the same track / workspace recur in every row
    overlaps = {}                    # the intersections' sizes, every annotation of a segment track in one call
    unconditional = type(workspace_generator) is UnconditionalWorkspace

    def observed():
        if state["observed"] is None:
            state["observed"] = computeCountsAll(counters, sum, segments, annotations, workspace, workspace_generator, _aflat=aflat)
            if unconditional and aflat is not None and len(aflat[2]) >= 16 and not annotations.hasPositions():
                # (counts, sum) of every annotation dictionary from one native pass over the collection's array: list sums
                # as SegmentList.sum() forms them (a uint32 accumulator each), added up per dictionary
                from . import _lib
                data, bases, flats = aflat
                nl = np.fromiter((len(f.keys) for f in flats), dtype=np.int64, count=len(flats))
                lb = np.concatenate([f.off[:-1] for f in flats]) + np.repeat(bases[:-1], nl)
                le = np.concatenate([f.off[1:] for f in flats]) + np.repeat(bases[:-1], nl)
                per = _lib.list_sums(data, lb, le)
                first = np.zeros(len(flats) + 1, dtype=np.int64)
                np.cumsum(nl, out=first[1:])
                run_ = np.zeros(len(per) + 1, dtype=np.int64)
                np.cumsum(per, out=run_[1:])
                tot = (run_[first[1:]] - run_[first[:-1]]).tolist()
                for t, f, x in zip(annotations.tracks, flats, tot):
                    sizes[id(annotations[t])] = (int(f.off[-1]), int(x))
        return state["observed"]

    def finish(track, outf, job):
        stat_vals = None
        if not conditional:
            # the values whose p-values the rows will ask for: large matrices get their statistics on the device
            stat_vals = {}
            for c, oc in zip(counters, observed()):
                stat_vals[c.name] = {}
                for annotation in annotations.tracks:
                    v = float(oc[track][annotation]) if track in oc and annotation in oc[track] else 0.0
                    if reference:
                        f = reference[track][annotation].fold
                        v = v / f if f > 0 else v
                    stat_vals[c.name][annotation] = v
        try:
            r, _ = _sample_finish(job, stat_vals)
        finally:
            if outf:
                outf.close()
        if r is not None:
            sampled_counts[track] = r
            # the track's result rows now, while the device samples the next track (assembled in the reference's order --
            # counter, track, annotation, gat/__init__.py:1040-1068 -- at the end)
            for counter_id in range(len(counters)):
                if track in observed()[counter_id]:
                    rows_by[(counter_id, track)] = make_rows(counter_id, track)

    rows_by = {}
    keep = []                        # the generated dictionaries stay alive so that their ids stay theirs

    def make_rows(counter_id, track):
        counter, rows = counters[counter_id], []
        for annotation, observed_value in observed()[counter_id][track].items():
            temp_segs, temp_annos, temp_workspace = workspace_generator(segments[track], annotations[annotation], workspace)
            keep.append((temp_segs, temp_annos, temp_workspace))
            if id(temp_workspace) not in sizes:
                sizes[id(temp_workspace)] = (temp_workspace.counts(), temp_workspace.sum())
            if sizes[id(temp_workspace)][1] == 0:
                continue
            ref = reference[track][annotation] if reference else None
            dev_stats = getattr(sampled_counts[track], "stats", None)
            rows.append(AnnotatorResultExtended(
                track=track, annotation=annotation, counter=counter.name, observed=observed_value,
                samples=sampled_counts[track][counter_id][annotation], track_segments=temp_segs,
                annotation_segments=temp_annos, workspace=temp_workspace, reference=ref,
                pseudo_count=pseudo_count, _sizes=sizes,
                _stats=dev_stats[counter_id][annotation] if dev_stats else None,
                _overlap=overlaps[track].get(annotation) if overlaps.get(track) else None))
        return rows

    inflight = collections.deque()
    try:
        for track in segments.tracks:
            outf = None
            if output_samples_pattern:
                # (under torch.distributed every rank writes the samples of its own shard: <file>.rank<r>)
                outf = open(re.sub("%s", track, output_samples_pattern) + (".rank%d" % rank if world > 1 else ""), "w")
            if conditional:
                observed()
                # ConditionalSampler.sample (gat/__init__.py:780-850): one sampling pass per annotation, each in the
                # workspace conditioned on (segments, that annotation); only that annotation is counted
                r = None
                if workspace.sum() > 0:
                    r = [collections.OrderedDict() for _ in counters]
                    for annotation in annotations.tracks:
                        ra, n_units = sample_counts(segments[track], annotations, workspace, sampler, counters, num_samples,
                                                    seed, samples_outfile=outf, workspace_generator=workspace_generator,
                                                    only_tracks=[annotation], mt_state=mt_state)
                        for k in range(len(counters)):
                            r[k][annotation] = ra[k][annotation]
                        seed = (seed + num_samples * n_units) & 0xFFFFFFFF
                if outf:
                    outf.close()
                if r is not None:
                    sampled_counts[track] = r
                continue
            try:
                job = _sample_start(segments[track], annotations, workspace, sampler, counters, num_samples, seed,
                                    samples_outfile=outf, workspace_generator=workspace_generator, mt_state=mt_state,
                                    _aflat=aflat, _shared=shared)
            except Exception:
                observed()           # (the reference has counted before it samples: an error there comes first)
                raise
            n_units = job.result[1] if job.result is not None else job.flat["n_units"]
            seed = (seed + num_samples * n_units) & 0xFFFFFFFF    # next track: disjoint unit streams
            inflight.append((track, outf, job))
            # ... while the device samples: first what needs no device (the observed counts' own device call is queued
            # behind the samples)
            if unconditional and job.result is None:              # (the rows see the dictionaries themselves)
                overlaps[track] = overlap_sizes(segments[track], annotations, _aflat=aflat)
            observed()
            while len(inflight) > 1:
                finish(*inflight.popleft())
        observed()
        while inflight:
            finish(*inflight.popleft())
    finally:
        for _, outf, job in inflight:
            job.abandon()
            if outf:
                outf.close()
        for a in shared.values():
            a.close()
    observed_counts = observed()

    annotator_results = []
    for counter_id, (counter, observed_count) in enumerate(zip(counters, observed_counts)):
        for track in observed_count:
            if track in sampled_counts:
                annotator_results.extend(rows_by[(counter_id, track)] if (counter_id, track) in rows_by else make_rows(counter_id, track))
    if output_counts_pattern and rank == 0:               # (every rank holds the gathered matrix: one writer)
        for counter in counters:
            with open(re.sub("%s", counter.name, output_counts_pattern), "w") as outfile:
                outfile.write("track\tannotation\tobserved\tcounts\n")
                for o in [x for x in annotator_results if x.counter == counter.name]:
                    outfile.write("%s\t%s\t%i\t%s\n" % (o.track, o.annotation, o.observed,
                                                      ",".join("%i" % x for x in o.samples)))
    return annotator_results


def fromCounts(filename):
    """annotator results from a counts table written by --output-counts-pattern (gat/__init__.py:1091-1117; the
    entry point of gat-compare.py)."""
    annotator_results = []
    with IO.openFile(filename, "r") as infile:
        header = infile.readline()
        if not header == "track\tannotation\tobserved\tcounts\n":
            raise ValueError("%s not a counts file: got %s" % (infile, header))
        for line in infile:
            track, annotation, observed, counts = line[:-1].split("\t")
            samples = np.array(list(map(float, counts.split(","))), dtype=np.float64)
            annotator_results.append(AnnotatorResult(track=track, annotation=annotation, counter="na",
                                                     observed=float(observed), samples=samples))
    return annotator_results


def buildParser(usage=None):
    """gat command line parser: the options of the reference's buildParser (gat/__init__.py:54-429)
    that concern the accelerated path, with the same names, destinations and defaults."""
    import optparse
    parser = optparse.OptionParser(version="%prog (gat_amd " + __version__ + ")", usage=usage)
    g = optparse.OptionGroup(parser, "Input options")
    g.add_option("-a", "--annotation-bed-file", "--annotations", "--annotation-file", dest="annotation_files",
                 type="string", action="append", help="filename with annotations")
    g.add_option("-s", "--segment-bed-file", "--segments", "--segment-file", dest="segment_files", type="string",
                 action="append", help="filename with segments")
    g.add_option("-w", "--workspace-bed-file", "--workspace", "--workspace-file", dest="workspace_files",
                 type="string", action="append", help="filename with workspace segments")
    g.add_option("-i", "--isochore-bed-file", "--isochores", "--isochore-file", dest="isochore_files", type="string",
                 action="append", help="filename with isochore segments")
    g.add_option("--ignore-segment-tracks", dest="ignore_segment_tracks", action="store_true",
                 help="all segments belong to one track called 'merged' [default]")
    g.add_option("--with-segment-tracks", dest="ignore_segment_tracks", action="store_false",
                 help="the segments file is arranged in tracks")
    g.add_option("--enable-split-tracks", dest="enable_split_tracks", action="store_true")
    g.add_option("--annotations-label", dest="annotations_label", type="string")
    g.add_option("--annotations-to-points", dest="annotations_to_points", type="choice", choices=("midpoint", "start", "end"),
                 help="convert annotations from segments to positions (counters annotation-overlap / annotation-midoverlap)")
    g.add_option("-l", "--sample-file", dest="sample_files", type="string", action="append",
                 help="files with samples written by --output-samples-pattern (needs that pattern; see gat_amd.run)")
    g.add_option("--input-counts-file", dest="input_filename_counts", type="string",
                 help="start from a counts table written by --output-counts-pattern (statistics re-computed)")
    g.add_option("--input-results-file", dest="input_filename_results", type="string",
                 help="start from a previous results table (re-computes the fdr)")
    g.add_option("--descriptions", dest="input_filename_descriptions", type="string",
                 help="tab-separated file mapping annotations to extra description columns")
    parser.add_option_group(g)
    g = optparse.OptionGroup(parser, "Output options")
    g.add_option("-o", "--order", dest="output_order", type="choice",
                 choices=("track", "annotation", "fold", "pvalue", "qvalue", "observed"))
    g.add_option("--output-tables-pattern", dest="output_tables_pattern", type="string")
    g.add_option("--output-counts-pattern", dest="output_counts_pattern", type="string")
    g.add_option("--output-samples-pattern", dest="output_samples_pattern", type="string")
    g.add_option("--output-stats", dest="output_stats", type="choice", action="append",
                 choices=("all", "annotations", "segments", "workspaces", "isochores", "overlap"),
                 help="write summary statistics of the collections at the stages of the input pipeline")
    g.add_option("--output-bed", dest="output_bed", type="choice", action="append",
                 choices=("all", "annotations", "segments", "workspaces", "isochores"),
                 help="write the collections after the isochores were applied as bed files")
    g.add_option("-P", "--output-filename-pattern", dest="output_filename_pattern", type="string",
                 help="pattern of the side files, %s is replaced by the section [default=%default]")
    g.add_option("--force", dest="output_force", action="store_true", help="overwrite existing side files")
    parser.add_option_group(g)
    g = optparse.OptionGroup(parser, "Sampling algorithm options")
    g.add_option("-c", "--counter", dest="counters", type="choice", action="append", choices=tuple(COUNTERS.keys()))
    g.add_option("-m", "--sampler", dest="sampler", type="choice", choices=("annotator", "segments"))
    g.add_option("-n", "--num-samples", dest="num_samples", type="int")
    g.add_option("--bucket-size", dest="bucket_size", type="int")
    g.add_option("--nbuckets", dest="nbuckets", type="int")
    parser.add_option_group(g)
    g = optparse.OptionGroup(parser, "Statistics options")
    g.add_option("-p", "--pvalue-method", dest="pvalue_method", type="choice", choices=("empirical", "norm"))
    g.add_option("-q", "--qvalue-method", dest="qvalue_method", type="choice",
                 choices=("storey", "BH", "bonferroni", "holm", "hommel", "hochberg", "BY", "none"))
    g.add_option("--qvalue-lambda", dest="qvalue_lambda", type="float", help="fdr computation: lambda")
    g.add_option("--qvalue-pi0-method", dest="qvalue_pi0_method", type="choice", choices=("smoother", "bootstrap"),
                 help="fdr computation: method for estimating pi0")
    g.add_option("--pseudo-count", dest="pseudo_count", type="float")
    parser.add_option_group(g)
    g = optparse.OptionGroup(parser, "Processing options")
    g.add_option("-t", "--num-threads", dest="num_threads", type="int", help="accepted and ignored: the GPU replaces the pool")
    g.add_option("--random-seed", dest="random_seed", type="int",
                 help="base of the random streams: work unit (sample s, isochore unit u) draws from numpy's legacy "
                      "RandomState seeded with (seed + s * n_units + u) mod 2^32.  The reference seeds ONE global stream with "
                      "this value, so its sampled columns are statistically equal to, not identical with, the ones printed here "
                      "(observed counts, sizes and densities are identical); with the per-unit seeding patched into the "
                      "reference the tables are byte-identical (tests/golden/make_goldens.py)")
    g.add_option("--reference-stream", dest="reference_stream", action="store_true",
                 help="(not in the reference) draw every sample from ONE stream seeded with --random-seed, in the reference's "
                      "order: the table of an unpatched reference run with the same seed, number for number.  One stream is one "
                      "chain of dependent draws: a single GPU wave runs it (0.4 ms per work unit of 400 segments)")
    g.add_option("--truncate-segments-to-workspace", dest="truncate_segments_to_workspace", action="store_true")
    g.add_option("--truncate-workspace-to-annotations", dest="truncate_workspace_to_annotations", action="store_true")
    g.add_option("--restrict-workspace", dest="restrict_workspace", action="store_true")
    g.add_option("--conditional", dest="conditional", type="choice",
                 choices=("unconditional", "annotation-centered", "segment-centered", "cooccurance"),
                 help="conditional workspace creation [default=%default]")
    g.add_option("--conditional-extension", dest="conditional_extension", type="int")
    g.add_option("--conditional-expansion", dest="conditional_expansion", type="float")
    g.add_option("--device", dest="device", type="int", help="HIP device ordinal [default=%default]")
    parser.add_option_group(g)
    g = optparse.OptionGroup(parser, "Common options")
    g.add_option("-v", "--verbose", dest="loglevel", type="int")
    g.add_option("-S", "--stdout", dest="stdout", type="string", metavar="FILE")
    g.add_option("-L", "--log", dest="stdlog", type="string", metavar="FILE")
    parser.add_option_group(g)
    parser.set_defaults(input_filename_counts=None, output_stats=[], output_bed=[], output_filename_pattern="%s", output_force=False,
                        input_filename_results=None, input_filename_descriptions=None, annotation_files=[], annotations_label=None, annotations_to_points=None, bucket_size=0,
                        counters=[], enable_split_tracks=False, ignore_segment_tracks=True, isochore_files=[],
                        nbuckets=100000, num_samples=1000, num_threads=0, output_counts_pattern=None,
                        output_order="fold", output_samples_pattern=None, output_tables_pattern="%s.tsv.gz",
                        overlapping_annotations=False, pseudo_count=1.0, pvalue_method="empirical", qvalue_method="BH",
                        qvalue_lambda=None, qvalue_pi0_method="smoother",
                        random_seed=None, reference_stream=False, restrict_workspace=False, sample_files=[], sampler="annotator", segment_files=[],
                        truncate_segments_to_workspace=False, truncate_workspace_to_annotations=False,
                        conditional="unconditional", conditional_extension=None, conditional_expansion=None,
                        workspace_files=[], device=0, loglevel=1, stdout=None, stdlog=None)
    return parser


def fromSegments(options, args=None):
    """run an analysis from BED files: scripts/gat-run.py:77-220 of the reference."""
    segments, annotations, workspaces, isochores = IO.buildSegments(options)
    workspace = IO.applyIsochores(segments, annotations, workspaces, options, isochores,
                                  truncate_segments_to_workspace=options.truncate_segments_to_workspace,
                                  truncate_workspace_to_annotations=options.truncate_workspace_to_annotations,
                                  restrict_workspace=options.restrict_workspace)
    if options.sampler == "annotator":
        sampler = SamplerAnnotator(bucket_size=options.bucket_size, nbuckets=options.nbuckets)
    elif options.sampler == "segments":
        sampler = SamplerSegments()                      # scripts/gat-run.py:133 passes no bucket arguments
    else:
        raise ValueError("sampler '%s' is outside the accelerated path" % options.sampler)
    counters = []
    for counter in options.counters:
        if counter not in COUNTERS:
            raise ValueError("unknown counter '%s'" % counter)
        counters.append(COUNTERS[counter]())
    # scripts/gat-run.py:162-186 (both centered modes insist on --conditional-expansion, as the reference does)
    if options.conditional == "unconditional":
        workspace_generator = UnconditionalWorkspace()
    elif options.conditional == "cooccurance":
        workspace_generator = ConditionalWorkspaceCooccurance()
    elif options.conditional in ("annotation-centered", "segment-centered"):
        if options.conditional_expansion is None:
            raise ValueError("please specify either --conditional-expansion or --conditional-extension")
        cls = (ConditionalWorkspaceAnnotationCentered if options.conditional == "annotation-centered"
               else ConditionalWorkspaceSegmentCentered)
        workspace_generator = cls(options.conditional_extension, options.conditional_expansion)
    else:
        raise ValueError("unknown conditional workspace '%s'" % options.conditional)
    return run(segments, annotations, workspace, sampler, counters, workspace_generator=workspace_generator,
               num_samples=options.num_samples, output_counts_pattern=options.output_counts_pattern,
               output_samples_pattern=options.output_samples_pattern, pseudo_count=options.pseudo_count,
               num_threads=options.num_threads, random_seed=options.random_seed,
               reference_stream=getattr(options, "reference_stream", False),
               sample_files=getattr(options, "sample_files", []))

"""gat_amd -- MI355X-native implementation of GAT's Monte-Carlo sampling + overlap-counting hot
path, drop-in for the reference's `gat.run()` seam (gat/__init__.py:855) and the classes it is
called with (gat/Engine.pyx).  Python host -> ctypes -> libgat_mi355.so (HIP, gfx950).
"""
import collections
import re

import numpy as np

from . import intervals, problem, synthetic            # noqa: F401
from .engine import (SegmentList, IntervalDictionary, IntervalCollection, Sampler, SamplerAnnotator,   # noqa: F401
                     Counter, CounterNucleotideOverlap, CounterNucleotideDensity, CounterSegmentOverlap,
                     CounterSegmentMidpointOverlap, CounterAnnotationOverlap, CounterAnnotationMidpointOverlap,
                     UnconditionalWorkspace, computeCounts, AnnotatorResult, AnnotatorResultExtended,
                     getTwoSidedPValue, updatePValues, get_context)

__version__ = "0.1"

COUNTERS = collections.OrderedDict([
    ("nucleotide-overlap", CounterNucleotideOverlap), ("nucleotide-density", CounterNucleotideDensity),
    ("segment-overlap", CounterSegmentOverlap), ("segment-midoverlap", CounterSegmentMidpointOverlap),
    ("annotation-overlap", CounterAnnotationOverlap), ("annotation-midoverlap", CounterAnnotationMidpointOverlap),
])


def sample_counts(segs, annotations, workspace, sampler, counters, num_samples, seed, ctx=None,
                  samples_outfile=None):
    """The batch seam: replaces UnconditionalSampler.sample (gat/__init__.py:704-778).

    segs / workspace: IntervalDictionary (isochore level); annotations: IntervalCollection.
    Returns [ {annotation: array of num_samples} per counter ] like the reference, or None for an
    empty workspace.  If torch.distributed is initialised, samples are sharded over the ranks and
    the count matrix is all-gathered (RCCL)."""
    from . import _lib, distributed
    if workspace.sum() == 0:
        return None, 0
    ctx = ctx or get_context()
    tracks = list(annotations.tracks)
    flat = problem.flatten_units(segs.asArrays(), workspace.asArrays(),
                                 [(t, annotations[t].asArrays()) for t in tracks],
                                 getattr(sampler, "bucket_size", 0), getattr(sampler, "nbuckets", 100000))
    names = [c.name for c in counters]
    rank, world = 0, 1
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(), dist.get_world_size()
    except ImportError:
        pass
    P = _lib.Problem(ctx, flat)
    try:
        begin, end = distributed.shard_range(num_samples, rank, world)
        local = P.sample_and_count(names, seed, begin, end)
        if samples_outfile is not None:
            seg, off = P.sample(seed, begin, end)
            C = flat["n_contigs"]
            for i in range(end - begin):
                samples_outfile.write("track name=%i\n" % (begin + i))
                for c in range(C):
                    for s, e in seg[off[i * C + c]:off[i * C + c + 1]].tolist():
                        samples_outfile.write("%s\t%i\t%i\n" % (flat["contig_names"][c], s, e))
    finally:
        P.close()
    if world > 1:
        per = distributed.padded_shard(num_samples, world)
        stack = np.zeros((len(names), len(tracks), per), dtype=np.int64)
        for k in range(len(names)):
            stack[k, :, :end - begin] = local[k].view(np.int64)
        full = distributed.gather_numpy(stack, num_samples)
        local = [full[k].view(np.float64) if names[k] == "nucleotide-density" else full[k] for k in range(len(names))]
    out = []
    for k in range(len(names)):
        out.append(collections.OrderedDict((t, local[k][a]) for a, t in enumerate(tracks)))
    return out, flat["n_units"]


def run(segments, annotations, workspace, sampler, counters, workspace_generator, **kwargs):
    """run an enrichment analysis: same signature and result type as the reference's gat.run
    (gat/__init__.py:855-1088).

    kwargs: num_samples, pseudo_count, reference, output_counts_pattern, output_samples_pattern
    (as in the reference) and random_seed (base of the per-unit streams; default: drawn from numpy's
    global RandomState, so numpy.random.seed() makes a run reproducible).  num_threads is accepted
    and ignored (the GPU replaces the process pool)."""
    num_samples = kwargs.get("num_samples", 10000)
    pseudo_count = kwargs.get("pseudo_count", 1.0)
    reference = kwargs.get("reference", None)
    output_counts_pattern = kwargs.get("output_counts_pattern", None)
    output_samples_pattern = kwargs.get("output_samples_pattern", None)
    seed = kwargs.get("random_seed", None)
    if seed is None:
        seed = int(np.random.randint(0, 2 ** 32))
    if getattr(workspace_generator, "is_conditional", False):
        raise NotImplementedError("conditional workspaces are outside the accelerated path")
    if not isinstance(sampler, SamplerAnnotator):
        raise NotImplementedError("only SamplerAnnotator runs on the GPU path")

    observed_counts = [computeCounts(counter=c, aggregator=sum, segments=segments, annotations=annotations,
                                     workspace=workspace, workspace_generator=workspace_generator) for c in counters]
    sampled_counts = {}
    for track in segments.tracks:
        outf = None
        if output_samples_pattern:
            outf = open(re.sub("%s", track, output_samples_pattern), "w")
        r, n_units = sample_counts(segments[track], annotations, workspace, sampler, counters, num_samples, seed,
                                   samples_outfile=outf)
        if outf:
            outf.close()
        if r is None:
            continue
        sampled_counts[track] = r
        seed = (seed + num_samples * n_units) & 0xFFFFFFFF        # next track: disjoint unit streams

    annotator_results = []
    for counter_id, (counter, observed_count) in enumerate(zip(counters, observed_counts)):
        for track, r in observed_count.items():
            if track not in sampled_counts:
                continue
            for annotation, observed in r.items():
                temp_segs, temp_annos, temp_workspace = workspace_generator(segments[track], annotations[annotation], workspace)
                if temp_workspace.sum() == 0:
                    continue
                ref = reference[track][annotation] if reference else None
                annotator_results.append(AnnotatorResultExtended(
                    track=track, annotation=annotation, counter=counter.name, observed=observed,
                    samples=sampled_counts[track][counter_id][annotation], track_segments=temp_segs,
                    annotation_segments=temp_annos, workspace=temp_workspace, reference=ref,
                    pseudo_count=pseudo_count))
    if output_counts_pattern:
        for counter in counters:
            with open(re.sub("%s", counter.name, output_counts_pattern), "w") as outfile:
                outfile.write("track\tannotation\tobserved\tcounts\n")
                for o in [x for x in annotator_results if x.counter == counter.name]:
                    outfile.write("%s\t%s\t%i\t%s\n" % (o.track, o.annotation, o.observed,
                                                      ",".join("%i" % x for x in o.samples)))
    return annotator_results

"""Synthetic hg19-shaped inputs for the BASELINE.json configurations.

Pure integer numpy, seedable, identical on every box.  Used by bench.py, the parity tests and
tests/golden/make_goldens.py.  Shapes follow the reference tutorial data
(tutorial/TutorialIntervalOverlap/*.bed.gz: short segments, ~10^4..10^5-interval annotation
tracks, ungapped-contig workspaces) but contain no reference data.
"""
import collections
import os

import numpy as np

SEG = np.dtype([("start", "<u4"), ("end", "<u4")])

# hg19 primary assembly (public UCSC chromosome sizes)
HG19 = collections.OrderedDict([
    ("chr1", 249250621), ("chr2", 243199373), ("chr3", 198022430), ("chr4", 191154276),
    ("chr5", 180915260), ("chr6", 171115067), ("chr7", 159138663), ("chr8", 146364022),
    ("chr9", 141213431), ("chr10", 135534747), ("chr11", 135006516), ("chr12", 133851895),
    ("chr13", 115169878), ("chr14", 107349540), ("chr15", 102531392), ("chr16", 90354753),
    ("chr17", 81195210), ("chr18", 78077248), ("chr19", 59128983), ("chr20", 63025520),
    ("chr21", 48129895), ("chr22", 51304566), ("chrX", 155270560), ("chrY", 59373566),
])
CHR22 = collections.OrderedDict([("chr22", 51304566)])


def _normalize(start, end):
    """sort by start, merge overlapping (not adjacent) intervals, drop empties."""
    keep = end > start
    start, end = start[keep], end[keep]
    order = np.argsort(start, kind="stable")
    start, end = start[order], end[order]
    if len(start) == 0:
        return np.empty(0, dtype=SEG)
    run = np.maximum.accumulate(end)
    head = np.ones(len(start), dtype=bool)
    head[1:] = start[1:] >= run[:-1]
    idx = np.flatnonzero(head)
    out = np.empty(len(idx), dtype=SEG)
    out["start"] = start[idx]
    last = np.append(idx[1:] - 1, len(start) - 1)
    out["end"] = run[last]
    return out


def random_segments(contigs, n, mean_len, seed):
    """n intervals placed uniformly over the genome, lengths geometric(1/mean_len), normalized.

    returns OrderedDict contig -> SEG array (contigs without intervals are omitted)."""
    rs = np.random.RandomState(seed)
    names = list(contigs.keys())
    sizes = np.array([contigs[c] for c in names], dtype=np.int64)
    cum = np.cumsum(sizes)
    pos = rs.randint(0, cum[-1], size=n).astype(np.int64)
    length = rs.geometric(1.0 / mean_len, size=n).astype(np.int64)
    ci = np.searchsorted(cum, pos, side="right")
    start = pos - (cum[ci] - sizes[ci])
    end = np.minimum(start + length, sizes[ci])
    out = collections.OrderedDict()
    for i, name in enumerate(names):
        m = ci == i
        if m.any():
            segs = _normalize(start[m].astype(np.uint32), end[m].astype(np.uint32))
            if len(segs):
                out[name] = segs
    return out


def workspace_contigs(contigs):
    """one workspace segment per contig, [0, size)."""
    out = collections.OrderedDict()
    for name, size in contigs.items():
        a = np.empty(1, dtype=SEG)
        a["start"], a["end"] = 0, size
        out[name] = a
    return out


def workspace_ungapped(contigs, pieces=12, gap=50000):
    """ungapped-style workspace: each contig in `pieces` blocks separated by assembly gaps
    (cf. the 282-interval contigs_ungapped.bed.gz of the reference tutorial)."""
    out = collections.OrderedDict()
    for name, size in contigs.items():
        edges = np.linspace(10000, size - 10000, pieces + 1).astype(np.int64)
        a = np.empty(pieces, dtype=SEG)
        a["start"] = edges[:-1] + gap // 2
        a["end"] = edges[1:] - gap // 2
        out[name] = a
    return out


def isochores_blocks(contigs, nclasses=8, block=1000000):
    """isochore tracks: class k owns every nclasses-th `block`-sized window of each contig.

    returns OrderedDict class_name -> OrderedDict contig -> SEG array."""
    out = collections.OrderedDict()
    for k in range(nclasses):
        per = collections.OrderedDict()
        for name, size in contigs.items():
            starts = np.arange(k * block, size, nclasses * block, dtype=np.int64)
            if len(starts) == 0:
                continue
            a = np.empty(len(starts), dtype=SEG)
            a["start"] = starts
            a["end"] = np.minimum(starts + block, size)
            per[name] = a
        out["iso%d" % k] = per
    return out


# Monte-Carlo samples of each BASELINE.json configuration (config4 / config5 are 8-GPU jobs: 12 500 / 125 000 per GPU)
CONFIG_SAMPLES = {"config1": 1000, "config2": 10000, "config3": 10000, "config4": 100000, "config5": 1000000,
                  "refdata": 10000}

# the reference's own integration-test data (test/data/*.bed.gz of the reference, kept as data fixtures under
# tests/golden/refdata: mouse ChIP-seq peaks, a workspace of 279 844 segments -- 6 600 to 21 000 per contig --, 7 annotation
# tracks; test/check_run.py, test/data/output_single.tsv:66-125 are its only published timings)
REFDATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "refdata")
REFDATA_TRACK = "Rela-120m"        # the 8 549-segment track: 30 samples/s in the reference's 2013 log


def refdata(track=REFDATA_TRACK):
    """the reference's test data set as a configuration: one segment track against the seven annotation tracks inside the
    fragmented workspace, read and prepared as gat-run.py does (IO.buildSegments / IO.applyIsochores, gat/IO.py:88-248)."""
    import gat_amd
    from gat_amd import IO
    d = REFDATA_DIR
    opts, _ = gat_amd.buildParser().parse_args(["--segments=%s" % os.path.join(d, "segments_single.bed.gz"),
                                               "--annotations=%s" % os.path.join(d, "annotations.bed.gz"),
                                               "--workspace=%s" % os.path.join(d, "workspace.bed.gz"), "--with-segment-tracks"])
    segments, annotations, workspaces, isochores = IO.buildSegments(opts)
    workspace = IO.applyIsochores(segments, annotations, workspaces, opts, isochores)
    tracks = [t for t in segments.tracks if track in t]
    if len(tracks) != 1:
        raise ValueError("refdata: segment track %r not found among %r" % (track, list(segments.tracks)))
    return dict(segments=segments[tracks[0]].asArrays(),
                annotations=[(t, annotations[t].asArrays()) for t in annotations.tracks],
                workspace=workspace.asArrays(), isochores=None, num_samples=10000, counter="nucleotide-overlap",
                segment_track=tracks[0])


def config(name, scale=1.0):
    """inputs of a BASELINE.json configuration.

    returns dict(segments=, annotations=[(track, dict)], workspace=, isochores=|None,
                 num_samples=, counter=)"""
    def n(x):
        return max(1, int(round(x * scale)))

    if name == "config1":      # chr22, 1k x 1 x 1k, 1000 samples
        return dict(segments=random_segments(CHR22, n(1000), 500, 11),
                    annotations=[("anno0", random_segments(CHR22, n(1000), 2000, 100))],
                    workspace=workspace_contigs(CHR22), isochores=None,
                    num_samples=1000, counter="nucleotide-overlap")
    if name == "config2":      # hg19, 10k x 1 x 10k, 10000 samples
        return dict(segments=random_segments(HG19, n(10000), 500, 11),
                    annotations=[("anno0", random_segments(HG19, n(10000), 2000, 100))],
                    workspace=workspace_contigs(HG19), isochores=None,
                    num_samples=10000, counter="nucleotide-overlap")
    if name == "config3":      # hg19 isochore-partitioned, 10k x 100 x 10k, 10000 samples
        return dict(segments=random_segments(HG19, n(10000), 500, 11),
                    annotations=[("anno%d" % i, random_segments(HG19, n(10000), 2000, 100 + i)) for i in range(100)],
                    workspace=workspace_contigs(HG19), isochores=isochores_blocks(HG19),
                    num_samples=10000, counter="nucleotide-overlap")
    if name == "config4":      # 100k x 1000 x 10k, 100000 samples (8 GPUs)
        return dict(segments=random_segments(HG19, n(100000), 500, 11),
                    annotations=[("anno%d" % i, random_segments(HG19, n(10000), 2000, 100 + i)) for i in range(1000)],
                    workspace=workspace_contigs(HG19), isochores=None,
                    num_samples=100000, counter="nucleotide-overlap")
    if name == "config5":      # density, 1M-interval annotation, ungapped workspace
        return dict(segments=random_segments(HG19, n(10000), 500, 11),
                    annotations=[("anno0", random_segments(HG19, n(1000000), 300, 100))],
                    workspace=workspace_ungapped(HG19), isochores=None,
                    num_samples=1000000, counter="nucleotide-density")
    if name == "refdata":      # the reference's own test data: a workspace of 13 000 segments per contig
        return refdata()
    raise ValueError("unknown config %r" % name)


def as_collections(cfg):
    """the inputs of a configuration as the host classes gat_amd.run() takes (what gat-run.py's fromSegments builds from
    BED files, gat/IO.py:88-248): (segments, annotations, workspace) with the isochores applied.  Returns them with the
    seconds the isochore split took."""
    import time
    import gat_amd

    def coll(tracks):
        c = gat_amd.IntervalCollection()
        for t, per in tracks:
            for contig, a in per.items():
                s = gat_amd.SegmentList(array=a)
                s.isNormalized = 1
                c.add(t, contig, s)
        return c

    segments = coll(cfg.get("segment_tracks") or [("merged", cfg["segments"])])
    annotations = coll(cfg["annotations"])
    workspaces = coll([("ws", cfg["workspace"])])
    workspaces.collapse()
    workspaces.restrict("collapsed")
    t0 = time.perf_counter()
    if cfg["isochores"]:
        isochores = coll(list(cfg["isochores"].items()))
        isochores.intersect(workspaces["collapsed"])
        workspaces.toIsochores(isochores, truncate=True)
        annotations.toIsochores(isochores, truncate=True)
        segments.toIsochores(isochores, truncate=False)
    return segments, annotations, workspaces["collapsed"], time.perf_counter() - t0


def small_genome():
    """4 small contigs, gapped workspace, 3 isochore classes, 3 annotation tracks; one (contig, isochore)
    unit without segments and one contig without annotations in track t1 (golden run_small_*)."""
    contigs = collections.OrderedDict([("chrA", 400000), ("chrB", 250000), ("chrC", 90000), ("chrD", 50000)])
    cfg = dict(segments=random_segments(contigs, 300, 120, 3),
               annotations=[("t%d" % i, random_segments(contigs, 150 + 40 * i, 400 + 150 * i, 50 + i)) for i in range(3)],
               workspace=workspace_ungapped(contigs, pieces=4, gap=4000),
               isochores=isochores_blocks(contigs, nclasses=3, block=30000))
    # make one (contig, isochore) unit segment-free and one contig annotation-free
    seg = cfg["segments"]
    seg["chrD"] = seg["chrD"][(seg["chrD"]["start"] // 30000) % 3 != 1]
    cfg["annotations"][1][1].pop("chrC", None)
    return contigs, cfg

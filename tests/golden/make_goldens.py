#!/usr/bin/env python
"""Generate the golden vectors in tests/golden/ from the REFERENCE ITSELF.

Run in the build container only (the reference never travels):

    bash tests/golden/build_reference.sh          # scratch build in /tmp/gatbuild, runs its tests
    PYTHONPATH=/tmp/gatbuild python tests/golden/make_goldens.py

Everything written here is data: inputs and the outputs the reference produced for them.
The files pin (a) oracle/gat_oracle.c and (b) the HIP path (tests/test_*).

Stream modes:
  mode 0  numpy.random.seed(seed) once, then the reference's own gat.run()      -> real reference run
  mode 1  numpy.random.seed((seed + sample_id*n_units + unit) mod 2^32) immediately before each
          sampler.sample(segs[iso], workspace[iso]) inside the reference's own gat.computeSample()
          (a duck-typed sampler wrapper does the re-seeding; sampler, interval algebra,
          fromIsochores and counters are the reference's)
"""
import collections
import hashlib
import json
import os
import sys

import numpy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))

import gat                                   # noqa: E402  (the reference, from PYTHONPATH)
import gat.Engine as Engine                  # noqa: E402
import gat.IO as IO                          # noqa: E402
from gat.SegmentList import SegmentList      # noqa: E402

from gat_amd import synthetic                # noqa: E402

assert "/root/repo" not in os.path.abspath(gat.__file__), "must import the reference, not the product"

SEG = synthetic.SEG
COUNTERS = collections.OrderedDict([
    ("nucleotide-overlap", Engine.CounterNucleotideOverlap),
    ("nucleotide-density", Engine.CounterNucleotideDensity),
    ("segment-overlap", Engine.CounterSegmentOverlap),
    ("segment-midoverlap", Engine.CounterSegmentMidpointOverlap),
    ("annotation-overlap", Engine.CounterAnnotationOverlap),
    ("annotation-midoverlap", Engine.CounterAnnotationMidpointOverlap),
])


def sl(pairs, normalize=False):
    return SegmentList(iter=[(int(a), int(b)) for a, b in pairs], normalize=normalize)


def arr(segmentlist):
    lst = segmentlist.asList()
    out = numpy.empty(len(lst), dtype=SEG)
    if lst:
        a = numpy.asarray(lst, dtype=numpy.int64)
        out["start"], out["end"] = a[:, 0], a[:, 1]
    return out


def pairs(a):
    return [(int(s), int(e)) for s, e in zip(a["start"], a["end"])]


# ------------------------------------------------------------------------------------------
# G1 interval algebra
def g1_algebra():
    rs = numpy.random.RandomState(1234)
    cases = []

    def rand_list(n, span, maxlen, allow_empty=True):
        st = rs.randint(0, span, size=n)
        ln = rs.randint(0 if allow_empty else 1, maxlen, size=n)
        return [(int(a), int(a + b)) for a, b in zip(st, ln)]

    # the reference's own fixtures (test/test_SegmentList.py:27-142, :206-227) re-expressed as data
    fixed = [
        [(x, x + 10) for x in range(0, 1000, 100)],
        [(x, x + 10) for x in range(100, 1100, 100)],
        [(0, 0)], [(0, 0), (0, 0)],
        [(0, i) for i in range(10)],
        [(x, x + 1000) for x in range(0, 1000, 100)],
        [(x, x + 100) for x in range(0, 1000, 100)],
        [(x, x + 100) for x in range(0, 1000, 10)],
        [(489, 589), (1966, 2066), (2786, 2886), (0, 0), (3889, 3972), (3998, 4098),
         (6441, 6541), (6937, 7054), (7392, 7492), (8154, 8254), (9046, 9146)],
        [],
    ]
    lists = fixed + [rand_list(rs.randint(1, 40), 2000, 120) for _ in range(60)]
    lists += [rand_list(rs.randint(1, 30), 200, 30) for _ in range(40)]
    # large coordinates (still < 2^31: the int32 lmin/lmax casts of gat/SegmentList.pyx:68-77 stay order-preserving)
    lists += [[(a + 2000000000, b + 2000000000) for a, b in rand_list(20, 5000, 300)] for _ in range(5)]
    for ss in lists:
        order = rs.permutation(len(ss))
        ss = [ss[i] for i in order]
        s = sl(ss)
        s.normalize()
        cases.append(dict(op="normalize", a=ss, expect=s.asList()))
        for d in (-1, 0, 1, 5, 50):
            s = sl(ss)
            s.merge(d)
            cases.append(dict(op="merge", a=ss, distance=d, expect=s.asList()))
    # test/test_SegmentList.py:218-227 testMergeNeighbours
    for y in range(0, 5):
        ss = [(x, x + 100 - y) for x in range(0, 1000, 100)]
        for d in range(0, y + 1):
            s = sl(ss)
            s.merge(d)
            cases.append(dict(op="merge", a=ss, distance=d, expect=s.asList()))

    def norm_list(n, span, maxlen):
        s = sl(rand_list(n, span, maxlen), normalize=True)
        return s.asList()

    pairs_ab = []
    a0 = [(x, x + 10) for x in range(0, 1000, 100)]
    # test/test_SegmentList.py:369-441 fixtures
    pairs_ab += [
        ([(0, 1000)], a0), (a0, a0[:]), ([(x, x + 10) for x in range(10, 1000, 100)], a0),
        ([(x, x + 10) for x in range(5, 1000, 100)], a0),
        (a0, [(x, x + 5) for x in range(0, 1000, 100)] + [(x + 5, x + 10) for x in range(0, 1000, 100)]),
        ([(x, x + 5) for x in range(500, 2000, 100)], a0),
        ([(0, 56)], [(0, 50), (75, 125)]), ([(0, 56)], [(0, 10)]),
        ([(0, 10), (10, 20)], [(0, 20)]), ([], a0), (a0, []),
    ]
    for _ in range(150):
        span = int(rs.choice([300, 2000, 20000]))
        pairs_ab.append((norm_list(rs.randint(1, 40), span, int(rs.choice([10, 80, 400]))),
                         norm_list(rs.randint(1, 40), span, int(rs.choice([10, 80, 400])))))
    for _ in range(5):
        pairs_ab.append(([(a + 2100000000, b + 2100000000) for a, b in norm_list(20, 3000, 200)],
                         [(a + 2100000000, b + 2100000000) for a, b in norm_list(20, 3000, 200)]))
    for a, b in pairs_ab:
        a = sorted(set(a))
        b = sorted(set(b))
        sa, sb = sl(a, normalize=True), sl(b, normalize=True)
        a, b = sa.asList(), sb.asList()
        x = sl(a, normalize=True); x.filter(sb)
        y = sl(a, normalize=True); y.intersect(sb)
        cases.append(dict(op="pair", a=a, b=b,
                          filter=x.asList(), intersect=y.asList(), sum_a=int(sa.sum()),
                          overlap=int(sa.overlapWithSegments(sb)),
                          isect_base=int(sa.intersectionWithSegments(sb)),
                          isect_mid=int(sa.intersectionWithSegments(sb, mode="midpoint")),
                          isect_base_rev=int(sb.intersectionWithSegments(sa)),
                          isect_mid_rev=int(sb.intersectionWithSegments(sa, mode="midpoint"))))
    # getInsertionPoint (test/test_SegmentList.py:167-195) + trim_ends
    for ss in ([(x, x + 10) for x in range(0, 100, 10)], [(x, x + 10) for x in range(0, 100, 20)],
               [(x, x + 10) for x in range(10, 100, 20)], norm_list(15, 400, 30)):
        s = sl(ss, normalize=True)
        pts = list(range(0, max(e for _, e in s.asList()) + 12))
        cases.append(dict(op="insertion_point", a=s.asList(), points=pts,
                          expect=[int(s.getInsertionPoint(p, p + 1)) for p in pts]))
    for _ in range(120):
        ss = norm_list(rs.randint(1, 12), 400, 40)
        if not ss:
            continue
        s = sl(ss, normalize=True)
        tot = s.sum()
        if tot < 2:
            continue
        lo, hi = ss[0][0], ss[-1][1]
        pos = int(rs.randint(max(0, lo - 3), hi + 3))
        size = int(rs.randint(1, tot))
        fwd = int(rs.randint(0, 2))
        s.trim_ends(pos, size, fwd)
        cases.append(dict(op="trim_ends", a=ss, pos=pos, size=size, forward=fwd, expect=s.asList()))
    # getLengthDistribution incl. auto bucket size and the ValueError edge (len == nbuckets*bucket)
    for ss, b, nb in ([[(0, 10), (20, 25), (30, 130)], 0, 1000], [[(0, 10), (20, 25), (30, 130)], 1, 1000],
                      [[(0, 10), (20, 25), (30, 130)], 7, 1000], [[(0, 10), (20, 25), (30, 130)], 0, 50],
                      [[(0, 100), (200, 250)], 0, 100], [[(0, 101), (200, 250)], 0, 100],
                      [[(0, 300000), (400000, 400010), (500000, 650001)], 0, 100000],
                      [norm_list(30, 100000, 3000), 0, 1000], [norm_list(30, 100000, 3000), 0, 100000]):
        s = sl(ss, normalize=True)
        try:
            h, bs = s.getLengthDistribution(b, nb)
            nz = numpy.flatnonzero(h)
            cases.append(dict(op="length_distribution", a=s.asList(), bucket_size=b, nbuckets=nb,
                              bucket_size_out=int(bs), nonzero=[int(i) for i in nz], counts=[int(h[i]) for i in nz]))
        except ValueError:
            cases.append(dict(op="length_distribution", a=s.asList(), bucket_size=b, nbuckets=nb, error="ValueError"))
    with open(os.path.join(HERE, "algebra.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))
    print("G1 algebra: %d cases" % len(cases))


# ------------------------------------------------------------------------------------------
# G2 numpy legacy RandomState: seed -> (lo, hi, value) triples in GAT's ranges
def g2_rng():
    seeds = [0, 1, 7, 42, 123456789, 4294967295]
    rs = numpy.random.RandomState(99)
    n = 4096
    los = numpy.zeros((len(seeds), n), dtype=numpy.int64)
    his = numpy.zeros((len(seeds), n), dtype=numpy.int64)
    vals = numpy.zeros((len(seeds), n), dtype=numpy.int64)
    raw = numpy.zeros((len(seeds), 1300), dtype=numpy.uint32)
    for si, seed in enumerate(seeds):
        kind = rs.randint(0, 6, size=n)
        lo = numpy.where(kind == 0, 1, numpy.where(kind == 1, 0, rs.randint(-100000, 250000000, size=n)))
        width = numpy.where(kind == 0, rs.randint(1, 20000, size=n),
                            numpy.where(kind == 1, rs.randint(1, 3100000000, size=n),
                                        numpy.where(kind == 2, 1, numpy.where(kind == 3, 2, rs.randint(1, 300000, size=n)))))
        hi = lo + width
        numpy.random.seed(seed)
        for i in range(n):
            vals[si, i] = numpy.random.randint(int(lo[i]), int(hi[i]))
        los[si], his[si] = lo, hi
        numpy.random.seed(seed)
        raw[si] = numpy.random.randint(0, 4294967296, size=1300, dtype=numpy.uint32) if False else \
            numpy.array([numpy.random.randint(0, 4294967296) for _ in range(1300)], dtype=numpy.uint32)
    numpy.savez_compressed(os.path.join(HERE, "rng.npz"), seeds=numpy.array(seeds, dtype=numpy.int64),
                           lo=los, hi=his, value=vals, raw=raw)
    print("G2 rng: %d seeds x %d draws (+1300 raw u32 each, spanning two twists)" % (len(seeds), n))


# ------------------------------------------------------------------------------------------
# G3 SamplerAnnotator.sample KATs (shapes of test/benchmark_gat.py:857-1170, plus dense/long cases)
def sampler_shapes():
    shapes = collections.OrderedDict()
    n, ss = 10, 100
    shapes["segmented_small_gap"] = ([(x, x + 990) for x in range(0, 1000 * n, 1000)], [(x, x + ss) for x in range(0, 1000 * n, 1000)])
    shapes["segmented_partial_overlap"] = ([(x, x + ss) for x in range(ss // 2, n * ss, 2 * ss)], [(x, x + ss) for x in range(0, n * 2 * ss, 2 * ss)])
    shapes["two_ws_unequal"] = ([(0, 50), (75, 100)], [(0, 50)])
    shapes["two_ws_equal"] = ([(0, 50), (55, 105)], [(0, 50)])
    shapes["two_ws_many"] = ([(0, 50), (55, 105)], [(x, x + 5) for x in range(0, 50, 10)])
    shapes["segmented_large_gap"] = ([(x, x + 900) for x in range(0, 1000 * n, 1000)], [(x, x + ss) for x in range(0, 1000 * n, 1000)])
    shapes["single_ws"] = ([(0, 10000)], [(x, x + ss) for x in range(0, 1000 * n, 1000)])
    shapes["single_ws_offset"] = ([(10000, 20000)], [(x, x + ss) for x in range(10000, 10000 + 1000 * n, 1000)])
    shapes["single_ws_single_seg"] = ([(0, 10000)], [(4500, 5500)])
    shapes["full_ws"] = ([(0, 100)], [(0, 200)])
    shapes["small_ws"] = ([(0, 100)], [(0, 50)])
    shapes["tiny_ws"] = ([(0, 12)], [(0, 4)])
    shapes["small_ws_many"] = ([(0, 100)], [(x, x + 5) for x in range(0, 100, 10)])
    shapes["segmented_2x"] = ([(x, x + 2 * ss) for x in range(0, 1000 * n, 1000)], [(x, x + ss) for x in range(0, 1000 * n, 1000)])
    rs = numpy.random.RandomState(5)
    # dense: 40 % of a 20 kb workspace covered -> many overlaps, repeated consolidation, trims
    st = numpy.sort(rs.randint(0, 20000, size=80))
    shapes["dense_40pct"] = ([(0, 8000), (8100, 20000)], sl([(int(a), int(a + 100)) for a in st], normalize=True).asList())
    # very dense: may exhaust the 20 unsuccessful rounds (gat/Engine.pyx:570-572)
    st = numpy.sort(rs.randint(0, 3000, size=60))
    shapes["dense_90pct"] = ([(100, 3100)], sl([(int(a), int(a + 80)) for a in st], normalize=True).asList())
    # long segments -> automatic bucket_size > 1 (second length draw, gat/Engine.pyx:432-433)
    st = numpy.sort(rs.randint(0, 90000000, size=40))
    ln = rs.randint(1000, 350000, size=40)
    shapes["long_segments_bucket4"] = ([(0, 50000000), (50010000, 100000000)],
                                       sl([(int(a), int(a + b)) for a, b in zip(st, ln)], normalize=True).asList())
    # segments partly outside the workspace / hanging off coordinate 0
    shapes["outside_ws"] = ([(50, 1000), (1500, 1600)], [(0, 100), (400, 450), (990, 1510), (1590, 1700), (3000, 3100)])
    # chr22-size case (config 1 shape)
    segs = synthetic.random_segments(synthetic.CHR22, 1000, 500, 11)["chr22"]
    shapes["chr22_1k"] = ([(0, 51304566)], pairs(segs))
    return shapes


def g3_sampler():
    out = []
    for name, (ws, segs) in sampler_shapes().items():
        w = sl(ws, normalize=True)
        s = sl(segs, normalize=True)
        for bucket_size, nbuckets in ((0, 100000), (1, 100000), (3, 200000)):
            if name == "chr22_1k" and bucket_size != 0:
                continue
            if name.startswith("long_segments") and bucket_size == 1:
                continue       # ValueError case, captured below
            runs = []
            for seed in (1, 2, 3, 1000003, 4294967295):
                sampler = Engine.SamplerAnnotator(bucket_size=bucket_size, nbuckets=nbuckets)
                numpy.random.seed(seed)
                r = sampler.sample(s, w)
                # position in the stream after the call: next raw 32-bit output
                nxt = int(numpy.random.randint(0, 4294967296))
                lst = r.asList()
                rec = dict(seed=seed, n=len(lst), sum=int(r.sum()), next_u32=nxt,
                           sha256=hashlib.sha256(arr(r).tobytes()).hexdigest())
                if len(lst) <= 200:
                    rec["out"] = lst
                else:
                    rec["head"], rec["tail"] = lst[:16], lst[-16:]
                runs.append(rec)
            out.append(dict(name=name, workspace=w.asList(), segments=s.asList(),
                            bucket_size=bucket_size, nbuckets=nbuckets, runs=runs))
    # ValueError: segment longer than nbuckets*bucket_size (gat/SegmentList.pyx:1170-1182)
    w = sl([(0, 1000000)], normalize=True)
    s = sl([(0, 500), (1000, 201000)], normalize=True)
    try:
        Engine.SamplerAnnotator(bucket_size=1, nbuckets=100000).sample(s, w)
        raise SystemExit("expected ValueError")
    except ValueError:
        out.append(dict(name="value_error", workspace=w.asList(), segments=s.asList(), bucket_size=1, nbuckets=100000,
                        error="ValueError"))
    with open(os.path.join(HERE, "sampler.json"), "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("G3 sampler: %d (shape, bucket) cases x 5 seeds" % len(out))


# ------------------------------------------------------------------------------------------
# G4/G7 batch seam: count matrices in both stream modes from the reference's gat.run / computeSample
def ref_collection(name, tracks):
    coll = Engine.IntervalCollection(name=name)
    for track, per in tracks:
        for contig, a in per.items():
            coll.add(track, contig, sl(pairs(a), normalize=True))
    return coll


class ReseedingSampler(object):
    """duck-typed sampler (gat/__init__.py:541 calls only .sample): re-seeds the global legacy
    RandomState per work unit, then delegates to the reference sampler."""

    def __init__(self, inner, segs, base_seed):
        self.inner = inner
        self.keys = list(segs.keys())
        self.unit_of = dict((id(segs[k]), i) for i, k in enumerate(self.keys))
        self.n_units = len(self.keys)
        self.base_seed = base_seed
        self.sample_id = 0
        self.records = []

    def sample(self, segments, workspace):
        u = self.unit_of[id(segments)]
        numpy.random.seed((self.base_seed + self.sample_id * self.n_units + u) & 0xFFFFFFFF)
        r = self.inner.sample(segments, workspace)
        self.records.append((self.keys[u], r))
        return r


def flat_problem(segs, workspace, annotations, bucket_size, nbuckets):
    """flatten the reference's structures exactly as gat.computeSample walks them."""
    units = list(segs.keys())
    contig_annotations = annotations.clone()
    contig_annotations.fromIsochores()
    contig_workspace = workspace.clone()
    contig_workspace.fromIsochores()
    seg_arrays, ws_arrays, unit_contig, contigs = [], [], [], []
    merge = 0
    for u in units:
        sa = arr(segs[u]) if u in segs else numpy.empty(0, dtype=SEG)
        wa = arr(workspace[u]) if u in workspace else numpy.empty(0, dtype=SEG)
        seg_arrays.append(sa)
        ws_arrays.append(wa)
        key = u.strip()
        if "." in key and key != ".":
            contig = key.split(".")[0]
            merge = 1
        else:
            contig = key
        skipped = len(sa) == 0 or len(wa) == 0
        if skipped:
            unit_contig.append(-1)
            continue
        if contig not in contigs:
            contigs.append(contig)
        unit_contig.append(contigs.index(contig))
    tracks = list(annotations.tracks)
    anno_arrays = []
    for t in tracks:
        for c in contigs:
            anno_arrays.append(arr(contig_annotations[t][c]) if c in contig_annotations[t] else numpy.empty(0, dtype=SEG))
    cws_nseg = [len(contig_workspace[c]) if c in contig_workspace else 0 for c in contigs]

    def cat(lst):
        return numpy.concatenate(lst) if lst else numpy.empty(0, dtype=SEG)

    def off(lst):
        return numpy.concatenate([[0], numpy.cumsum([len(x) for x in lst])]).astype(numpy.int64)

    return dict(n_units=len(units), unit_names=numpy.array(units), segs=cat(seg_arrays), seg_off=off(seg_arrays),
                ws=cat(ws_arrays), ws_off=off(ws_arrays), unit_contig=numpy.array(unit_contig, dtype=numpy.int32),
                n_contigs=len(contigs), contig_names=numpy.array(contigs), merge_contigs=merge,
                n_tracks=len(tracks), track_names=numpy.array(tracks), annos=cat(anno_arrays), anno_off=off(anno_arrays),
                cws_nseg=numpy.array(cws_nseg, dtype=numpy.int64), bucket_size=bucket_size, nbuckets=nbuckets)


class Options(object):
    truncate_segments_to_workspace = False
    output_stats = []
    output_bed = []


def build_reference_inputs(cfg, truncate_segments=False):
    """the part of IO.buildSegments/applyIsochores (gat/IO.py:88-293) after file parsing, done by the reference."""
    segments = ref_collection("segments", [("merged", cfg["segments"])])
    annotations = ref_collection("annotations", cfg["annotations"])
    workspaces = ref_collection("workspaces", [("ws", cfg["workspace"])])
    workspaces.collapse()
    workspaces.restrict("collapsed")
    isochores = None
    if cfg.get("isochores"):
        isochores = ref_collection("isochores", list(cfg["isochores"].items()))
        isochores.intersect(workspaces["collapsed"])
    opts = Options()
    opts.truncate_segments_to_workspace = truncate_segments
    workspace = IO.applyIsochores(segments, annotations, workspaces, opts, isochores)
    return segments, annotations, workspace


def run_case(name, cfg, counters, num_samples, seed, bucket_size=0, nbuckets=100000, keep_lists=True,
             truncate_segments=False, sampler_class=None):
    segments, annotations, workspace = build_reference_inputs(cfg, truncate_segments)
    counter_objs = [COUNTERS[c]() for c in counters]
    wsgen = Engine.UnconditionalWorkspace()
    flat = flat_problem(segments["merged"], workspace, annotations, bucket_size, nbuckets)
    tracks = list(annotations.tracks)

    # ---- mode 0: the reference's own gat.run on one global stream
    numpy.random.seed(seed)
    sampler_class = sampler_class or Engine.SamplerAnnotator
    sampler = sampler_class(bucket_size=bucket_size, nbuckets=nbuckets)
    results = gat.run(segments, annotations, workspace, sampler, counter_objs, wsgen,
                      num_samples=num_samples, pseudo_count=1.0)
    counts0 = numpy.zeros((len(counters), len(tracks), num_samples), dtype=numpy.float64)
    observed = numpy.zeros((len(counters), len(tracks)), dtype=numpy.float64)
    stats = []
    rows = []
    for r in results:
        k, a = counters.index(r.counter), tracks.index(r.annotation)
        counts0[k, a] = r.samples
        observed[k, a] = r.observed
        stats.append([k, a, r.expected, r.stddev, r.fold, r.pvalue])
        rows.append(str(r))

    # ---- mode 1: the reference's computeSample with per-unit re-seeding
    segs = segments["merged"]
    contig_annotations = annotations.clone()
    contig_annotations.fromIsochores()
    contig_workspace = workspace.clone()
    contig_workspace.fromIsochores()
    rsampler = ReseedingSampler(sampler_class(bucket_size=bucket_size, nbuckets=nbuckets), segs, seed)
    counts1 = numpy.zeros((len(counters), len(tracks), num_samples), dtype=numpy.float64)
    lists, list_off, sha = [], [0], hashlib.sha256()
    for x in range(num_samples):
        rsampler.sample_id = x
        rsampler.records = []
        w = gat.WorkData("merged", x, rsampler, segs, annotations, contig_annotations, workspace, contig_workspace, counter_objs)
        res = gat.computeSample((w, None, None, None))
        for k in range(len(counters)):
            for a, t in enumerate(tracks):
                counts1[k, a, x] = res[k][t]
        d = Engine.IntervalDictionary()
        for key, r in rsampler.records:
            d.add(key, r.clone())
        d.fromIsochores()
        for c in flat["contig_names"]:
            a = arr(d[str(c)]) if str(c) in d else numpy.empty(0, dtype=SEG)
            sha.update(a.tobytes())
            if keep_lists:
                lists.append(a)
                list_off.append(list_off[-1] + len(a))
    # the reference's result rows (24 columns, gat/Engine.pyx:1950-1974) for the mode-1 count matrix
    rows1 = []
    for k, cname in enumerate(counters):
        for a, t in enumerate(tracks):
            r = Engine.AnnotatorResultExtended(track="merged", annotation=t, counter=cname, observed=observed[k, a],
                                               samples=list(counts1[k, a]), track_segments=segments["merged"],
                                               annotation_segments=annotations[t], workspace=workspace,
                                               reference=None, pseudo_count=1.0)
            rows1.append(str(r))
    out = dict(flat)
    out.update(rows_mode1=numpy.array(rows1), sampler=(1 if sampler_class is Engine.SamplerSegments else 0))
    out.update(seed=seed, num_samples=num_samples, counters=numpy.array(counters),
               counts_mode0=counts0, counts_mode1=counts1, observed=observed,
               stats_mode0=numpy.array(stats, dtype=numpy.float64), rows_mode0=numpy.array(rows),
               samples_sha256_mode1=sha.hexdigest())
    if keep_lists:
        out["samples_mode1"] = numpy.concatenate(lists) if lists else numpy.empty(0, dtype=SEG)
        out["samples_off_mode1"] = numpy.array(list_off, dtype=numpy.int64)
    numpy.savez_compressed(os.path.join(HERE, "run_%s.npz" % name), **out)
    print("G4 run_%s: units=%d contigs=%d tracks=%d S=%d  mean counts mode0=%s mode1=%s" % (
        name, flat["n_units"], flat["n_contigs"], flat["n_tracks"], num_samples,
        numpy.round(counts0.mean(axis=2).ravel()[:4], 2), numpy.round(counts1.mean(axis=2).ravel()[:4], 2)))


small_genome = synthetic.small_genome


def g4_runs():
    all6 = list(COUNTERS.keys())
    # P1: BASELINE config 1 (chr22, 1k x 1 x 1k), 1000 samples
    cfg = synthetic.config("config1")
    run_case("config1", cfg, ["nucleotide-overlap"], 1000, 7, keep_lists=False)
    # P2: small genome with isochores, gapped workspace, all six counters, empty units
    _, cfg = small_genome()
    run_case("small_isochores", cfg, all6, 60, 11)
    # P2b: same without isochores (keys without '.': fromIsochores passes lists through)
    cfg2 = dict(cfg)
    cfg2["isochores"] = None
    run_case("small_contigs", cfg2, all6, 60, 12)
    # P2c: truncate segments to the workspace (--truncate-segments-to-workspace)
    run_case("small_isochores_truncated", cfg, ["nucleotide-overlap", "segment-overlap"], 30, 13, truncate_segments=True)
    # P2d: SamplerSegments (gat/Engine.pyx:653): fixed number of placements, merged by fromIsochores
    run_case("small_isochores_sampler_segments", cfg, all6, 40, 14, sampler_class=Engine.SamplerSegments)
    # P3: BASELINE config 2 shape (hg19, 10k x 2 x 10k), 12 samples
    cfg = synthetic.config("config2")
    cfg["annotations"].append(("anno1", synthetic.random_segments(synthetic.HG19, 10000, 2000, 101)))
    run_case("config2_s12", cfg, ["nucleotide-overlap", "segment-overlap"], 12, 21, keep_lists=False)
    # P4: density counter on a multi-segment (ungapped-style) workspace
    contigs = collections.OrderedDict(list(synthetic.HG19.items())[18:22])
    cfg = dict(segments=synthetic.random_segments(contigs, 800, 500, 31),
               annotations=[("dense", synthetic.random_segments(contigs, 20000, 300, 32))],
               workspace=synthetic.workspace_ungapped(contigs, pieces=7), isochores=None)
    run_case("density_ungapped", cfg, ["nucleotide-density", "nucleotide-overlap"], 40, 33, keep_lists=False)
    # P5: dense segments in a small workspace (several consolidation rounds, trims)
    contigs = collections.OrderedDict([("c1", 60000), ("c2", 30000)])
    cfg = dict(segments=synthetic.random_segments(contigs, 260, 150, 41),
               annotations=[("a", synthetic.random_segments(contigs, 100, 300, 42))],
               workspace=synthetic.workspace_ungapped(contigs, pieces=3, gap=2000), isochores=None)
    run_case("dense", cfg, ["nucleotide-overlap", "annotation-overlap"], 100, 43)
    # P6: long segments -> bucket_size > 1
    contigs = collections.OrderedDict(list(synthetic.HG19.items())[:3])
    cfg = dict(segments=synthetic.random_segments(contigs, 400, 60000, 51),
               annotations=[("a", synthetic.random_segments(contigs, 3000, 5000, 52))],
               workspace=synthetic.workspace_contigs(contigs), isochores=None)
    run_case("long_segments", cfg, ["nucleotide-overlap"], 30, 53)


# ------------------------------------------------------------------------------------------
# G6 enrichment statistics (gat/Engine.pyx:1635-1718) incl. the reference's own KATs
def g6_stats():
    rs = numpy.random.RandomState(77)
    cases = []

    def one(observed, samples, pseudo=1.0):
        r = Engine.AnnotatorResult("track", "annotation", "counter", observed, samples, reference=None, pseudo_count=pseudo)
        text = str(r).split("\t")
        cases.append(dict(observed=float(observed), samples=[float(x) for x in samples], pseudo_count=pseudo,
                          expected=r.expected, stddev=r.stddev, fold=r.fold, pvalue=r.pvalue,
                          lower95=float(text[4]), upper95=float(text[5]), row=text[2:]))

    # test/test_gat.py:272-284
    one(16, [0] * 66 + [1] * 2 + [2] * 20 + [3] * 1 + [4] * 6 + [6] * 2 + [8] * 2 + [16] * 1)
    # test/test_gat.py:239-270
    for y in range(1, 10):
        samples = [1] * y + [0] * (10 - y)
        for s in (0, 1):
            one(s, samples)
    for _ in range(60):
        n = int(rs.choice([1, 2, 5, 19, 20, 21, 100, 1000]))
        lam = float(rs.choice([0.5, 3, 50, 5000]))
        samples = rs.poisson(lam, size=n)
        obs = int(rs.choice([0, int(lam), int(samples.min()), int(samples.max()), int(samples.max()) + 5, int(numpy.median(samples))]))
        one(obs, [int(x) for x in samples], pseudo=float(rs.choice([1.0, 0.0, 0.5])))
    for _ in range(10):
        samples = rs.gamma(2.0, 0.15, size=100)
        one(float(rs.choice(samples)) if rs.rand() < 0.5 else float(rs.gamma(2.0, 0.15)), [float(x) for x in samples], pseudo=0.0)
    one(0, [0] * 50)         # expected == 0 -> fold 1.0

    # with a reference result (gat/Engine.pyx:1660-1700: expected, CI and the p-value are scaled by reference.fold)
    class Ref(object):
        def __init__(self, fold):
            self.fold = fold
    for _ in range(24):
        n = int(rs.choice([5, 20, 100, 1000]))
        samples = rs.poisson(float(rs.choice([3, 50, 5000])), size=n)
        obs = int(rs.choice([0, int(samples.mean()), int(samples.max()) + 3]))
        fold = float(rs.choice([0.5, 1.0, 1.7, 4.0]))
        r = Engine.AnnotatorResult("track", "annotation", "counter", obs, samples, reference=Ref(fold), pseudo_count=1.0)
        text = str(r).split("\t")
        cases.append(dict(observed=float(obs), samples=[float(x) for x in samples], pseudo_count=1.0, reference_fold=fold,
                          expected=r.expected, stddev=r.stddev, fold=r.fold, pvalue=r.pvalue,
                          lower95=float(text[4]), upper95=float(text[5]), row=text[2:]))
    with open(os.path.join(HERE, "stats.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))
    print("G6 stats: %d cases" % len(cases))


# ------------------------------------------------------------------------------------------
# G5 the command line tool end to end: the reference's scripts/gat-run.py (option parsing, BED
# reading, isochores, gat.run, BH q-values, table output) with the per-unit re-seeding patched into
# gat.computeSample, so that its table is the one the GPU build has to print byte for byte.
def write_bed(path, tracks, with_track_lines=True):
    with open(path, "w") as f:
        for name, per in tracks:
            if with_track_lines:
                f.write("track name=%s\n" % name)
            for contig, a in per.items():
                for s, e in pairs(a):
                    f.write("%s\t%i\t%i\n" % (contig, s, e))


def reference_cli():
    """the reference's scripts/gat-run.py as a module, and gat.computeSample wrapped so that every work unit re-seeds the
    global stream (the per-unit contract; segment tracks take disjoint ranges of unit streams)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gat_run_ref", os.path.join(os.path.dirname(gat.__file__), "..", "scripts", "gat-run.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    original = gat.computeSample
    state = dict(track=None, base=0, n_units=0, sampler=None)

    def patched(args):
        w = args[0]
        if state["track"] != w.track:                      # next segment track: disjoint unit streams
            if state["track"] is not None:
                state["base"] = (state["base"] + state["num_samples"] * state["n_units"]) & 0xFFFFFFFF
            state["track"] = w.track
            state["sampler"] = ReseedingSampler(w.sampler, w.segments, state["base"])
            state["n_units"] = state["sampler"].n_units
        rs = state["sampler"]
        rs.base_seed = state["base"]
        rs.sample_id = int(w.sample_id)
        rs.records = []
        return original((w._replace(sampler=rs),) + tuple(args[1:]))

    return mod, state, patched, original


def g5s_sample_file():
    """--sample-file (gat/__init__.py:952-961): the reference loads the files into a SamplesFile (track name from the
    pattern's regex), requires --output-samples-pattern, does NOT write sample files then (:977) -- and samples afresh all
    the same (UnconditionalSampler never looks at self.samples).  Inputs: the sample files the patterns run of g5 wrote."""
    cli_dir = os.path.join(HERE, "cli")
    pat = os.path.join(cli_dir, "aux", "patterns")
    mod, state, patched, original = reference_cli()
    files = [os.path.join(pat, "samples_segA.bed"), os.path.join(pat, "samples_segB.bed")]
    before = [open(f).read() for f in files]
    base = ["gat-run.py", "--segments=%s" % os.path.join(cli_dir, "segments.bed"),
            "--annotations=%s" % os.path.join(cli_dir, "annotations.bed"),
            "--workspace=%s" % os.path.join(cli_dir, "workspace.bed"),
            "--isochores=%s" % os.path.join(cli_dir, "isochores.bed"), "--with-segment-tracks",
            "--num-samples=5", "--random-seed=21", "--counter=nucleotide-overlap",
            "--log=%s" % os.path.join(cli_dir, "ref.log")]
    gat.computeSample = patched
    try:
        out = os.path.join(cli_dir, "aux", "expected_sample_file.tsv")
        state.update(track=None, base=21, n_units=0, sampler=None, num_samples=5)
        mod.main(base + ["--stdout=%s" % out, "--output-samples-pattern=%s" % os.path.join(pat, "samples_%s.bed"),
                         "--sample-file=%s" % os.path.join(pat, "samples_seg*.bed")])
        lines = [l for l in open(out) if not l.startswith("#")]
        with open(out, "w") as f:
            f.writelines(lines)
        assert [open(f).read() for f in files] == before, "the reference rewrote the sample files"
        errors = {}
        for name, extra in (("no_pattern", ["--sample-file=%s" % files[0]]),
                            ("pattern_does_not_match", ["--sample-file=%s" % files[0],
                                                        "--output-samples-pattern=%s" % os.path.join(pat, "other_%s.txt")])):
            state.update(track=None, base=21, n_units=0, sampler=None, num_samples=5)
            try:
                mod.main(base + ["--stdout=%s" % os.path.join(cli_dir, "aux", "tmp.tsv")] + extra)
                errors[name] = None
            except Exception as e:                             # noqa: BLE001
                errors[name] = type(e).__name__
        if os.path.exists(os.path.join(cli_dir, "aux", "tmp.tsv")):
            os.remove(os.path.join(cli_dir, "aux", "tmp.tsv"))
        with open(os.path.join(cli_dir, "aux", "sample_file_errors.json"), "w") as f:
            json.dump(errors, f, indent=1)
        print("G5s --sample-file: %d rows; errors %s" % (len(lines) - 1, errors))
    finally:
        gat.computeSample = original
    if os.path.exists(os.path.join(cli_dir, "ref.log")):
        os.remove(os.path.join(cli_dir, "ref.log"))


def g5_cli():
    cli_dir = os.path.join(HERE, "cli")
    os.makedirs(cli_dir, exist_ok=True)
    contigs, cfg = synthetic.small_genome()
    seg2 = synthetic.random_segments(contigs, 120, 200, 77)
    write_bed(os.path.join(cli_dir, "segments.bed"), [("segA", cfg["segments"]), ("segB", seg2)])
    write_bed(os.path.join(cli_dir, "annotations.bed"), cfg["annotations"])
    write_bed(os.path.join(cli_dir, "workspace.bed"), [("ws", cfg["workspace"])], with_track_lines=False)
    write_bed(os.path.join(cli_dir, "isochores.bed"), list(cfg["isochores"].items()))
    mod, state, patched, original = reference_cli()

    cases = collections.OrderedDict([
        ("default", ["--num-samples=100", "--random-seed=5"]),
        ("isochores_track_order", ["--num-samples=80", "--random-seed=6", "--isochores=%s" % os.path.join(cli_dir, "isochores.bed"),
                                   "--order=track", "--counter=segment-overlap"]),
        ("segment_tracks", ["--num-samples=60", "--random-seed=7", "--with-segment-tracks", "--order=annotation",
                            "--qvalue-method=holm"]),
        ("density_truncated", ["--num-samples=50", "--random-seed=8", "--counter=nucleotide-density",
                               "--truncate-segments-to-workspace", "--order=pvalue", "--pseudo-count=0.5"]),
        # workspace generators (gat/Engine.pyx:2093-2153); the conditional ones need build_reference.sh's str(annoid) fix
        ("segment_centered_expansion", ["--num-samples=40", "--random-seed=10", "--conditional=segment-centered",
                                        "--conditional-expansion=3", "--order=annotation"]),
        ("segment_centered_extension", ["--num-samples=40", "--random-seed=11", "--conditional=segment-centered",
                                        "--conditional-extension=700", "--conditional-expansion=1", "--with-segment-tracks",
                                        "--order=track", "--counter=nucleotide-density"]),
        ("storey_norm", ["--num-samples=50", "--random-seed=14", "--qvalue-method=storey", "--pvalue-method=norm",
                         "--with-segment-tracks"]),
        ("storey_lambda", ["--num-samples=50", "--random-seed=15", "--qvalue-method=storey", "--qvalue-lambda=0.3",
                           "--counter=segment-overlap", "--order=qvalue"]),
        ("cooccurance", ["--num-samples=30", "--random-seed=12", "--conditional=cooccurance", "--order=annotation"]),
        ("annotation_centered", ["--num-samples=30", "--random-seed=13", "--conditional=annotation-centered",
                                 "--conditional-expansion=1.5", "--order=annotation",
                                 "--isochores=%s" % os.path.join(cli_dir, "isochores.bed")]),
        # point annotations (gat/IO.py:134-135, gat/PositionList.pyx:288-336, :490-540): the two counters that call into
        # PositionList run; every other counter, isochores and --truncate-workspace-to-annotations end in a TypeError
        ("points_midpoint", ["--num-samples=40", "--random-seed=16", "--annotations-to-points=midpoint",
                             "--counter=annotation-overlap"]),
        ("points_start_tracks", ["--num-samples=30", "--random-seed=17", "--annotations-to-points=start",
                                 "--counter=annotation-midoverlap", "--with-segment-tracks", "--order=track"]),
        ("points_end", ["--num-samples=30", "--random-seed=18", "--annotations-to-points=end",
                        "--counter=annotation-overlap", "--order=annotation", "--truncate-segments-to-workspace"]),
    ])
    # the reference's own integration-test data (test/data/*.bed.gz, test/check_run.py): real mouse ChIP-seq
    # intervals, 279 844 workspace segments; copied as data fixtures into tests/golden/refdata/
    refdata = os.path.join(HERE, "refdata")
    ref_case = ["--num-samples=60", "--random-seed=9", "--with-segment-tracks", "--order=track"]
    gat.computeSample = patched
    try:
        out = os.path.join(refdata, "expected_mode1_s60.tsv")
        argv = ["gat-run.py", "--segments=%s" % os.path.join(refdata, "segments_single.bed.gz"),
                "--annotations=%s" % os.path.join(refdata, "annotations.bed.gz"),
                "--workspace=%s" % os.path.join(refdata, "workspace.bed.gz"),
                "--stdout=%s" % out, "--log=%s" % os.path.join(cli_dir, "ref.log")] + ref_case
        state.update(track=None, base=9, n_units=0, sampler=None, num_samples=60)
        mod.main(argv)
        lines = [l for l in open(out) if not l.startswith("#")]
        with open(out, "w") as f:
            f.writelines(lines)
        print("G5 refdata: %d rows" % (len(lines) - 1))
        for name, extra in cases.items():
            out = os.path.join(cli_dir, "expected_%s.tsv" % name)
            argv = ["gat-run.py", "--segments=%s" % os.path.join(cli_dir, "segments.bed"),
                    "--annotations=%s" % os.path.join(cli_dir, "annotations.bed"),
                    "--workspace=%s" % os.path.join(cli_dir, "workspace.bed"),
                    "--stdout=%s" % out, "--log=%s" % os.path.join(cli_dir, "ref.log")] + extra
            seed = int([x for x in extra if x.startswith("--random-seed")][0].split("=")[1])
            ns = int([x for x in extra if x.startswith("--num-samples")][0].split("=")[1])
            state.update(track=None, base=seed, n_units=0, sampler=None, num_samples=ns)
            mod.main(argv)
            lines = [l for l in open(out) if not l.startswith("#")]
            with open(out, "w") as f:
                f.writelines(lines)
            print("G5 cli %s: %d rows" % (name, len(lines) - 1))
        # --output-counts-pattern / --output-samples-pattern / --output-tables-pattern (gat/__init__.py:1072-1086,
        # :515-559; gat/IO.py:497-503): two counters -> one table file per counter, the count matrix per counter, the
        # sampled lists per segment track at isochore level
        pat = os.path.join(cli_dir, "aux", "patterns")
        os.makedirs(pat, exist_ok=True)
        argv = ["gat-run.py", "--segments=%s" % os.path.join(cli_dir, "segments.bed"),
                "--annotations=%s" % os.path.join(cli_dir, "annotations.bed"),
                "--workspace=%s" % os.path.join(cli_dir, "workspace.bed"),
                "--isochores=%s" % os.path.join(cli_dir, "isochores.bed"), "--with-segment-tracks",
                "--num-samples=5", "--random-seed=21", "--counter=nucleotide-overlap", "--counter=segment-overlap",
                "--output-tables-pattern=%s" % os.path.join(pat, "table_%s.tsv"),
                "--output-counts-pattern=%s" % os.path.join(pat, "counts_%s.tsv"),
                "--output-samples-pattern=%s" % os.path.join(pat, "samples_%s.bed"),
                "--stdout=%s" % os.path.join(pat, "stdout.txt"), "--log=%s" % os.path.join(cli_dir, "ref.log")]
        state.update(track=None, base=21, n_units=0, sampler=None, num_samples=5)
        mod.main(argv)
        for fn in os.listdir(pat):
            if fn.startswith("table_") or fn == "stdout.txt":
                lines = [l for l in open(os.path.join(pat, fn)) if not l.startswith("#")]
                with open(os.path.join(pat, fn), "w") as f:
                    f.writelines(lines)
        print("G5 patterns: %s" % sorted(os.listdir(pat)))
        # results-table round trip: --input-results-file re-computes the fdr of a previous table; --descriptions
        # appends columns (scripts/gat-run.py:287-291, gat/IO.py:296-328)
        aux = os.path.join(cli_dir, "aux")
        os.makedirs(aux, exist_ok=True)
        with open(os.path.join(aux, "descriptions.tsv"), "w") as f:
            f.write("annotation\tdescription\tgroup\nt0\tfirst track\tA\nt1\tsecond track\tB\n")
        for name, extra in (("results_file_descriptions", ["--input-results-file=%s" % os.path.join(cli_dir, "expected_default.tsv"),
                                                           "--qvalue-method=bonferroni", "--order=annotation",
                                                           "--descriptions=%s" % os.path.join(aux, "descriptions.tsv")]),
                            ("results_file_storey", ["--input-results-file=%s" % os.path.join(cli_dir, "expected_segment_tracks.tsv"),
                                                     "--qvalue-method=storey", "--order=pvalue"])):
            out = os.path.join(aux, "expected_%s.tsv" % name)
            mod.main(["gat-run.py", "--stdout=%s" % out, "--log=%s" % os.path.join(cli_dir, "ref.log")] + extra)
            lines = [l for l in open(out) if not l.startswith("#")]
            with open(out, "w") as f:
                f.writelines(lines)
    finally:
        gat.computeSample = original
    if os.path.exists(os.path.join(cli_dir, "ref.log")):
        os.remove(os.path.join(cli_dir, "ref.log"))
    with open(os.path.join(cli_dir, "cases.json"), "w") as f:
        json.dump(cases, f, indent=1)


# ------------------------------------------------------------------------------------------
# G8 q-values (gat/Engine.pyx:2025-2040 getQValues -> gat/Stats.py:26-160 computeQValues (Storey) and :192-258
# adjustPValues) and the "norm" p-value (gat/Engine.pyx:1979-1990)
def g8_qvalues():
    import warnings
    rs = numpy.random.RandomState(808)
    cases = []
    # vlambda=None is what the command line passes (gat/IO.py:474-477); without the keyword getQValues hands
    # computeQValues an array, whose `vlambda == None` test (gat/Stats.py:46) is elementwise under numpy >= 1.13 and
    # makes the call fail into the all-1.0 fallback (gat/Engine.pyx:2033-2035)
    methods = [("storey", dict(vlambda=None)), ("storey", dict(vlambda=0.5)), ("storey", dict(vlambda=0.0)),
               ("storey", dict(vlambda=None, pi0_method="bootstrap")), ("BH", {}), ("bonferroni", {}), ("holm", {}),
               ("hochberg", {}), ("BY", {}), ("none", {}), ("hommel", {})]
    for it in range(40):
        m = int(rs.choice([1, 2, 3, 5, 17, 60, 250]))
        kind = it % 4
        if kind == 0:
            pv = rs.uniform(0, 1, m)
        elif kind == 1:                                   # empirical p-values: multiples of 1/S with ties
            pv = rs.randint(1, 101, m) / 100.0
        elif kind == 2:                                   # enriched for small values
            pv = numpy.minimum(1.0, rs.beta(0.3, 2.0, m))
        else:
            pv = numpy.round(rs.uniform(0, 1, m), 1)
        for name, kw in methods:
            numpy.random.seed(1000 + it)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                try:
                    q = Engine.getQValues([float(x) for x in pv], method=name, **kw)
                    cases.append(dict(pvalues=[float(x) for x in pv], method=name, kwargs=kw, seed=1000 + it,
                                      expect=[float(x) for x in q]))
                except Exception as e:                     # noqa: BLE001
                    cases.append(dict(pvalues=[float(x) for x in pv], method=name, kwargs=kw, seed=1000 + it,
                                      error=type(e).__name__))
    normed = []
    for it in range(30):
        samples = rs.poisson(float(rs.choice([0.0, 3, 500])), size=int(rs.choice([5, 100]))).astype(float)
        obs = float(rs.choice(samples)) + float(rs.choice([0, 1, 40]))
        r = Engine.AnnotatorResult("t", "a", "c", obs, samples, reference=None, pseudo_count=1.0)
        normed.append(dict(observed=obs, samples=[float(x) for x in samples], expect=float(Engine.getNormedPValue(obs, r))))
    with open(os.path.join(HERE, "qvalues.json"), "w") as f:
        json.dump(dict(qvalues=cases, normed=normed), f, separators=(",", ":"))
    print("G8 q-values: %d cases (%d errors), %d normed p-values" % (len(cases), sum(1 for c in cases if "error" in c), len(normed)))


# ------------------------------------------------------------------------------------------
# G7 workspace generators (gat/Engine.pyx:2061-2153) and the segment-list operations under them
# (extend_segments / expand_segments, gat/SegmentList.pyx:1551-1591)
def g7_workspaces():
    rng = numpy.random.RandomState(4242)

    def rand_dict(keys, n, span, maxlen):
        d = Engine.IntervalDictionary()
        for k in keys:
            if rng.randint(0, 6) == 0:
                continue
            m = int(rng.randint(0, n))
            starts = rng.randint(0, span, m)
            lens = rng.randint(1, maxlen, m)
            d.add(k, sl([(int(a), int(a + b)) for a, b in zip(starts, lens)], normalize=True))
        return d

    def dump(d):
        return None if d is None else [[k, v.asList()] for k, v in d.items()]

    cases = []
    keys = ["chr1", "chr2", "chr3", "chrX"]
    gens = [("cooccurance", {}), ("annotation-centered", dict(extension=150)), ("annotation-centered", dict(expansion=2.5)),
            ("segment-centered", dict(extension=40)), ("segment-centered", dict(expansion=0.5)),
            ("segment-centered", dict(expansion=7.0))]
    for it in range(15):
        segs = rand_dict(keys, 40, 100000, 400)
        annos = rand_dict(keys, 30, 100000, 3000)
        ws = rand_dict(keys, 8, 100000, 40000)
        for name, kw in gens:
            if name == "cooccurance":
                g = Engine.ConditionalWorkspaceCooccurance()
            elif name == "annotation-centered":
                g = Engine.ConditionalWorkspaceAnnotationCentered(**kw)
            else:
                g = Engine.ConditionalWorkspaceSegmentCentered(**kw)
            with_annos = it % 3 != 0
            a, b, c = g(segs, annos if (with_annos or name != "segment-centered") else None, ws)
            cases.append(dict(generator=name, kwargs=kw, segments=dump(segs),
                              annotations=dump(annos) if (with_annos or name != "segment-centered") else None,
                              workspace=dump(ws), expect=[dump(a), dump(b), dump(c)]))
    ops = []
    for it in range(40):
        m = int(rng.randint(0, 30))
        starts = rng.randint(0, 5000, m)
        lens = rng.randint(1, 600, m)
        pairs_ = [(int(a), int(a + b)) for a, b in zip(starts, lens)]
        s1 = sl(pairs_)
        ext = int(rng.randint(0, 3000))
        s1.extend_segments(ext)
        s2 = sl(pairs_)
        exp = float(rng.choice([0.25, 0.5, 0.99, 1.0, 1.5, 2.0, 3.3, 50.0]))
        s2.expand_segments(exp)
        ops.append(dict(a=pairs_, extension=ext, extended=s1.asList(), expansion=exp, expanded=s2.asList()))
    with open(os.path.join(HERE, "workspaces.json"), "w") as f:
        json.dump(dict(generators=cases, ops=ops), f, separators=(",", ":"))
    print("G7 workspace generators: %d cases, %d segment ops" % (len(cases), len(ops)))


def g5u_cli_unpatched():
    """tables of the reference's gat-run.py AS IT IS -- numpy.random.seed(--random-seed) once, one stream for the whole run
    (scripts/gat-run.py:267-271) -- on the G5 inputs: what gat-run.py --reference-stream of this repository has to print"""
    import importlib.util
    cli_dir = os.path.join(HERE, "cli")
    spec = importlib.util.spec_from_file_location("gat_run_ref_u", os.path.join(os.path.dirname(gat.__file__), "..", "scripts", "gat-run.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cases = collections.OrderedDict([
        ("reference_stream_default", ["--num-samples=40", "--random-seed=31"]),
        ("reference_stream_tracks_isochores", ["--num-samples=25", "--random-seed=32", "--with-segment-tracks",
                                               "--isochores=%s" % os.path.join(cli_dir, "isochores.bed"), "--order=track",
                                               "--counter=segment-overlap"]),
        ("reference_stream_density", ["--num-samples=30", "--random-seed=33", "--counter=nucleotide-density",
                                      "--truncate-segments-to-workspace", "--order=annotation"]),
    ])
    for name, extra in cases.items():
        out = os.path.join(cli_dir, "expected_%s.tsv" % name)
        argv = ["gat-run.py", "--segments=%s" % os.path.join(cli_dir, "segments.bed"),
                "--annotations=%s" % os.path.join(cli_dir, "annotations.bed"),
                "--workspace=%s" % os.path.join(cli_dir, "workspace.bed"),
                "--stdout=%s" % out, "--log=%s" % os.path.join(cli_dir, "ref.log")] + extra
        mod.main(argv)
        lines = [l for l in open(out) if not l.startswith("#")]
        with open(out, "w") as f:
            f.writelines(lines)
        print("G5u cli %s: %d rows" % (name, len(lines) - 1))
    with open(os.path.join(cli_dir, "cases_reference_stream.json"), "w") as f:
        json.dump(dict((k, [x.replace(cli_dir + os.sep, "") for x in v]) for k, v in cases.items()), f, indent=1)


# ------------------------------------------------------------------------------------------
# G9: the two shapes north_star singles out -- config 3 (100 tracks x isochores) and config 5 (density against a
# 1M-interval annotation) -- at their FULL interval counts through the reference's own computeSample (mode 1) and gat.run
# (mode 0), a few samples each (the reference needs tens of milliseconds per sample and track).  The inputs are not stored
# (synthetic.config(name) regenerates them, pure numpy): the file carries the hash of every array of the flattened problem
# as the REFERENCE's structures gave it, so a test first proves it works on the same problem, then compares.
INPUT_KEYS = ("segs", "seg_off", "ws", "ws_off", "unit_contig", "annos", "anno_off", "cws_nseg")


def input_hashes(flat):
    out = {}
    for k in INPUT_KEYS:
        a = numpy.ascontiguousarray(flat[k])
        if k in ("seg_off", "ws_off", "anno_off", "cws_nseg"):
            a = a.astype(numpy.int64)
        elif k == "unit_contig":
            a = a.astype(numpy.int32)
        out[k] = hashlib.sha256(a.tobytes()).hexdigest()
    return out


def config_case(name, num_samples, seed, counters=None, tag=None):
    """counters: the counters run side by side (default: the configuration's own and the other nucleotide counter);
    tag: the golden's name where it is not the configuration's (run_<tag>_s<n>.npz)"""
    import time
    cfg = synthetic.config(name)
    if counters is None:
        counters = [cfg["counter"], "nucleotide-density" if cfg["counter"] == "nucleotide-overlap" else "nucleotide-overlap"]
    segments, annotations, workspace = build_reference_inputs(cfg)
    counter_objs = [COUNTERS[c]() for c in counters]
    flat = flat_problem(segments["merged"], workspace, annotations, 1, 100000)
    tracks = list(annotations.tracks)
    # mode 0: the reference's gat.run on one global stream
    numpy.random.seed(seed)
    t0 = time.time()
    results = gat.run(segments, annotations, workspace, Engine.SamplerAnnotator(bucket_size=1, nbuckets=100000), counter_objs,
                      Engine.UnconditionalWorkspace(), num_samples=num_samples, pseudo_count=1.0)
    t_run = time.time() - t0
    counts0 = numpy.zeros((len(counters), len(tracks), num_samples), dtype=numpy.float64)
    observed = numpy.zeros((len(counters), len(tracks)), dtype=numpy.float64)
    for r in results:
        k, a = counters.index(r.counter), tracks.index(r.annotation)
        counts0[k, a] = r.samples
        observed[k, a] = r.observed
    # mode 1: computeSample with per-unit re-seeding; sampled lists per (sample, contig) hashed one by one and together
    segs = segments["merged"]
    contig_annotations = annotations.clone()
    contig_annotations.fromIsochores()
    contig_workspace = workspace.clone()
    contig_workspace.fromIsochores()
    rsampler = ReseedingSampler(Engine.SamplerAnnotator(bucket_size=1, nbuckets=100000), segs, seed)
    counts1 = numpy.zeros((len(counters), len(tracks), num_samples), dtype=numpy.float64)
    sha, lens, first_last = hashlib.sha256(), [], []
    for x in range(num_samples):
        rsampler.sample_id = x
        rsampler.records = []
        w = gat.WorkData("merged", x, rsampler, segs, annotations, contig_annotations, workspace, contig_workspace, counter_objs)
        res = gat.computeSample((w, None, None, None))
        for k in range(len(counters)):
            for a, t in enumerate(tracks):
                counts1[k, a, x] = res[k][t]
        d = Engine.IntervalDictionary()
        for key, r in rsampler.records:
            d.add(key, r.clone())
        d.fromIsochores()
        for c in flat["contig_names"]:
            a = arr(d[str(c)]) if str(c) in d else numpy.empty(0, dtype=SEG)
            sha.update(a.tobytes())
            lens.append(len(a))
            first_last.append([int(a[0]["start"]), int(a[-1]["end"])] if len(a) else [0, 0])
    h = input_hashes(flat)
    numpy.savez_compressed(os.path.join(HERE, "run_%s_s%d.npz" % (tag or name, num_samples)),
                           config=name, seed=seed, num_samples=num_samples, counters=numpy.array(counters),
                           counts_mode0=counts0, counts_mode1=counts1, observed=observed,
                           samples_sha256_mode1=sha.hexdigest(), sample_list_lengths=numpy.array(lens, dtype=numpy.int64),
                           sample_list_first_last=numpy.array(first_last, dtype=numpy.int64),
                           input_keys=numpy.array(list(h.keys())), input_sha256=numpy.array(list(h.values())),
                           n_units=flat["n_units"], n_contigs=flat["n_contigs"], n_tracks=flat["n_tracks"],
                           unit_names=flat["unit_names"], contig_names=flat["contig_names"], track_names=flat["track_names"],
                           reference_seconds_per_sample_gat_run=t_run / num_samples)
    print("G9 run_%s_s%d (%s): units=%d contigs=%d tracks=%d intervals=%d; the reference's gat.run: %.3f s per sample; mean counts %s"
          % (tag or name, num_samples, ", ".join(counters), flat["n_units"], flat["n_contigs"], flat["n_tracks"], len(flat["annos"]), t_run / num_samples,
             numpy.round(counts1.mean(axis=2).ravel()[:3], 3)))


def g9_configs():
    config_case("config3", 4, 303)
    config_case("config5", 2, 505)
    config_case("config4", 2, 404)          # 100 k segments x 1 000 tracks: 2.3 s per sample of the reference's gat.run
    # every counter of the reference side by side on config 3's inputs (100 tracks x 192 isochore units): the segment- and
    # annotation-side counters at full interval counts, not only the two nucleotide counters of the metric
    config_case("config3", 3, 313, counters=list(COUNTERS), tag="config3all")
    # ... and on the headline shape (config 2: 10 k segments x 1 track x 10 k intervals, 24 contigs, no isochores)
    config_case("config2", 4, 202, counters=list(COUNTERS), tag="config2all")


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g5u", "g6", "g7", "g8", "g9", "g5s"]
    if "g9" in which:
        g9_configs()
    if "g5s" in which:
        g5s_sample_file()
    if "g5u" in which:
        g5u_cli_unpatched()
    if "g1" in which:
        g1_algebra()
    if "g2" in which:
        g2_rng()
    if "g3" in which:
        g3_sampler()
    if "g4" in which:
        g4_runs()
    if "g5" in which:
        g5_cli()
    if "g6" in which:
        g6_stats()
    if "g7" in which:
        g7_workspaces()
    if "g8" in which:
        g8_qvalues()

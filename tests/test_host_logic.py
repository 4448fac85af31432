"""Host-side logic (numpy interval algebra, problem flattening, statistics) against the
reference-generated goldens and the oracle.  CPU only."""
import collections
import json
import os

import numpy as np
import pytest

from gat_amd import engine, intervals as iv, problem, synthetic
from oracle import oracle as O

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _pairs(a):
    return [tuple(x) for x in a]


def _lst(a):
    return [(int(s), int(e)) for s, e in zip(a["start"], a["end"])]


@pytest.fixture(scope="module")
def algebra():
    with open(os.path.join(G, "algebra.json")) as f:
        return json.load(f)


def test_intervals_against_reference_goldens(algebra):
    n = 0
    for c in algebra:
        if c["op"] == "normalize":
            assert _lst(iv.normalize(iv.as_segments(c["a"]))) == _pairs(c["expect"])
        elif c["op"] == "merge":
            assert _lst(iv.merge(iv.as_segments(c["a"]), c["distance"])) == _pairs(c["expect"]), c
        elif c["op"] == "pair":
            a, b = iv.as_segments(c["a"]), iv.as_segments(c["b"])
            assert _lst(iv.filter(a, b)) == _pairs(c["filter"])
            assert _lst(iv.intersect(a, b)) == _pairs(c["intersect"])
            assert iv.total(a) == c["sum_a"]
            assert iv.overlap(a, b) == c["overlap"]
        elif c["op"] == "length_distribution":
            s = engine.SegmentList(iter=c["a"], normalize=True)
            if "error" in c:
                with pytest.raises(ValueError):
                    s.getLengthDistribution(c["bucket_size"], c["nbuckets"])
            else:
                h, b = s.getLengthDistribution(c["bucket_size"], c["nbuckets"])
                assert b == c["bucket_size_out"]
                nz = np.flatnonzero(h)
                assert [int(i) for i in nz] == c["nonzero"] and [int(h[i]) for i in nz] == c["counts"]
        else:
            continue
        n += 1
    assert n > 800


def test_segmentlist_api():
    s = engine.SegmentList()
    assert len(s) == 0 and s.isNormalized
    s.add(0, 100)
    assert len(s) == 1
    s.clear()
    assert len(s) == 0
    with pytest.raises(OverflowError):          # test/test_SegmentList.py:353-355
        engine.SegmentList(iter=[(-100, 5)])
    a = engine.SegmentList(iter=[(x, x + 10) for x in range(0, 1000, 100)], normalize=True)
    b = engine.SegmentList(iter=[(0, 1000)], normalize=True)
    b.intersect(a)
    assert b.asList() == a.asList()
    c = engine.SegmentList(iter=[(x, x + 5) for x in range(500, 2000, 100)], normalize=True)
    c.filter(a)
    assert c.asList() == [(500, 505), (600, 605), (700, 705), (800, 805), (900, 905)]
    s1 = engine.SegmentList(iter=[(x, x + 100) for x in range(0, 1000, 100)])
    s2 = engine.SegmentList(iter=[(x, x + 100) for x in range(2000, 3000, 100)])
    s1.extend(s2)
    assert s1.sum() == 2 * s2.sum() and len(s1) == 2 * len(s2)


def test_isochore_roundtrip():
    """test/test_gat.py:87-114: toIsochores / fromIsochores round trip."""
    d = engine.IntervalDictionary()
    for contig in ("contig1", "contig2"):
        d.add(contig, engine.SegmentList(iter=[(x, x + 10) for x in range(0, 1000, 100)], normalize=True))
    iso = engine.IntervalCollection()
    iso.add("highGC", "contig1", engine.SegmentList(iter=[(0, 500)], normalize=True))
    iso.add("lowGC", "contig1", engine.SegmentList(iter=[(500, 1000)], normalize=True))
    iso.add("highGC", "contig2", engine.SegmentList(iter=[(0, 250)], normalize=True))
    iso.add("lowGC", "contig2", engine.SegmentList(iter=[(250, 1000)], normalize=True))
    orig = d.clone()
    d.toIsochores(iso)
    assert sorted(d.keys()) == sorted(["contig2.highGC", "contig1.highGC", "contig2.lowGC", "contig1.lowGC"])
    d.fromIsochores()
    assert sorted(d.keys()) == ["contig1", "contig2"]
    for k in d.keys():
        assert d[k].asList() == orig[k].asList()


@pytest.mark.parametrize("name,iso,trunc", [("small_isochores", True, False), ("small_contigs", False, False),
                                            ("small_isochores_truncated", True, True)])
def test_flatten_matches_reference_walk(name, iso, trunc):
    z = np.load(os.path.join(G, "run_%s.npz" % name))
    _, cfg = synthetic.small_genome()
    f = problem.flatten_arrays(cfg["segments"], cfg["annotations"], cfg["workspace"], cfg["isochores"] if iso else None,
                               truncate_segments=trunc)
    for k in ("segs", "seg_off", "ws", "ws_off", "unit_contig", "annos", "anno_off", "cws_nseg"):
        assert np.array_equal(f[k], z[k]), k
    assert f["merge_contigs"] == int(z["merge_contigs"]) and list(f["unit_names"]) == [str(x) for x in z["unit_names"]]
    assert list(f["contig_names"]) == [str(x) for x in z["contig_names"]]


def test_flatten_via_collections_matches_arrays():
    """the class-based route (IntervalCollection.toIsochores ...) == the array route."""
    _, cfg = synthetic.small_genome()
    want = problem.flatten_arrays(cfg["segments"], cfg["annotations"], cfg["workspace"], cfg["isochores"])

    def coll(tracks):
        c = engine.IntervalCollection()
        for t, per in tracks:
            for contig, a in per.items():
                s = engine.SegmentList(array=a)
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
    f = problem.flatten_units(segments["merged"].asArrays(), workspaces["collapsed"].asArrays(),
                              [(t, annotations[t].asArrays()) for t in annotations.tracks])
    for k in ("segs", "seg_off", "ws", "ws_off", "unit_contig", "annos", "anno_off", "cws_nseg"):
        assert np.array_equal(f[k], want[k]), k


def test_enrichment_statistics_against_reference():
    with open(os.path.join(G, "stats.json")) as f:
        cases = json.load(f)
    class Ref(object):
        def __init__(self, fold):
            self.fold = fold
    assert sum(1 for c in cases if "reference_fold" in c) >= 20
    for c in cases:
        ref = Ref(c["reference_fold"]) if "reference_fold" in c else None
        r = engine.AnnotatorResult("track", "annotation", "counter", c["observed"], c["samples"], reference=ref,
                                   pseudo_count=c["pseudo_count"])
        assert r.expected == c["expected"] and r.stddev == c["stddev"] and r.fold == c["fold"]
        assert r.pvalue == c["pvalue"]
        assert str(r).split("\t")[2:] == c["row"]
    # the reference's own known answers (test/test_gat.py:120-129 is float data, :272-284 ties)
    samples = [0] * 66 + [1] * 2 + [2] * 20 + [3] * 1 + [4] * 6 + [6] * 2 + [8] * 2 + [16] * 1
    assert engine.AnnotatorResult("t", "a", "c", 16, samples).pvalue == 0.01


def test_pvalue_matches_oracle_restatement():
    rs = np.random.RandomState(4)
    for _ in range(200):
        s = np.sort(rs.poisson(rs.choice([1, 20, 400]), size=int(rs.choice([1, 7, 100]))).astype(np.float64))
        v = float(rs.choice(s)) if rs.rand() < 0.7 else float(rs.randint(0, 500))
        e = float(np.mean(s))
        assert engine.getTwoSidedPValue(s, e, v) == O.two_sided_pvalue(s, e, v)


def test_bed_reader_and_input_pipeline():
    """readFromBed / buildSegments / applyIsochores on the CLI fixture files (track lines, name column,
    ignore_tracks) -- the structures the reference builds before gat.run."""
    import gat_amd
    from gat_amd import io as IO
    cli = os.path.join(G, "cli")
    d = IO.readFromBed(os.path.join(cli, "segments.bed"))
    assert sorted(d.keys()) == ["segA", "segB"]
    _, cfg = synthetic.small_genome()
    for contig, a in cfg["segments"].items():
        assert d["segA"][contig].asList() == [(int(s), int(e)) for s, e in zip(a["start"], a["end"])]
    d = IO.readFromBed(os.path.join(cli, "segments.bed"), ignore_tracks=True)
    assert list(d.keys()) == ["merged"]
    d = IO.readFromBed(os.path.join(cli, "workspace.bed"))
    assert list(d.keys()) == ["workspace.bed"]                      # no track line, 3 columns: file name
    opts, _ = gat_amd.buildParser().parse_args(["--segments=%s" % os.path.join(cli, "segments.bed"),
                                               "--annotations=%s" % os.path.join(cli, "annotations.bed"),
                                               "--workspace=%s" % os.path.join(cli, "workspace.bed"),
                                               "--isochores=%s" % os.path.join(cli, "isochores.bed")])
    segments, annotations, workspaces, isochores = IO.buildSegments(opts)
    assert list(segments.tracks) == ["merged"] and sorted(annotations.tracks) == ["t0", "t1", "t2"]
    assert sorted(isochores.tracks) == ["iso0", "iso1", "iso2"]
    ws = IO.applyIsochores(segments, annotations, workspaces, opts, isochores)
    assert all("." in k for k in ws.keys()) and all("." in k for k in segments["merged"].keys())
    flat = problem.flatten_units(segments["merged"].asArrays(), ws.asArrays(),
                                 [(t, annotations[t].asArrays()) for t in annotations.tracks])
    assert flat["merge_contigs"] == 1 and flat["n_units"] == 12 and flat["n_tracks"] == 3


def test_qvalues_bh():
    """Stats.adjustPValues BH (gat/Stats.py:236-240) == textbook Benjamini-Hochberg."""
    from gat_amd import stats
    p = np.array([0.01, 0.04, 0.03, 0.005, 0.5, 0.22])
    q = stats.adjustPValues(p, "BH")
    o = np.argsort(p)
    want = np.minimum.accumulate((p[o] * len(p) / np.arange(1, len(p) + 1))[::-1])[::-1]
    assert np.allclose(q[o], np.minimum(want, 1.0), rtol=0, atol=1e-15)
    assert np.array_equal(stats.adjustPValues([0.2], "BH"), [0.2])


def _dict_from(dump):
    import gat_amd as gat
    if dump is None:
        return None
    d = gat.IntervalDictionary()
    for k, pairs in dump:
        d.add(k, gat.SegmentList(iter=[tuple(p) for p in pairs], normalize=True))
    return d


def _dump(d):
    return None if d is None else [[k, [list(p) for p in v.asList()]] for k, v in d.items()]


def test_workspace_generators_match_reference():
    """ConditionalWorkspace* (gat/Engine.pyx:2093-2153) against what the reference returned
    (tests/golden/workspaces.json, make_goldens.py g7)."""
    import json
    import os
    import gat_amd as gat
    with open(os.path.join(os.path.dirname(__file__), "golden", "workspaces.json")) as f:
        g = json.load(f)
    for case in g["generators"]:
        kw = case["kwargs"]
        if case["generator"] == "cooccurance":
            gen = gat.ConditionalWorkspaceCooccurance()
        elif case["generator"] == "annotation-centered":
            gen = gat.ConditionalWorkspaceAnnotationCentered(**kw)
        else:
            gen = gat.ConditionalWorkspaceSegmentCentered(**kw)
        segs, annos, ws = _dict_from(case["segments"]), _dict_from(case["annotations"]), _dict_from(case["workspace"])
        before = (_dump(segs), _dump(annos), _dump(ws))
        a, b, c = gen(segs, annos, ws)
        assert [_dump(a), _dump(b), _dump(c)] == case["expect"], (case["generator"], kw)
        assert (_dump(segs), _dump(annos), _dump(ws)) == before          # inputs are never mutated
    for op in g["ops"]:
        s = gat.SegmentList(iter=[tuple(p) for p in op["a"]])
        s.extend_segments(op["extension"])
        assert [list(p) for p in s.asList()] == op["extended"]
        s = gat.SegmentList(iter=[tuple(p) for p in op["a"]])
        s.expand_segments(op["expansion"])
        assert [list(p) for p in s.asList()] == op["expanded"]


def test_centered_workspace_needs_a_size():
    import gat_amd as gat
    with pytest.raises(ValueError):
        gat.ConditionalWorkspaceSegmentCentered()
    with pytest.raises(ValueError):
        gat.SegmentList(iter=[(1, 5)]).expand_segments(0.0)


def test_qvalues_match_reference():
    """getQValues / computeQValues / adjustPValues and the 'norm' p-value against the reference's output
    (tests/golden/qvalues.json, make_goldens.py g8): Storey with smoother and bootstrap pi0, p.adjust family."""
    import json
    import os
    import warnings
    import gat_amd as gat
    with open(os.path.join(os.path.dirname(__file__), "golden", "qvalues.json")) as f:
        g = json.load(f)
    n_storey = 0
    for case in g["qvalues"]:
        np.random.seed(case["seed"])
        if "error" in case:
            with pytest.raises(NotImplementedError):
                gat.getQValues(case["pvalues"], method=case["method"], **case["kwargs"])
            continue
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            q = gat.getQValues(case["pvalues"], method=case["method"], **case["kwargs"])
        assert [float(x) for x in q] == case["expect"], (case["method"], case["kwargs"], len(case["pvalues"]))
        n_storey += case["method"] == "storey"
    assert n_storey >= 100
    for case in g["normed"]:
        r = gat.AnnotatorResult("t", "a", "c", case["observed"], case["samples"], reference=None, pseudo_count=1.0)
        assert float(gat.getNormedPValue(case["observed"], r)) == case["expect"]


def test_cli_results_file_round_trip(tmp_path):
    """--input-results-file (fdr re-computed on a previous table) and --descriptions against the reference's
    output for the same files (tests/golden/cli/aux/, make_goldens.py g5).  No sampling, so no GPU."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gat_run_amd", os.path.join(root, "scripts", "gat-run.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cli = os.path.join(root, "tests", "golden", "cli")
    cases = {
        "results_file_descriptions": ["--input-results-file=%s" % os.path.join(cli, "expected_default.tsv"),
                                      "--qvalue-method=bonferroni", "--order=annotation",
                                      "--descriptions=%s" % os.path.join(cli, "aux", "descriptions.tsv")],
        "results_file_storey": ["--input-results-file=%s" % os.path.join(cli, "expected_segment_tracks.tsv"),
                                "--qvalue-method=storey", "--order=pvalue"],
    }
    for name, extra in cases.items():
        out = str(tmp_path / (name + ".tsv"))
        assert mod.main(["gat-run.py", "--stdout=%s" % out, "--log=%s" % str(tmp_path / "log")] + extra) == 0
        got = [l for l in open(out) if not l.startswith("#")]
        want = open(os.path.join(cli, "aux", "expected_%s.tsv" % name)).readlines()
        assert got == want, name


def test_cli_side_files_match_reference(tmp_path):
    """--output-stats / --output-bed (-P pattern): the collection summaries and bed dumps of the input pipeline
    (gat/IO.py:20-32, :150-279; gat/Engine.pyx:2959-2981, :3152-3165) against the files the reference wrote for the
    same inputs (tests/golden/cli/aux/stats/, from `gat-run.py --isochores=... --output-stats=all --output-bed=all`
    ; the overlap summary is counted on the device: tests/test_hip_parity.py).  Host only: no sampling."""
    import os
    import gat_amd as gat
    from gat_amd import IO
    cli = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cli")
    want_dir = os.path.join(cli, "aux", "stats")
    base = ["--segments=%s" % os.path.join(cli, "segments.bed"), "--annotations=%s" % os.path.join(cli, "annotations.bed"),
            "--workspace=%s" % os.path.join(cli, "workspace.bed")]
    sel = ["--output-stats=annotations", "--output-stats=segments", "--output-stats=workspaces", "--output-stats=isochores",
           "--output-bed=all", "-P", str(tmp_path / "%s")]
    opts, _ = gat.buildParser().parse_args(base + ["--isochores=%s" % os.path.join(cli, "isochores.bed")] + sel)
    segments, annotations, workspaces, isochores = IO.buildSegments(opts)
    IO.applyIsochores(segments, annotations, workspaces, opts, isochores)
    names = ["stats_annotations_isochores", "stats_isochores_raw", "stats_segments_isochores", "stats_workspaces_collapsed",
             "stats_workspaces_input", "stats_workspaces_isochores", "annotations_isochores.bed", "segments_isochores.bed",
             "workspaces_isochores.bed"]
    for n in names:
        assert open(str(tmp_path / n)).read() == open(os.path.join(want_dir, n)).read(), n
    with pytest.raises(OSError):                          # side files are not overwritten without --force
        segments, annotations, workspaces, isochores = IO.buildSegments(opts)


def test_from_counts_reads_the_counts_table():
    """gat.fromCounts (gat/__init__.py:1091-1117) on the counts table the reference wrote (tests/golden/cli/aux/patterns/):
    rows come back as AnnotatorResults whose statistics are those of the table the same run printed."""
    import os
    import gat_amd as gat
    pat = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cli", "aux", "patterns")
    results = gat.fromCounts(os.path.join(pat, "counts_nucleotide-overlap.tsv"))
    table = [l.rstrip("\n").split("\t") for l in open(os.path.join(pat, "table_nucleotide-overlap.tsv"))][1:]
    by_key = dict(((r[0], r[1]), r) for r in table)
    assert len(results) == len(table) > 0
    for r in results:
        row = by_key[(r.track, r.annotation)]
        assert str(r).split("\t")[2:10] == row[2:10]        # observed .. pvalue (the q-value is set by the output stage)
    with pytest.raises(ValueError):
        gat.fromCounts(os.path.join(pat, "table_nucleotide-overlap.tsv"))


def test_position_list_host_logic():
    """PositionList (gat/PositionList.pyx) as --annotations-to-points uses it: positions from segments by
    midpoint / start / end, normalize = sort + equal positions once, intersect keeps positions inside segments,
    sum() counts positions; what the reference's PositionList cannot do raises TypeError here too."""
    import gat_amd as gat
    s = gat.SegmentList(iter=[(10, 20), (15, 31), (40, 40), (100, 101)])
    want = {"midpoint": [15, 23, 100], "start": [10, 15, 100], "end": [20, 31, 101]}
    for method, positions in want.items():
        p = gat.PositionList()
        p.fromSegmentList(s, method=method)
        assert p.asList() == positions and len(p) == 3 and p.sum() == 3
    with pytest.raises(ValueError):
        gat.PositionList().fromSegmentList(s, method="centre")
    p = gat.PositionList(iter=[7, 3, 7, 50, 3, 99], normalize=True)
    assert p.asList() == [3, 7, 50, 99] and p.isNormalized and p.max() == 99 and p.min() == 3
    q = p.clone()
    q.intersect(gat.SegmentList(iter=[(0, 4), (7, 8), (50, 50), (60, 99)], normalize=True))
    assert q.asList() == [3, 7] and p.asList() == [3, 7, 50, 99]
    assert isinstance(q, gat.PositionList) and q.asArray()["end"].tolist() == [4, 8]      # one-base intervals on the device side
    with pytest.raises(TypeError):
        gat.SegmentList(iter=[(0, 10)], normalize=True).intersect(p)
    for name in ("filter", "merge", "extend"):
        with pytest.raises(TypeError):
            getattr(p, name)(p)
    c = gat.IntervalCollection("annotations")
    c.add("t0", "chr1", gat.SegmentList(iter=[(0, 10), (4, 14), (30, 40)]))
    c.toPositions("start")
    c.normalize()
    assert c.hasPositions() and c["t0"]["chr1"].asList() == [0, 4, 30] and c.sum() == 3 and c.counts() == 3
    with pytest.raises(TypeError):
        c.merge()
    d = c["t0"].clone()
    d.intervals["chr1.iso"] = d.intervals.pop("chr1")
    with pytest.raises(TypeError):
        d.fromIsochores()


def test_numpy_summation_model():
    """gat_null_stats restates numpy.mean / numpy.std on the device; what it restates is numpy's summation order for a
    contiguous float64 array: chunks of 8 192 elements (the ufunc buffer), each summed pairwise (halves split at a
    multiple of 8 down to blocks of at most 128, a block with 8 running sums and a serial remainder), chunk sums added
    left to right.  This pins that model against the numpy at hand for many lengths."""
    def pw(a):
        n = len(a)
        if n < 8:
            r = 0.0
            for x in a:
                r = r + x
            return r
        if n <= 128:
            r = [a[j] for j in range(8)]
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
        n2 = n // 2
        n2 -= n2 % 8
        return pw(a[:n2]) + pw(a[n2:])

    def np_sum(a):
        r = None
        for i in range(0, len(a), 8192):
            p = pw(a[i:i + 8192])
            r = p if r is None else r + p
        return r

    rs = np.random.RandomState(3)
    for n in list(range(1, 140)) + [255, 1000, 8191, 8192, 8193, 10000, 16385, 30011]:
        a = rs.random_sample(n) * 1e5 if n % 2 else rs.randint(0, 10 ** 7, n).astype(np.float64)
        al = [float(x) for x in a]
        m = np_sum(al) / n
        assert m == float(np.mean(a)), n
        v = np_sum([(x - m) * (x - m) for x in al]) / n
        assert float(np.sqrt(v)) == float(np.std(a)), n


def _coll(tracks):
    import gat_amd
    c = gat_amd.IntervalCollection()
    for t, per in tracks:
        for contig, a in per.items():
            s = gat_amd.SegmentList(array=a)
            s.isNormalized = 1
            c.add(t, contig, s)
    return c


def test_to_isochores_one_pass_equals_list_by_list():
    """IntervalDictionary.toIsochores (gat/Engine.pyx:2837-2855) as one vectorised pass over the dictionary against the
    list-by-list form of the reference: random tracks with missing contigs, empty lists, 1-4 isochore classes with and
    without gaps, truncate and filter; keys, lists, flags, sums, and the flat form it leaves behind."""
    import collections
    from gat_amd import engine, intervals as iv
    rs = np.random.RandomState(5)

    def rand_list(n, size):
        s = np.sort(rs.randint(0, size, n))
        return iv.normalize(iv.make(s, s + rs.randint(1, 50, n)))

    for trial in range(60):
        contigs = ["c%d" % i for i in range(rs.randint(1, 5))]
        K = rs.randint(1, 5)
        block = rs.randint(20, 200)
        iso = []
        for k in range(K):
            per = collections.OrderedDict()
            for c in contigs:
                if rs.rand() < 0.2:
                    continue
                st = np.arange(k * block, 3000, K * block)
                per[c] = iv.make(st, st + block - (rs.randint(0, 3) if rs.rand() < .5 else 0))
            iso.append(("iso%d" % k, per))
        tr = []
        for t in range(3):
            per = collections.OrderedDict()
            for c in contigs + ["zz"]:
                if rs.rand() < 0.3:
                    continue
                per[c] = rand_list(rs.randint(0, 40), 3200)
            tr.append(("t%d" % t, per))
        for truncate in (True, False):
            # A: the whole collection in one call of the library (gat_isochore_split; lists made on demand), C: one vectorised
            # numpy pass per dictionary, B: list by list as the reference does
            A, B, Cc = _coll(tr), _coll(tr), _coll(tr)
            isoA, isoB = _coll(iso), _coll(iso)
            A.toIsochores(isoA, truncate)
            assert all(isinstance(A[t].intervals, engine._LazyLists) for t in A.tracks) or not any(len(A[t]) for t in A.tracks)
            orig = engine.IntervalDictionary._to_isochores_flat
            orig_native = engine.IntervalCollection._to_isochores_native
            engine.IntervalCollection._to_isochores_native = lambda *a, **k: False
            try:
                Cc.toIsochores(_coll(iso), truncate)
                engine.IntervalDictionary._to_isochores_flat = lambda *a, **k: False
                B.toIsochores(isoB, truncate)
            finally:
                engine.IntervalDictionary._to_isochores_flat = orig
                engine.IntervalCollection._to_isochores_native = orig_native
            # (the keys a look-up added to the isochore tracks, in the reference's order)
            assert [list(isoA[t].keys()) for t in isoA.tracks] == [list(isoB[t].keys()) for t in isoB.tracks]
            for t in A.tracks:
                assert list(A[t].keys()) == list(B[t].keys()) == list(Cc[t].keys())
                assert A[t].sum() == B[t].sum() and A[t].counts() == B[t].counts()        # (from the flat form: no list is made)
                f = A[t]._flat()
                assert f is A[t]._flat()
                for k in A[t].keys():
                    assert np.array_equal(A[t][k].asArray(), B[t][k].asArray()), (trial, truncate, t, k)
                    assert np.array_equal(Cc[t][k].asArray(), B[t][k].asArray()), (trial, truncate, t, k)
                    assert A[t][k].isNormalized == B[t][k].isNormalized and type(A[t][k]) is type(B[t][k])
                assert A[t]._flat() is f                                                   # the lists made meanwhile are its views
                assert [np.array_equal(x.asArray(), y.asArray()) for x, y in zip(A[t].intervals.values(), B[t].intervals.values())] \
                    == [True] * len(B[t])
            cf = A._flat()
            assert cf is A._flat() and len(cf[0]) == sum(len(A[t]._flat().data) for t in A.tracks)
    # overlapping classes are not a partition: the shortcut steps aside and the result is still the reference's
    seg = [("t", collections.OrderedDict([("c0", iv.make([10, 100], [60, 180]))]))]
    iso2 = [("a", collections.OrderedDict([("c0", iv.make([0], [120]))])), ("b", collections.OrderedDict([("c0", iv.make([50], [200]))]))]
    A = _coll(seg)
    A.toIsochores(_coll(iso2), True)
    assert A["t"]["c0.a"].asList() == [(10, 60), (100, 120)] and A["t"]["c0.b"].asList() == [(50, 60), (100, 180)]


def test_lists_made_on_demand_behave_like_a_dictionary():
    """engine._LazyLists (what IntervalCollection.toIsochores leaves in a dictionary): reading, replacing, adding and deleting
    keys, iteration, clone / fromIsochores on top of it; the flat form is kept while nothing was replaced and rebuilt after."""
    import collections
    from gat_amd import engine, intervals as iv
    import gat_amd
    seg = [("t", collections.OrderedDict([("c0", iv.make([10, 100, 300], [60, 180, 320])), ("c1", iv.make([5], [500]))]))]
    iso = [("a", collections.OrderedDict([("c0", iv.make([0], [120])), ("c1", iv.make([0], [100]))])),
           ("b", collections.OrderedDict([("c0", iv.make([120], [400])), ("c1", iv.make([100], [600]))]))]
    A = _coll(seg)
    A.toIsochores(_coll(iso), True)
    d = A["t"]
    assert isinstance(d.intervals, engine._LazyLists) and list(d.keys()) == ["c0.a", "c0.b", "c1.a", "c1.b"]
    assert "c0.a" in d and "c9.a" not in d and len(d) == 4
    assert dict.__getitem__(d.intervals, "c0.b") is None                      # not made yet
    f = d._flat()
    assert d.sum() == 50 + 20 + 60 + 20 + 95 + 400 and d.counts() == 6
    assert d["c0.b"].asList() == [(120, 180), (300, 320)] and d["c0.b"] is d["c0.b"]
    assert d.intervals.get("c1.a").asList() == [(5, 100)] and d.intervals.get("nope") is None
    assert d._flat() is f
    import copy
    import pickle
    e = pickle.loads(pickle.dumps(d.intervals))                               # (travels as the plain dictionary it stands for)
    assert type(e) is collections.defaultdict and list(e.keys()) == list(d.keys()) and e["c1.b"].asList() == [(100, 500)]
    assert copy.copy(d.intervals)["c0.b"].asList() == [(120, 180), (300, 320)] and d._flat() is f
    # the copies CPython makes of a dict subclass without asking it -- dict(d), OrderedDict(d), {**d}, update(d) -- hold the lists
    B = _coll(seg)
    B.toIsochores(_coll(iso), True)
    lazy = B["t"].intervals
    assert dict.__getitem__(lazy, "c1.b") is None
    for cp in (dict(lazy), collections.OrderedDict(lazy), {**lazy}):
        assert all(v is not None for v in cp.values()) and cp["c1.b"].asList() == [(100, 500)] and list(cp) == list(d.keys())
    other = {}
    other.update(lazy)
    assert other["c0.b"].asList() == [(120, 180), (300, 320)]
    assert copy.deepcopy(lazy)["c1.a"].asList() == [(5, 100)]
    c = d.clone()                                                             # (goes through items(): everything is made)
    assert [k for k, _ in c.items()] == list(d.keys()) and c["c1.b"].asList() == [(100, 500)]
    assert d._flat() is f
    d["c0.a"].normalize()                                                     # (in place, the view stays: still the same form)
    d["c0.a"] = gat_amd.SegmentList(array=iv.make([1], [2]))                  # replaced: the flat form follows
    f2 = d._flat()
    assert f2 is not f and d.sum() == 1 + 20 + 60 + 95 + 400 and d._flat() is f2
    del d["c1.a"]
    assert list(d.keys()) == ["c0.a", "c0.b", "c1.b"] and d.counts() == 4
    d["c7.a"]                                                                 # a look-up adds an empty list (defaultdict)
    assert list(d.keys())[-1] == "c7.a" and len(d["c7.a"]) == 0
    d.fromIsochores()
    assert list(d.keys()) == ["c0", "c1", "c7"] and d["c0"].asList() == [(1, 2), (120, 180), (300, 320)]


def test_flat_form_follows_the_lists():
    """IntervalDictionary._flat: one array for all lists, found again while nothing changed, rebuilt when a list is replaced,
    added or removed; sum() / counts() through it equal the per-list ones."""
    import collections
    import gat_amd
    from gat_amd import intervals as iv
    per = collections.OrderedDict(("k%d" % i, iv.make(np.arange(i) * 10, np.arange(i) * 10 + 3)) for i in range(20))
    d = _coll([("t", per)])["t"]
    f = d._flat()
    assert f is d._flat() and f.keys == list(per.keys()) and len(f.data) == sum(range(20))
    assert d.sum() == 3 * sum(range(20)) and d.counts() == sum(range(20))
    for i, k in enumerate(per):
        assert np.array_equal(d[k].asArray(), per[k]) and np.array_equal(f.data[f.off[i]:f.off[i + 1]], per[k])
    d["k3"].merge(7)                                    # [0,3) [10,13) [20,23) become one: the list has a new array
    g = d._flat()
    assert g is not f and len(g.data) == sum(range(20)) - 2 and d.counts() == len(g.data)
    d["new"] = gat_amd.SegmentList(iter=[(1, 2)], normalize=True)
    assert d._flat().keys[-1] == "new" and d.counts() == len(g.data) + 1
    del d["k19"]
    assert "k19" not in d._flat().keys
    b, e = d._flat().ranges(["k5", "nope", "k2"], base=100)
    assert (e - b).tolist() == [5, 0, 2] and b[0] == 100 + d._flat().off[d._flat().position("k5")]


def test_flatten_dictionaries_equals_flatten_units():
    """problem.flatten_dictionaries (the host classes' flat forms + annotation lists handed over with a group id each)
    describes the same problem as problem.flatten_units on the arrays: identical units, contigs, workspaces, and --
    grouping the lists as the library does (gat_prep.hip: group_annotations) -- identical contig-level annotations."""
    from gat_amd import intervals as iv, problem, synthetic
    _, cfg = synthetic.small_genome()
    segments, annotations, workspace, _ = synthetic.as_collections(cfg)
    segs = segments["merged"]
    tracks = list(annotations.tracks)
    new = problem.flatten_dictionaries(segs, workspace, annotations, tracks, 1, 1000)
    old = problem.flatten_units(segs.asArrays(), workspace.asArrays(), [(t, annotations[t].asArrays()) for t in tracks], 1, 1000)
    for k in ("n_units", "n_contigs", "merge_contigs", "n_tracks", "unit_names", "contig_names", "bucket_size", "nbuckets"):
        assert old[k] == new[k], k
    for k in ("segs", "seg_off", "ws", "ws_off", "unit_contig", "cws_nseg"):
        assert np.array_equal(old[k], new[k]), k
    assert len(new["anno_group"]) == len(new["anno_off"]) == len(new["anno_end"])
    C = new["n_contigs"]
    for g in range(new["n_tracks"] * C):
        members = np.flatnonzero(new["anno_group"] == g)
        parts = [new["annos"][new["anno_off"][l]:new["anno_end"][l]] for l in members]
        got = iv.merge(np.concatenate(parts), 0) if parts else iv.EMPTY
        assert np.array_equal(got, old["annos"][old["anno_off"][g]:old["anno_off"][g + 1]]), g
    # lists of contigs without an active unit carry -1
    assert (new["anno_group"] == -1).sum() == sum(1 for t in tracks for k in annotations[t].keys()
                                                  if problem.split_key(k)[0] not in new["contig_names"])


def test_numpy_summation_self_check():
    """the run-time form of test_numpy_summation_model: what decides whether gat_null_stats may stand in for numpy"""
    import gat_amd
    assert gat_amd._numpy_summation_model_holds() is True


def test_intersection_sizes_equal_the_intersections():
    """gat_intersection_sizes (the overlap_* columns, gat/Engine.pyx:1911-1928) against intersect() list by list"""
    from gat_amd import _lib, intervals as iv
    rs = np.random.RandomState(23)

    def mk(n):
        s = np.sort(rs.randint(0, 5000, n))
        return iv.normalize(iv.make(s, s + rs.randint(1, 40, n)))
    n_groups, n_tracks = 5, 4
    a = [mk(rs.randint(0, 60)) for _ in range(n_groups)]
    b = [mk(rs.randint(0, 90)) for _ in range(n_tracks * n_groups)]
    a[2] = iv.EMPTY.copy()
    off = lambda ls: np.concatenate([[0], np.cumsum([len(x) for x in ls])]).astype(np.int64)  # noqa: E731
    bo = off(b)
    pairs, bases = _lib.intersection_sizes(np.concatenate(a), off(a), np.concatenate(b), bo[:-1], bo[1:], n_tracks)
    for t in range(n_tracks):
        inter = [iv.intersect(a[g], b[t * n_groups + g]) for g in range(n_groups)]
        assert pairs[t] == sum(len(x) for x in inter) and bases[t] == sum(iv.total(x) for x in inter)
    # touching segments do not intersect; identical ones do, once
    x = iv.make([0, 10, 30], [10, 20, 40])
    y = iv.make([10, 30], [15, 40])
    p, s = _lib.intersection_sizes(x, [0, 3], y, [0], [2], 1)
    assert (int(p[0]), int(s[0])) == (2, 15)


def test_plain_runs_do_not_import_torch():
    """torch is plumbing for the multi-GPU path only: importing the package, asking for the process-group state and for the
    default device must not import it (its import costs a plain gat-run.py seconds, minutes on a cold machine)"""
    import subprocess
    import sys
    code = ("import sys, gat_amd; from gat_amd import engine; "
            "assert gat_amd._dist_state() == (0, 1, None); assert engine.default_device() == 0; "
            "assert not any(m == 'torch' or m.startswith('torch.') for m in sys.modules), 'torch was imported'")
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_pinned_register_check_catches_a_stray_use(tmp_path):
    """tools/check_pinned_regs.py (run by the library's Makefile on the device assembly): inside a hand-pipelined loop only the
    loop's own loads into v96..v127 and its moves out of them may name those registers; a temporary there, alone or inside
    a register range, fails the build; outside the markers anything goes."""
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    tool = os.path.join(root, "tools", "check_pinned_regs.py")
    head = "_ZN3gat12k_place_pipeILi0ELi1ELi0EEEvNS_11SamplerArgsE:\n\tv_add_u32_e32 v97, s3, v82\n\t; GAT_PINNED_BEGIN\n"
    loads = "".join("\tglobal_load_dword v%d, v[2:3], off offset:%d\n" % (96 + q, 256 * q) for q in range(7))
    body = (loads + "\tglobal_load_dword v127, v[2:3], off offset:1792\n"
            "\ts_waitcnt vmcnt(24)\n\tv_mov_b32 v5, v96\n\tv_add_u32_e32 v6, v5, v7\n")
    tail = "\t; GAT_PINNED_END\n\tds_read_b32 v100, v3\n\ts_endpgm\n"

    def run(text):
        f = tmp_path / "k.s"
        f.write_text(text)
        return subprocess.run([sys.executable, tool, str(f)], capture_output=True, text=True)

    assert run(head + body + tail).returncode == 0
    for stray in ("\tv_add_u32_e32 v97, s3, v82\n", "\tds_read_b128 v[94:97], v3\n", "\tv_mov_b32 v100, v96\n"):
        r = run(head + body + stray + tail)
        assert r.returncode != 0 and "pinned-register loop" in (r.stderr + r.stdout), stray
    # the waits count the loop's loads: a spill (scratch / buffer access) or a chunk that is not eight loads fails too
    assert run(head + body + "\tscratch_load_dword v3, off, s32 offset:4\n" + tail).returncode != 0
    assert run(head + body.replace("\tglobal_load_dword v127, v[2:3], off offset:1792\n", "") + tail).returncode != 0
    assert run("\tv_mov_b32 v1, v2\n").returncode != 0            # no markers: the wrong file
    # the loop's own asm statements (;;#ASMSTART .. ;;#ASMEND) may READ a row register in place -- the written-out steps do --
    # but never write one or load into one; the same instruction from the compiler is a stray use
    def asm(t):
        return "\t;;#ASMSTART\n" + t + "\t;;#ASMEND\n"
    assert run(head + body + asm("\tv_and_b32 v40, s60, v96\n\tv_cmp_ge_u32 vcc, s54, v40\n") + tail).returncode == 0
    assert run(head + body + "\tv_and_b32 v40, s60, v96\n" + tail).returncode != 0
    assert run(head + body + asm("\tv_mov_b32 v96, v1\n") + tail).returncode != 0
    assert run(head + body + asm("\tds_read_b32 v97, v3\n") + tail).returncode != 0
    assert run(head + asm(loads + "\tglobal_load_dword v127, v[2:3], off offset:1792 nt\n") +
               asm("\ts_waitcnt vmcnt(24)\n\tv_mov_b32 v5, v96\n") + tail).returncode == 0


_BED_CASES = {
    "three_columns": "chr2\t10\t20\nchr1\t5\t9\nchr2\t30\t40\n",
    "names_and_order": "chr2\t10\t20\tB\nchr1\t5\t9\tA\nchr2\t30\t40\tA\nchr1\t50\t60\tB\nchr3\t1\t2\tA\nchr2\t41\t45\tB\n",
    "empty_names_beside_the_files_own": "chr1\t1\t2\t\nchr1\t3\t4\tx\nchr2\t5\t6\t\nchr1\t7\t8\tf.bed\n",
    "comments_and_blank_lines": "# header\nchr1\t1\t2\tA\n\n#chr1\t9\t9\tZ\nchr1\t3\t4\tA\n",
    "six_columns": "chr1\t1\t2\tA\t0\t+\nchr1\t3\t4\tB\t0\t-\n",
    "na_like_names": "chr1\t1\t2\tNA\nchr1\t3\t4\tnull\nchr1\t5\t6\tnan\nchr1\t7\t8\tN/A\n",
    "hash_in_a_name": "chr1\t1\t2\ta#b\nchr1\t3\t4\ta#b\n",
    "quotes": 'chr1\t1\t2\t"q\nchr1\t3\t4\tq"\n',
    "track_lines": 'track name="T1"\nchr1\t1\t2\ntrack name=T2\nchr1\t3\t4\tignored\n',
    "contig_called_track": "track7\t1\t2\n",
    "short_line": "chr1\t1\t2\tA\nchr1\t3\n",
    "short_name": "chr1\t1\t2\tA\nchr1\t3\t4\n",
    "more_fields_later": "chr1\t1\t2\nchr1\t3\t4\tA\n",
    "carriage_returns": "chr1\t1\t2\tA\r\nchr1\t3\t4\tA\r\n",
    "start_behind_end": "chr1\t1\t2\nchr1\t9\t4\n",
    "not_a_number": "chr1\t1\tx\n",
    "spaces_and_signs": "chr1\t 1\t+2\nchr1\t3\t4 \n",
    "negative": "chr1\t-5\t4\n",
    "float": "chr1\t1.0\t4\n",
    "no_final_newline": "chr1\t1\t2\tA\nchr1\t3\t4\tB",
    "empty": "",
    "only_comments": "# nothing\n",
}


def _read_bed_either_way(monkeypatch, paths, lines, **kw):
    from gat_amd import io as IO
    monkeypatch.setattr(IO, "_BED_TABLE_MIN_BYTES", 0)
    if lines:
        monkeypatch.setenv("GAT_BED_LINE_READER", "1")
    else:
        monkeypatch.delenv("GAT_BED_LINE_READER", raising=False)
    try:
        r = IO.readFromBed(paths, **kw)
    except Exception as e:                       # noqa: BLE001 -- the kind of the error is what is compared
        return type(e).__name__
    return [(t, [(c, r[t][c]._a.tolist()) for c in r[t].keys()]) for t in r.keys()]


@pytest.mark.parametrize("case", sorted(_BED_CASES))
@pytest.mark.parametrize("ignore_tracks", [False, True])
def test_bed_table_reader_equals_the_line_reader(tmp_path, monkeypatch, case, ignore_tracks):
    """readFromBed's table parser (files from 8 MB on) against the line-by-line reader it falls back to: the same tracks in
    the same order, the same contigs in the same order, the same intervals -- or the same kind of error"""
    f = tmp_path / "f.bed"
    f.write_text(_BED_CASES[case], newline="")
    want = _read_bed_either_way(monkeypatch, str(f), True, ignore_tracks=ignore_tracks)
    got = _read_bed_either_way(monkeypatch, str(f), False, ignore_tracks=ignore_tracks)
    assert got == want, case


def test_bed_table_reader_over_several_files(tmp_path, monkeypatch):
    """a track's lists grow over files in file order; a track in two files is an error unless split tracks are allowed"""
    import gzip
    a, b = tmp_path / "a.bed", tmp_path / "b.bed.gz"
    a.write_text("chr2\t1\t2\tT\nchr1\t3\t4\tU\n")
    with gzip.open(str(b), "wt") as f:
        f.write("chr1\t5\t6\tT\nchr2\t7\t8\tT\nchr9\t1\t3\n")
    for kw in (dict(allow_multiple=True), dict(allow_multiple=False), dict(ignore_tracks=True)):
        want = _read_bed_either_way(monkeypatch, [str(a), str(b)], True, **kw)
        got = _read_bed_either_way(monkeypatch, [str(a), str(b)], False, **kw)
        assert got == want, kw
    assert _read_bed_either_way(monkeypatch, [str(a), str(b)], False, allow_multiple=False) == "ValueError"


def test_list_sums_equal_segmentlist_sum():
    """gat_list_sums (the *_size columns of a run over 10^4 lists): SegmentList.sum() per list -- a uint32 accumulator,
    gat/SegmentList.pyx:1607 -- for lists given as ranges of one array"""
    from gat_amd import _lib, intervals as iv
    rs = np.random.RandomState(5)
    lists = []
    for _ in range(40):
        n = int(rs.randint(0, 50))
        s = np.sort(rs.randint(0, 1 << 20, n))
        lists.append(iv.normalize(iv.make(s, s + rs.randint(1, 3000, n))))
    lists.append(iv.make([0, 1 << 31], [(1 << 31) - 1, 0xffffffff]))          # sums beyond 2^32 wrap like the reference's accumulator
    data = np.concatenate(lists)
    end = np.cumsum([len(x) for x in lists])
    got = _lib.list_sums(data, end - [len(x) for x in lists], end)
    for g, x in zip(got.tolist(), lists):
        assert g == int((x["end"].astype(np.int64) - x["start"]).sum() & 0xFFFFFFFF)


def test_contig_list_lengths_equal_from_isochores():
    """problem.contig_list_lengths -- len(dictionary.fromIsochores()[contig]) for every contig in one vectorised pass (the
    density counter's divisor, gat/Engine.pyx:1437) -- against problem.from_isochores on the arrays: touching and overlapping
    pieces merge (merge(0)), empty segments vanish, keys without a dot pass through"""
    import gat_amd
    rs = np.random.RandomState(8)
    for case in range(30):
        d = gat_amd.IntervalDictionary()
        dotted = case % 3 != 0
        for c in range(int(rs.randint(1, 5))):
            for k in range(int(rs.randint(1, 5)) if dotted else 1):
                n = int(rs.randint(0, 30))
                s = np.sort(rs.randint(0, 2000, n))
                e = s + rs.randint(0, 120, n)                          # zero-length ones included
                sl = gat_amd.SegmentList(array=iv.make(s, e))
                d.add("chr%d.iso%d" % (c, k) if dotted else "chr%d" % c, sl)
        want = dict((c, len(a)) for c, a in problem.from_isochores(d.asArrays()).items())
        got = problem.contig_list_lengths(d)
        if dotted:
            assert dict(got) == want, case
        else:
            assert dict(got) == dict((c, len(a)) for c, a in d.asArrays().items()), case
        assert list(got.keys()) == list(want.keys())


def test_contig_list_lengths_with_dotted_and_plain_keys_of_one_contig():
    """a plain key replaces what the dotted keys of its contig have gathered (gat/Engine.pyx:2857-2876: new[isochore] =
    segmentlist); dotted keys behind it extend the plain list"""
    import gat_amd
    d = gat_amd.IntervalDictionary()
    d.add("chr1.a", gat_amd.SegmentList(array=iv.make([0, 10, 20], [5, 15, 25])))
    d.add("chr1", gat_amd.SegmentList(array=iv.make([100, 200], [150, 250])))
    d.add("chr1.c", gat_amd.SegmentList(array=iv.make([300, 400, 500], [350, 450, 550])))
    d.add("chr2.a", gat_amd.SegmentList(array=iv.make([1], [2])))
    want = collections.OrderedDict((c, len(a)) for c, a in problem.from_isochores(d.asArrays()).items())
    assert want == collections.OrderedDict([("chr1", 5), ("chr2", 1)])
    assert problem.contig_list_lengths(d) == want


def test_collection_ranges_equal_the_dictionaries_ranges():
    """IntervalCollection._ranges (where the lists of given keys lie in the collection's one array; kept for the length of a
    run()) against _DictFlat.ranges dictionary by dictionary: same keys, keys in another order, keys a dictionary lacks"""
    _, cfg = synthetic.small_genome()
    segments, annotations, workspace, _ = synthetic.as_collections(cfg)
    tracks = list(annotations.tracks)
    aflat = annotations._flat(tracks)
    keys = list(annotations[tracks[0]].keys())
    annotations._ranges_memo = {}
    try:
        for target in (keys, keys[::-1], keys[:3] + ["nowhere.iso9"] + keys[3:]):
            b, e = annotations._ranges(aflat, target)
            bb, ee = zip(*[f.ranges(target, base) for f, base in zip(aflat[2], aflat[1])])
            assert np.array_equal(b, np.concatenate(bb)) and np.array_equal(e, np.concatenate(ee))
            b2, e2 = annotations._ranges(aflat, target)                # (from the memo)
            assert b2 is b and e2 is e
    finally:
        annotations._ranges_memo = None
    # a dictionary whose keys differ from the others' takes the general path
    del annotations[tracks[1]][keys[0]]
    aflat = annotations._flat(tracks)
    b, e = annotations._ranges(aflat, keys)
    bb, ee = zip(*[f.ranges(keys, base) for f, base in zip(aflat[2], aflat[1])])
    assert np.array_equal(b, np.concatenate(bb)) and np.array_equal(e, np.concatenate(ee))


def test_result_rows_keep_the_device_rows_until_asked():
    """AnnotatorResult with the device's statistics keeps the row as it came (int64, or a row that fetches the matrix from the
    device on demand) and makes the float copy the reference holds when somebody asks: same samples, same p-values"""
    import gat_amd
    rs = np.random.RandomState(4)
    row = rs.randint(0, 500, 1000).astype(np.int64)
    observed = 260.0
    eager = gat_amd.AnnotatorResult("t", "a", "c", observed, row)
    srt = np.sort(row)
    off = int(0.05 * len(row))
    stats = (float(np.mean(row.astype(np.float64))), float(np.std(row.astype(np.float64))), float(srt[off]), float(srt[len(row) - off]),
             int((row < observed).sum()), int((row == observed).sum()))
    fetched = []

    class Row(object):                                   # stands in for gat_amd._DeviceRow
        def __len__(self):
            return len(row)

        def __array__(self, dtype=None, copy=None):
            fetched.append(1)
            return row.astype(dtype) if dtype is not None else row

    for samples in (row, Row()):
        lazy = gat_amd.AnnotatorResult("t", "a", "c", observed, samples, _stats=stats)
        assert str(lazy) == str(eager) and not fetched
        assert lazy.getEmpiricalPValue(observed) == eager.getEmpiricalPValue(observed) and not fetched   # (counted on the device)
        assert lazy.getEmpiricalPValue(100.0) == eager.getEmpiricalPValue(100.0)                           # (another value: the samples)
        assert np.array_equal(lazy.samples, eager.samples) and lazy.getSample(7) == eager.getSample(7)
    assert fetched


def test_device_counts_rows_fetch_once():
    """gat_amd._DeviceCounts (the gathered matrix under nccl): rank 0 reads it back at once, the other ranks' rows on demand,
    once for all rows"""
    torch = pytest.importorskip("torch")
    import gat_amd
    m = np.arange(2 * 3 * 5, dtype=np.int64).reshape(2, 3, 5)
    counts = gat_amd._DeviceCounts(torch.from_numpy(m.copy()), ["nucleotide-overlap", "segment-overlap"])
    rows = counts.rows(1, 3, True)
    assert counts.host is None and len(rows[2]) == 5
    assert np.array_equal(np.array(rows[2], dtype=np.float64), m[1, 2].astype(np.float64))
    assert counts.host is not None and counts.tensor is None
    assert np.array_equal(counts.rows(0, 3, False)[1], m[0, 1]) and list(rows[0]) == m[1, 0].tolist() and rows[1][3] == m[1, 1, 3]

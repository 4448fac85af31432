"""ctypes binding of the CPU oracle (oracle/gat_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package (gat_amd/).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SEG = np.dtype([("start", "<u4"), ("end", "<u4")])

COUNTER_IDS = {
    "nucleotide-overlap": 0,
    "nucleotide-density": 1,
    "segment-overlap": 2,
    "segment-midoverlap": 3,
    "annotation-overlap": 4,
    "annotation-midoverlap": 5,
}


class RNG(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("mti", C.c_int), ("ndraws", C.c_uint64)]


class Problem(C.Structure):
    _fields_ = [
        ("n_units", C.c_int32),
        ("segs", C.c_void_p),
        ("seg_off", C.c_void_p),
        ("ws", C.c_void_p),
        ("ws_off", C.c_void_p),
        ("unit_contig", C.c_void_p),
        ("n_contigs", C.c_int32),
        ("merge_contigs", C.c_int32),
        ("n_tracks", C.c_int32),
        ("annos", C.c_void_p),
        ("anno_off", C.c_void_p),
        ("cws_nseg", C.c_void_p),
        ("bucket_size", C.c_uint32),
        ("nbuckets", C.c_int32),
        ("sampler", C.c_int32),
    ]


def build(force=False):
    """compile oracle/libgat_oracle.so (and oracle/_ref when the reference tree is present)."""
    so = os.path.join(_HERE, "libgat_oracle.so")
    src = os.path.join(_HERE, "gat_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        L = C.CDLL(so)
        vp, sz, u32, i32, i64 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int32, C.c_int64
        L.gato_searchsorted_u32.restype = C.c_long
        L.gato_searchsorted_u32.argtypes = [vp, sz, u32]
        L.gato_searchsorted_seg.restype = C.c_long
        L.gato_searchsorted_seg.argtypes = [vp, sz, u32]
        for name in ("gato_normalize",):
            getattr(L, name).restype = sz
            getattr(L, name).argtypes = [vp, sz]
        L.gato_merge.restype = sz
        L.gato_merge.argtypes = [vp, sz, i32]
        L.gato_check.restype = C.c_int
        L.gato_check.argtypes = [vp, sz]
        L.gato_filter.restype = sz
        L.gato_filter.argtypes = [vp, sz, vp, sz]
        L.gato_intersect.restype = C.c_long
        L.gato_intersect.argtypes = [vp, sz, vp, sz, vp, sz]
        L.gato_sum.restype = u32
        L.gato_sum.argtypes = [vp, sz]
        L.gato_overlap_with_segments.restype = u32
        L.gato_overlap_with_segments.argtypes = [vp, sz, vp, sz]
        L.gato_intersection_with_segments.restype = u32
        L.gato_intersection_with_segments.argtypes = [vp, sz, vp, sz, C.c_int]
        L.gato_get_insertion_point.restype = C.c_int
        L.gato_get_insertion_point.argtypes = [vp, sz, u32, u32]
        L.gato_trim_ends.restype = C.c_int
        L.gato_trim_ends.argtypes = [vp, sz, u32, u32, C.c_int]
        L.gato_length_distribution.restype = C.c_int
        L.gato_length_distribution.argtypes = [vp, sz, u32, C.c_int, vp, C.POINTER(u32)]
        L.gato_rng_seed.restype = None
        L.gato_rng_seed.argtypes = [C.POINTER(RNG), u32]
        L.gato_rng_u32.restype = u32
        L.gato_rng_u32.argtypes = [C.POINTER(RNG)]
        L.gato_randint.restype = i64
        L.gato_randint.argtypes = [C.POINTER(RNG), i64, i64]
        L.gato_sampler_annotator.restype = C.c_int
        L.gato_sampler_annotator.argtypes = [C.POINTER(RNG), vp, sz, vp, sz, u32, C.c_int, vp, sz,
                                             C.POINTER(sz), C.POINTER(C.c_int)]
        L.gato_sampler_segments.restype = C.c_int
        L.gato_sampler_segments.argtypes = [C.POINTER(RNG), vp, sz, vp, sz, u32, C.c_int, vp, sz, C.POINTER(sz)]
        L.gato_counter.restype = C.c_double
        L.gato_counter.argtypes = [C.c_int, vp, sz, vp, sz, i64]
        L.gato_run_samples.restype = C.c_int
        L.gato_run_samples.argtypes = [C.POINTER(Problem), vp, C.c_int, u32, C.c_int, i64, i64, vp, vp, i64, vp]
        L.gato_two_sided_pvalue.restype = C.c_double
        L.gato_two_sided_pvalue.argtypes = [vp, C.c_long, C.c_double, C.c_double]
        _LIB = L
    return _LIB


def segs(x):
    """any iterable of (start, end) / SEG array -> contiguous SEG array (copy)."""
    if isinstance(x, np.ndarray) and x.dtype == SEG:
        return np.ascontiguousarray(x).copy()
    a = np.asarray(list(x), dtype=np.int64).reshape(-1, 2)
    out = np.empty(len(a), dtype=SEG)
    out["start"] = a[:, 0].astype(np.uint32)
    out["end"] = a[:, 1].astype(np.uint32)
    return out


def aslist(a):
    return [(int(s), int(e)) for s, e in zip(a["start"], a["end"])]


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def normalize(x):
    a = segs(x)
    n = lib().gato_normalize(_p(a), len(a))
    return a[:n].copy()


def merge(x, distance):
    a = segs(x)
    n = lib().gato_merge(_p(a), len(a), distance)
    return a[:n].copy()


def check(x):
    a = segs(x)
    return bool(lib().gato_check(_p(a), len(a)))


def filter(x, other):  # noqa: A001 - mirrors SegmentList.filter
    a, b = segs(x), segs(other)
    n = lib().gato_filter(_p(a), len(a), _p(b), len(b))
    return a[:n].copy()


def intersect(x, other):
    a, b = segs(x), segs(other)
    out = np.empty(len(a) + len(b) + 1, dtype=SEG)
    n = lib().gato_intersect(_p(a), len(a), _p(b), len(b), _p(out), len(out))
    assert n >= 0
    return out[:n].copy()


def total(x):
    a = segs(x)
    return int(lib().gato_sum(_p(a), len(a)))


def overlap_with_segments(x, other):
    a, b = segs(x), segs(other)
    return int(lib().gato_overlap_with_segments(_p(a), len(a), _p(b), len(b)))


def intersection_with_segments(x, other, mode="base"):
    a, b = segs(x), segs(other)
    return int(lib().gato_intersection_with_segments(_p(a), len(a), _p(b), len(b), int(mode == "midpoint")))


def get_insertion_point(x, start, end):
    a = segs(x)
    return int(lib().gato_get_insertion_point(_p(a), len(a), start, end))


def trim_ends(x, pos, size, forward):
    a = segs(x)
    rc = lib().gato_trim_ends(_p(a), len(a), pos, size, int(forward))
    if rc:
        raise AssertionError("trim_ends rc=%d" % rc)
    return a


def length_distribution(x, bucket_size=0, nbuckets=100000):
    a = segs(x)
    hist = np.zeros(nbuckets, dtype=np.int64)
    b = C.c_uint32(0)
    rc = lib().gato_length_distribution(_p(a), len(a), bucket_size, nbuckets, _p(hist), C.byref(b))
    if rc:
        raise ValueError("segment too large for nbuckets*bucket_size")
    return hist, int(b.value)


class RandomState:
    """numpy legacy RandomState restated (seed / randint only)."""

    def __init__(self, seed):
        self.state = RNG()
        lib().gato_rng_seed(C.byref(self.state), seed)

    def seed(self, seed):
        lib().gato_rng_seed(C.byref(self.state), seed)

    def u32(self):
        return int(lib().gato_rng_u32(C.byref(self.state)))

    def randint(self, lo, hi):
        return int(lib().gato_randint(C.byref(self.state), lo, hi))

    @property
    def ndraws(self):
        return int(self.state.ndraws)


def sampler_annotator(rng, segments, workspace, bucket_size=0, nbuckets=100000):
    """SamplerAnnotator(bucket_size, nbuckets).sample(segments, workspace) on `rng`."""
    a, w = segs(segments), segs(workspace)
    cap = 4 * len(a) + 1024
    while True:
        out = np.empty(cap, dtype=SEG)
        n = C.c_size_t(0)
        nun = C.c_int(0)
        save = RNG.from_buffer_copy(rng.state)
        rc = lib().gato_sampler_annotator(C.byref(rng.state), _p(a), len(a), _p(w), len(w), bucket_size, nbuckets,
                                          _p(out), cap, C.byref(n), C.byref(nun))
        if rc == -3:
            rng.state = save
            cap *= 2
            continue
        break
    if rc == -1:
        raise ValueError("oracle sampler: ValueError")
    if rc:
        raise AssertionError("oracle sampler rc=%d" % rc)
    return out[: n.value].copy(), nun.value


def sampler_segments(rng, segments, workspace, bucket_size=0, nbuckets=100000):
    """SamplerSegments(bucket_size, nbuckets).sample(segments, workspace) on `rng`."""
    a, w = segs(segments), segs(workspace)
    out = np.empty(len(a) + 1, dtype=SEG)
    n = C.c_size_t(0)
    rc = lib().gato_sampler_segments(C.byref(rng.state), _p(a), len(a), _p(w), len(w), bucket_size, nbuckets,
                                     _p(out), len(out), C.byref(n))
    if rc == -1:
        raise ValueError("oracle sampler: ValueError")
    if rc:
        raise AssertionError("oracle sampler rc=%d" % rc)
    return out[: n.value].copy()


def counter(name, segments, annotations, ws_nseg=1):
    a, b = segs(segments), segs(annotations)
    return float(lib().gato_counter(COUNTER_IDS[name], _p(a), len(a), _p(b), len(b), ws_nseg))


def run_samples(flat, counters, seed, stream_mode, sample_begin, sample_end, want_samples=False):
    """Run computeSample over [sample_begin, sample_end) on a flat problem.

    `flat` is a dict of numpy arrays with the fields of gato_problem (see gat_oracle.h);
    returns (counts, samples) where counts is a list (per counter) of [n_tracks, n_samples]
    arrays (int64, or float64 for nucleotide-density)."""
    L = lib()
    keep = {}

    def arr(name, dtype):
        keep[name] = np.ascontiguousarray(flat[name], dtype=dtype)
        return _p(keep[name])

    p = Problem()
    p.n_units = int(flat["n_units"])
    p.segs = arr("segs", SEG)
    p.seg_off = arr("seg_off", np.int64)
    p.ws = arr("ws", SEG)
    p.ws_off = arr("ws_off", np.int64)
    p.unit_contig = arr("unit_contig", np.int32)
    p.n_contigs = int(flat["n_contigs"])
    p.merge_contigs = int(flat["merge_contigs"])
    p.n_tracks = int(flat["n_tracks"])
    p.annos = arr("annos", SEG)
    p.anno_off = arr("anno_off", np.int64)
    p.cws_nseg = arr("cws_nseg", np.int64)
    p.bucket_size = int(flat.get("bucket_size", 0))
    p.nbuckets = int(flat.get("nbuckets", 100000))
    p.sampler = int(flat.get("sampler", 0))
    ids = np.array([COUNTER_IDS[c] for c in counters], dtype=np.int32)
    ns = sample_end - sample_begin
    counts = np.zeros((len(ids), p.n_tracks, ns), dtype=np.int64)
    samples = soff = None
    cap = 0
    if want_samples:
        cap = int(4 * len(keep["segs"]) + 1024) * ns
        soff = np.zeros(ns * p.n_contigs + 1, dtype=np.int64)
    while True:
        if want_samples:
            samples = np.empty(cap, dtype=SEG)
        rc = L.gato_run_samples(C.byref(p), _p(ids), len(ids), seed, stream_mode, sample_begin, sample_end,
                                _p(counts), _p(samples) if want_samples else None, cap,
                                _p(soff) if want_samples else None)
        if rc == -3 and want_samples and cap < (1 << 33):   # a few long segments can come back as thousands of short ones
            cap *= 4
            continue
        break
    if rc == -1:
        raise ValueError("oracle run_samples: ValueError")
    if rc:
        raise AssertionError("oracle run_samples rc=%d" % rc)
    out = []
    for k, c in enumerate(counters):
        out.append(counts[k].view(np.float64).copy() if c == "nucleotide-density" else counts[k].copy())
    if want_samples:
        return out, (samples[: soff[-1]].copy(), soff)
    return out, None


def two_sided_pvalue(sorted_samples, expected, val):
    a = np.ascontiguousarray(sorted_samples, dtype=np.float64)
    return float(lib().gato_two_sided_pvalue(_p(a), len(a), expected, val))


def ref_searchsorted_u32(base, target):
    """the reference's own utils/gat_utils.c searchsorted (oracle/_ref build), or None if absent."""
    so = os.path.join(_HERE, "_ref", "libgat_utils_ref.so")
    if not os.path.exists(so):
        return None
    L = C.CDLL(so)
    cmp_t = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)

    def cmp_position(a, b):  # gat/Engine.pyx:119 cmpPosition
        x = C.cast(a, C.POINTER(C.c_uint32))[0]
        y = C.cast(b, C.POINTER(C.c_uint32))[0]
        d = (x - y) & 0xFFFFFFFF
        return d - (1 << 32) if d >= (1 << 31) else d

    L.searchsorted.restype = C.c_long
    L.searchsorted.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, cmp_t]
    b = np.ascontiguousarray(base, dtype=np.uint32)
    t = C.c_uint32(target)
    return int(L.searchsorted(_p(b), len(b), 4, C.byref(t), cmp_t(cmp_position)))

"""Host-side mirror of the reference's operator interface for the hot path (gat/Engine.pyx,
gat/SegmentList.pyx): same class and method names, argument meaning and error behaviour, so that
code written against `gat.Engine` / `gat.SegmentList` reads the same against `gat_amd`.

Containers are thin numpy wrappers; everything the hot path computes per sample (placement,
consolidation, fromIsochores on samples, counters) is done by the HIP kernels behind
libgat_mi355.so -- see gat_amd/__init__.py:run and gat_amd/_lib.py.  There is no CPU fallback.
"""
import collections
import math
import operator

import numpy as np

from . import _lib
from . import intervals as iv
from . import problem as _problem

SEG = iv.SEG

_CTX = {}
_DEFAULT_DEVICE = None


def set_device(device):
    """the HIP device every later get_context() without an argument uses (gat-run.py --device)."""
    global _DEFAULT_DEVICE
    _DEFAULT_DEVICE = int(device)


def default_device():
    """process default: set_device() / the first explicit get_context(device); under an initialised
    torch.distributed one rank drives one GPU, LOCAL_RANK; else device 0."""
    if _DEFAULT_DEVICE is not None:
        return _DEFAULT_DEVICE
    import os
    import sys
    dist = sys.modules.get("torch.distributed")           # (never imported here: see gat_amd._dist_state)
    if dist is not None and dist.is_available() and dist.is_initialized() and "LOCAL_RANK" in os.environ:
        return int(os.environ["LOCAL_RANK"])
    return 0


def get_context(device=None, stream=None):
    """process-wide gat_ctx per (device, stream); device None = the process default (see default_device)."""
    global _DEFAULT_DEVICE
    if device is None:
        device = default_device()
    elif _DEFAULT_DEVICE is None:
        _DEFAULT_DEVICE = int(device)
    key = (int(device), stream)
    if key not in _CTX:
        _CTX[key] = _lib.Context(int(device), stream)
    return _CTX[key]


# ------------------------------------------------------------------------------------------------
class SegmentList(object):
    """list of half-open (start, end) uint32 segments (gat/SegmentList.pyx:146).

    "normalized" = sorted by start, non-overlapping, no empty segments."""

    is_points = False

    def __init__(self, allocate=0, clone=None, iter=None, normalize=False, array=None):  # noqa: A002
        self.isNormalized = 1
        if clone is not None:
            self._a = clone._a.copy()
            self.isNormalized = clone.isNormalized
        elif array is not None:
            self._a = np.ascontiguousarray(array, dtype=SEG)
            self.isNormalized = 0
        elif iter is not None:
            self._a = iv.as_segments(iter)
            self.isNormalized = 0 if len(self._a) else 1
        else:
            self._a = iv.EMPTY.copy()
        if normalize:
            self.normalize()

    # --- container protocol
    def __len__(self):
        return len(self._a)

    def __iter__(self):
        for s, e in zip(self._a["start"].tolist(), self._a["end"].tolist()):
            yield (s, e)

    def __getitem__(self, i):
        return (int(self._a["start"][i]), int(self._a["end"][i]))

    def __eq__(self, other):
        return isinstance(other, SegmentList) and np.array_equal(self._a, other._a)

    def __str__(self):
        return str(self.asList())

    @property
    def isEmpty(self):
        return len(self._a) == 0

    def asList(self):
        return list(self)

    def asArray(self):
        return self._a

    def asLengths(self):
        return (self._a["end"].astype(np.int64) - self._a["start"]).tolist()

    # --- construction
    def add(self, start, end):
        assert start <= end, "attempting to add invalid segment %i-%i" % (start, end)
        if start < 0 or end >= (1 << 32):
            raise OverflowError("can't convert negative value to unsigned int")
        self._a = np.concatenate([self._a, iv.make([start], [end])])
        self.isNormalized = 0

    def extend(self, other):
        self._a = np.concatenate([self._a, other._a])
        self.isNormalized = 0
        return self

    def clear(self):
        self._a = iv.EMPTY.copy()
        self.isNormalized = 1

    def extend_segments(self, extension):
        """grow every interval by `extension` on both sides, truncating at 0; the list is not
        normalized afterwards (gat/SegmentList.pyx:1551-1565)."""
        start = self._a["start"].astype(np.int64)
        end = self._a["end"].astype(np.int64)
        start -= np.minimum(int(extension), start)
        end += int(extension)
        self._a = iv.make(start & 0xFFFFFFFF, end & 0xFFFFFFFF)
        self.isNormalized = 0

    def expand_segments(self, expansion):
        """scale every interval by `expansion` around its midpoint (gat/SegmentList.pyx:1567-1591)."""
        if expansion <= 0:
            raise ValueError("invalid expansion: %f <= 0" % expansion)
        e_1 = (float(expansion) - 1.0) / 2.0
        start = self._a["start"].astype(np.int64)
        end = self._a["end"].astype(np.int64)
        length = (end - start).astype(np.int32).astype(np.int64)                  # PositionDifference l
        extension = np.floor(length.astype(np.float64) * e_1).astype(np.int64)
        start -= np.minimum(extension, start)
        end += extension
        self._a = iv.make(start & 0xFFFFFFFF, end & 0xFFFFFFFF)
        self.isNormalized = 0

    def clone(self):
        return SegmentList(clone=self)

    # --- algebra (gat/SegmentList.pyx:697, :756, :1401, :1469)
    def sort(self):
        self._a = self._a[np.argsort(self._a["start"], kind="stable")]

    def normalize(self):
        self._a = iv.normalize(self._a)
        self.isNormalized = 1

    def merge(self, distance):
        self._a = iv.merge(self._a, distance)
        self.isNormalized = 1

    def check(self):
        self.isNormalized = 1 if iv.is_normalized(self._a) else 0
        if not self.isNormalized:
            raise ValueError("segment list is not normalized")
        return self.isNormalized

    def filter(self, other):  # noqa: A003
        if other is self:
            self.clear()
            return
        self._a = iv.filter(self._a, other._a)

    def intersect(self, other):
        if getattr(other, "is_points", False):
            raise _points_type_error()
        assert self.isNormalized, "intersection of a non-normalized list"
        assert other.isNormalized, "intersection with non-normalized list"
        if other is self:
            return
        self._a = iv.intersect(self._a, other._a)

    def sum(self):  # noqa: A003
        return iv.total(self._a)

    def max(self):  # noqa: A003
        assert self.isNormalized, "maximum from non-normalized list"
        return int(self._a["end"][-1]) if len(self._a) else 0

    def min(self):  # noqa: A003
        assert self.isNormalized, "minimum from non-normalized list"
        return int(self._a["start"][0]) if len(self._a) else 0

    def largest(self):
        if len(self._a) == 0:
            raise ValueError("largest segment from empty list")
        lengths = self._a["end"].astype(np.int64) - self._a["start"]
        i = int(np.argmax(lengths))
        return self[i]

    def getLengthDistribution(self, bucket_size=0, nbuckets=100000):
        """gat/SegmentList.pyx:1148-1184."""
        assert bucket_size >= 0, "bucket_size is 0"
        assert nbuckets > 0, "nbuckets is 0"
        lengths = self._a["end"].astype(np.int64) - self._a["start"]
        if bucket_size == 0:
            s, e = self.largest()
            bucket_size = int(math.ceil((e - s) / float(nbuckets)))
        idx = (lengths + bucket_size - 1) // bucket_size
        if len(idx) and idx.max() >= nbuckets:
            bad = int(np.argmax(idx >= nbuckets))
            raise ValueError("segment %i-%i too large: increase nbuckets (%i) or bucket_size (%i)" %
                             (self._a["start"][bad], self._a["end"][bad], nbuckets, bucket_size))
        return np.bincount(idx, minlength=nbuckets).astype(np.int64), bucket_size

    # --- counters' primitives (device)
    def overlapWithSegments(self, other):
        """gat/SegmentList.pyx:1026-1076, on the GPU."""
        assert self.isNormalized and other.isNormalized, "intersection from non-normalized list"
        if other is self:
            return self.sum()
        return int(_count_pair("nucleotide-overlap", other, self))

    def intersectionWithSegments(self, other, mode="base"):
        """gat/SegmentList.pyx:1078-1146, on the GPU."""
        assert self.isNormalized and other.isNormalized, "intersection from non-normalized list"
        if other is self:
            return self.sum()
        return int(_count_pair("segment-midoverlap" if mode == "midpoint" else "segment-overlap", self, other))


def _count_pair(counter, segments, annotations, ws_nseg=1):
    ctx = get_context()
    a, b = segments._a, annotations._a
    r = ctx.count_lists([counter], a, [0, len(a)], 1, b, [0, len(b)], 1, [ws_nseg], 1)
    return r[0][0, 0]


POINT_COUNTERS = ("annotation-overlap", "annotation-midoverlap")


def _points_type_error(what="other"):
    """what the reference raises when a PositionList reaches code typed for SegmentList (every counter but the two
    annotation counters, fromIsochores, merge: gat/Engine.pyx:2200, :2866, :3007)."""
    return TypeError("Argument '%s' has incorrect type (expected gat.SegmentList.SegmentList, got "
                     "gat.PositionList.PositionList)" % what)


class PositionList(SegmentList):
    """sorted list of positions (gat/PositionList.pyx:42), what --annotations-to-points turns annotation intervals
    into.  A position p is held as the one-base interval [p, p+1): "p lies in a segment" (intersectionWithSegments,
    gat/PositionList.pyx:490-540) is then exactly CounterAnnotationOverlap on that interval, so the flat problem and
    the device kernels take a PositionList as it is."""
    is_points = True

    def __init__(self, allocate=0, clone=None, iter=None, sort=False, normalize=False):  # noqa: A002
        SegmentList.__init__(self, clone=clone)
        if clone is None and iter is not None:
            p = np.fromiter(iter, dtype=np.int64)
            self._a = iv.make(p, p + 1)
            self.isNormalized = 0
        if sort:
            self.sort()
        if normalize:
            self.normalize()

    def fromSegmentList(self, segments, method="midpoint"):
        """gat/PositionList.pyx:288-336: one position per non-empty segment."""
        a = segments._a[segments._a["end"] != segments._a["start"]]
        start, end = a["start"].astype(np.int64), a["end"].astype(np.int64)
        if method == "midpoint":
            p = start + (end - start) // 2
        elif method == "start":
            p = start
        elif method == "end":
            p = end
        else:
            raise ValueError("unknow method '%s'" % method)
        self._a = iv.make(p, p + 1)
        self.isNormalized = 1 if segments.isNormalized else 0

    def __iter__(self):
        return iter(self._a["start"].tolist())

    def __getitem__(self, i):
        return int(self._a["start"][i])

    def asList(self):
        return self._a["start"].tolist()

    def add(self, pos):
        if pos < 0 or pos >= (1 << 32) - 1:
            raise OverflowError("can't convert negative value to unsigned int")
        self._a = np.concatenate([self._a, iv.make([pos], [pos + 1])])
        self.isNormalized = 0

    def clone(self):
        return PositionList(clone=self)

    def sort(self):
        self._a = self._a[np.argsort(self._a["start"], kind="stable")]

    def normalize(self):
        """sort, equal positions once (gat/PositionList.pyx:350-374)."""
        self._a = np.unique(self._a)
        self.isNormalized = 1

    def sum(self):  # noqa: A003
        """the number of positions (gat/PositionList.pyx:237)."""
        return len(self._a)

    def max(self):  # noqa: A003
        return int(self._a["start"][-1]) if len(self._a) else 0

    def min(self):  # noqa: A003
        return int(self._a["start"][0]) if len(self._a) else 0

    def intersect(self, other):
        """keep the positions that lie in a segment of `other` (gat/PositionList.pyx:542-585)."""
        if getattr(other, "is_points", False):
            raise _points_type_error()
        assert other.isNormalized, "Intersection with non-normalized segments"
        self._a = iv.intersect(self._a, other._a)

    def intersectionWithSegments(self, other, mode="base"):
        """number of positions inside segments of `other`; `mode` is unused (gat/PositionList.pyx:490-540)."""
        assert other.isNormalized, "Intersection with non-normalized segments"
        return int(_count_pair("annotation-overlap", other, self))

    def _unsupported(self, *args, **kwargs):
        raise _points_type_error()
    # the reference's PositionList has none of these; what reaches for them there dies with a TypeError / AttributeError
    filter = merge = extend = overlapWithSegments = extend_segments = expand_segments = getLengthDistribution = _unsupported


# ------------------------------------------------------------------------------------------------
class _DictFlat(object):
    """the lists of an IntervalDictionary in ONE array: list i (key keys[i]) = data[off[i]:off[i + 1]], and the lists hold
    exactly those views (`arrays`), which is how a later look finds out whether a list was replaced since.  A run over an
    isochore-partitioned collection touches 10^4 small lists; every pass over them that makes a numpy call per list costs
    more than the sampling on the device, so the passes work on this form (IntervalDictionary._flat)."""
    __slots__ = ("data", "off", "keys", "arrays", "index")

    def __init__(self, data, off, keys, arrays):
        self.data, self.off, self.keys, self.arrays = data, off, keys, arrays
        self.index = None

    def position(self, key):
        if self.index is None:
            self.index = dict((k, i) for i, k in enumerate(self.keys))
        return self.index.get(key, -1)

    def ranges(self, target_keys, base=0):
        """(begin, end) of the lists of `target_keys` in `data` (+ base); a key this dictionary lacks gives (0, 0)"""
        if target_keys == self.keys:
            return self.off[:-1] + base, self.off[1:] + base
        pos = np.fromiter((self.position(k) for k in target_keys), dtype=np.int64, count=len(target_keys))
        have = pos >= 0
        p = np.where(have, pos, 0)
        return np.where(have, self.off[p] + base, 0), np.where(have, self.off[p + 1] + base, 0)


_GET_A = operator.attrgetter("_a")
_GET_NORMALIZED = operator.attrgetter("isNormalized")
_GET_POINTS = operator.attrgetter("is_points")


class _IsochorePrep(object):
    """the isochore classes of a toIsochores call, for all contigs at once: their segments in coordinates
    (contig number << 32) + position, sorted by start, each with its class.  valid: every class list is a normalized
    SegmentList and no two segments overlap (isochores partition a contig) -- what the vectorised split relies on.  Shared
    by the dictionaries of a collection (IntervalCollection.toIsochores)."""

    def __init__(self, tracks):
        self.ids = {}
        self.valid = len(tracks) >= 1
        starts, ends, labels = [], [], []
        for k, (_, vv) in enumerate(tracks):
            if not isinstance(vv, IntervalDictionary) or vv._has_points() or not vv._all_normalized():
                self.valid = False
                return
            f = vv._flat()
            if len(f.data) == 0:
                continue
            cid = np.fromiter((self.contig_id(c) for c in f.keys), dtype=np.int64, count=len(f.keys))
            hi = np.repeat(cid << 32, np.diff(f.off))
            starts.append(hi + f.data["start"])
            ends.append(hi + f.data["end"])
            labels.append(np.full(len(f.data), k, dtype=np.int64))
        if starts:
            b_s, b_e, lab = np.concatenate(starts), np.concatenate(ends), np.concatenate(labels)
            order = np.argsort(b_s, kind="stable")
            self.b_start, self.b_end, self.label = b_s[order], b_e[order], lab[order]
            if bool(np.any(self.b_start >= self.b_end)) or bool(np.any(self.b_end[:-1] > self.b_start[1:])):
                self.valid = False
        else:
            self.b_start = self.b_end = self.label = np.empty(0, dtype=np.int64)

    def contig_id(self, contig):
        i = self.ids.get(contig)
        if i is None:
            i = self.ids[contig] = len(self.ids)
        return i


def _new_list(array, like):
    """a list of `like`'s class and flag around `array` (what clone() + a replacing operation leave)"""
    s = like.__class__.__new__(like.__class__)
    s._a = array
    s.isNormalized = like.isNormalized
    return s


_MISSING = object()


class _LazyLists(collections.defaultdict):
    """key -> SegmentList whose lists are slices of ONE array (data[off[i]:off[i + 1]] for key i) and are only made when
    somebody asks for them: toIsochores of config 3's annotations leaves 19 200 lists, and making 19 200 Python objects costs
    more than splitting a million intervals (the run itself works on the array: IntervalDictionary._flat).  Until it is made a
    key holds None in the underlying dict; every way to a value goes through here.  `protos[i // per]` is the list key i was
    split from: the new list takes its class and its isNormalized."""

    def __init__(self, keys, data, off, protos, per):
        super().__init__(SegmentList)
        dict.update(self, dict.fromkeys(keys))
        self._keys, self._data, self._off, self._protos, self._per = keys, data, off, protos, per
        self._index = None
        self._made = []                 # (key, the view it was given): how _flat() finds out whether a list was replaced since
        self._touched = False           # a key was set or deleted: the flat form has to be looked at again

    def _make(self, key):
        if self._index is None:
            self._index = dict(zip(self._keys, range(len(self._keys))))
        i = self._index[key]
        w = self._data[self._off[i]:self._off[i + 1]]
        s = _new_list(w, self._protos[i // self._per])
        dict.__setitem__(self, key, s)
        self._made.append((key, w))
        return s

    def _make_all(self):
        if len(self._made) < len(self._keys):
            get = dict.get
            for k in self._keys:
                if get(self, k, _MISSING) is None:
                    self._make(k)

    def pristine(self):
        """nothing set or deleted, and every list that was made still holds the view it was given"""
        if self._touched:
            return False
        get = dict.__getitem__
        return all(get(self, k)._a is w for k, w in self._made)

    def all_normalized(self):
        return all(p.isNormalized for p in self._protos) and all(dict.__getitem__(self, k).isNormalized for k, _ in self._made)

    def __iter__(self):
        # A Python-level __iter__ takes dict(d) / OrderedDict(d) / {**d} / other.update(d) off CPython's fast path for dict
        # subclasses, which copies the RAW values -- the None placeholders -- without asking anybody: with it they go through
        # keys() and [], and get the lists.  Iterating makes nothing by itself.
        return dict.__iter__(self)

    def __getitem__(self, key):
        v = dict.get(self, key, _MISSING)
        if v is None:
            return self._make(key)
        if v is _MISSING:
            self._touched = True
            return self.__missing__(key)
        return v

    def get(self, key, default=None):
        v = dict.get(self, key, _MISSING)
        if v is None:
            return self._make(key)
        return default if v is _MISSING else v

    def __setitem__(self, key, value):
        self._touched = True
        dict.__setitem__(self, key, value)

    def __delitem__(self, key):
        self._touched = True
        dict.__delitem__(self, key)

    def values(self):
        self._make_all()
        return dict.values(self)

    def items(self):
        self._make_all()
        return dict.items(self)

    def pop(self, *args):
        self._make_all()
        self._touched = True
        return dict.pop(self, *args)

    def popitem(self):
        self._make_all()
        self._touched = True
        return dict.popitem(self)

    def setdefault(self, key, default=None):
        self._make_all()
        self._touched = True
        return dict.setdefault(self, key, default)

    def update(self, *args, **kwargs):
        self._touched = True
        dict.update(self, *args, **kwargs)

    def clear(self):
        self._touched = True
        self._made = []
        self._keys = []
        dict.clear(self)

    def copy(self):
        self._make_all()
        return collections.defaultdict(SegmentList, dict.items(self))

    __copy__ = copy

    def __eq__(self, other):
        self._make_all()
        return dict.__eq__(self, other)

    def __ne__(self, other):
        return not self.__eq__(other)

    def __reduce__(self):
        """pickled / copied as the plain dictionary it stands for"""
        self._make_all()
        return (collections.defaultdict, (SegmentList,), None, None, iter(list(dict.items(self))))

    __hash__ = None


class IntervalDictionary(object):
    """key (contig or contig.isochore) -> SegmentList (gat/Engine.pyx:2741)."""

    def __init__(self, name=None):
        self.intervals = collections.defaultdict(SegmentList)
        self.name = name
        self._flat_cache = None

    def _flat(self):
        """the lists as one array (_DictFlat); built once and found again as long as no list was replaced, added or removed"""
        f = self._flat_cache
        d = self.intervals
        if f is not None and f.arrays is None:
            # (the lists of a toIsochores pass, most of them not made yet: _LazyLists)
            if isinstance(d, _LazyLists) and d.pristine() and len(d) == len(f.keys):
                return f
            f = None
        if f is not None and len(f.arrays) == len(d) and all(map(operator.is_, map(_GET_A, d.values()), f.arrays)) \
                and list(d.keys()) == f.keys:
            return f
        vals = list(d.values())
        arrays = [v._a for v in vals]
        off = np.zeros(len(arrays) + 1, dtype=np.int64)
        if arrays:
            np.cumsum(np.fromiter((len(a) for a in arrays), dtype=np.int64, count=len(arrays)), out=off[1:])
        data = np.concatenate(arrays) if off[-1] else iv.EMPTY.copy()
        o = off.tolist()
        views = [data[o[i]:o[i + 1]] for i in range(len(arrays))]
        for v, w in zip(vals, views):
            v._a = w                                   # same content; from now on the list IS its slice of `data`
        f = self._flat_cache = _DictFlat(data, off, list(d.keys()), views)
        return f

    def _all_normalized(self):
        d = self.intervals
        if isinstance(d, _LazyLists) and not d._touched:
            return d.all_normalized()
        return all(map(_GET_NORMALIZED, d.values()))

    def _has_points(self):
        d = self.intervals
        if isinstance(d, _LazyLists) and not d._touched:
            return False                                   # (toIsochores' one-pass form does not take points)
        return any(map(_GET_POINTS, d.values()))

    def __len__(self):
        return len(self.intervals)

    def __getitem__(self, key):
        return self.intervals[key]

    def __setitem__(self, key, val):
        self.intervals[key] = val

    def __delitem__(self, key):
        del self.intervals[key]

    def __contains__(self, key):
        return key in self.intervals

    def keys(self):
        return self.intervals.keys()

    def items(self):
        return self.intervals.items()

    def add(self, contig, segmentlist):
        self.intervals[contig] = segmentlist

    def sum(self):  # noqa: A003
        if len(self.intervals) < 16 or self._has_points():
            return sum(x.sum() for x in self.intervals.values())
        # SegmentList.sum() list by list (a uint32 accumulator each, gat/SegmentList.pyx:1607), then Python's sum
        f = self._flat()
        run = np.zeros(len(f.data) + 1, dtype=np.int64)
        np.cumsum(f.data["end"].astype(np.int64) - f.data["start"], out=run[1:])
        return int(((run[f.off[1:]] - run[f.off[:-1]]) & 0xFFFFFFFF).sum())

    def counts(self):
        if len(self.intervals) < 16:
            return sum(len(x) for x in self.intervals.values())
        return int(self._flat().off[-1])

    def clone(self):
        r = IntervalDictionary(self.name)
        for k, v in self.intervals.items():
            r[k] = v.clone()
        return r

    def normalize(self, remove_empty=False):
        """gat/Engine.pyx:2731 (a dictionary keeps its empty keys); IntervalCollection.normalize
        (:2942) drops them."""
        if remove_empty:
            for k in [k for k, v in self.intervals.items() if len(v) == 0]:
                del self.intervals[k]
        for v in self.intervals.values():
            v.normalize()

    def extend(self, extension):
        """gat/Engine.pyx:2719."""
        for v in self.intervals.values():
            v.extend_segments(extension)

    def expand(self, expansion):
        """gat/Engine.pyx:2725."""
        for v in self.intervals.values():
            v.expand_segments(expansion)

    def _pairwise(self, other, op):
        for contig in list(self.intervals.keys()):
            if contig in other:
                op(self.intervals[contig], other[contig])
            else:
                del self.intervals[contig]

    def intersect(self, other):
        self._pairwise(other, lambda a, b: a.intersect(b))

    def filter(self, other):  # noqa: A003
        self._pairwise(other, lambda a, b: a.filter(b))

    def _flat64(self, order):
        """all lists as one pair of sorted int64 arrays, the position of the key in `order` in the high bits."""
        starts, ends = [], []
        for i, k in enumerate(order):
            v = self.intervals.get(k)
            if v is None or len(v) == 0:
                continue
            starts.append(v._a["start"].astype(np.int64) + (i << 33))
            ends.append(v._a["end"].astype(np.int64) + (i << 33))
        if not starts:
            return np.empty(0, np.int64), np.empty(0, np.int64)
        return np.concatenate(starts), np.concatenate(ends)

    def intersect_stats(self, other, _cache=None):
        """(counts(), sum()) of `clone(); intersect(other)` without building it: the number of overlapping pairs and
        their total overlap over the keys both dictionaries hold, in two vectorised passes (the per-key loop over
        small lists dominated gat.run() on isochore-partitioned inputs).  None if a list is not normalized (the
        caller then takes the list-by-list path, which raises as the reference does)."""
        if other is self:
            return None
        for d in (self, other):
            if any(not v.isNormalized or getattr(v, "is_points", False) for v in d.intervals.values()):
                return None
        # keys only one side holds contribute nothing, so every key of self keeps its own position and the flat
        # form of self can be reused for every `other` of a run
        order = list(self.intervals.keys())
        ck = ("flat", id(self))
        if _cache is not None and ck in _cache:
            a_s, a_e = _cache[ck]
        else:
            a_s, a_e = self._flat64(order)
            if _cache is not None:
                _cache[ck] = (a_s, a_e)
        b_s, b_e = other._flat64(order)
        if len(a_s) == 0 or len(b_s) == 0:
            return 0, 0
        j0 = np.searchsorted(b_e, a_s, side="right")
        j1 = np.searchsorted(b_s, a_e, side="left")
        npairs = int(np.maximum(j1 - j0, 0).sum())
        cum = np.concatenate([[0], np.cumsum(b_e - b_s)])

        def below(x):                                  # bases of `other` below position x
            k = np.searchsorted(b_s, x, side="left")
            last = np.maximum(k - 1, 0)
            return np.where(k > 0, cum[k] - (b_e[last] - np.minimum(x, b_e[last])), 0)
        return npairs, int((below(a_e) - below(a_s)).sum())

    def toIsochores(self, isochores, truncate=False, _prep=None):
        """gat/Engine.pyx:2837-2855: every list split by the isochore tracks, key `contig.isochore`.  One vectorised pass
        over the whole dictionary when the isochore classes do not overlap one another (they partition the contigs);
        otherwise list by list as the reference does, assertions included."""
        contigs = list(self.intervals.keys())
        tracks = list(isochores.items())
        for _, other_vv in tracks:
            for contig in contigs:
                other_vv[contig]                       # (a dictionary look-up inserts the missing key, as the reference's does)
        if self._to_isochores_flat(contigs, tracks, truncate, _prep if _prep is not None else _IsochorePrep(tracks)):
            return
        for contig in contigs:
            segmentlist = self.intervals[contig]
            for other_track, other_vv in tracks:
                newlist = segmentlist.clone()
                if truncate:
                    newlist.intersect(other_vv[contig])
                else:
                    newlist.filter(other_vv[contig])
                self.intervals["%s.%s" % (contig, other_track)] = newlist
            del self.intervals[contig]

    def _to_isochores_flat(self, contigs, tracks, truncate, prep):
        """toIsochores for all lists and classes at once, in coordinates (contig number << 32) + position: the classes'
        segments B sorted by start, for every segment of this dictionary the range of B it overlaps, one piece (truncate)
        or one copy per class touched (filter), the pieces brought into (contig, class) order.  False when the shortcut does
        not apply."""
        if not prep.valid or not contigs or self._has_points():
            return False
        if truncate and not self._all_normalized():
            return False
        f = self._flat()
        a = f.data
        na, nk, K = len(a), len(contigs), len(tracks)
        lens = np.diff(f.off)
        rank = np.repeat(np.arange(nk, dtype=np.int64), lens)                 # list (contig position) of every segment
        if na:
            # normalized list by list (what SegmentList.check asks for): the shortcut reads the lists as sorted and disjoint
            if bool(np.any(a["start"] >= a["end"])):
                return False
            inner = np.ones(na, dtype=bool)
            inner[f.off[1:-1][f.off[1:-1] < na]] = False                      # the first segment of a list has no predecessor
            if na > 1 and bool(np.any((a["end"][:-1] > a["start"][1:]) & inner[1:])):
                return False
        cid = np.fromiter((prep.contig_id(c) for c in contigs), dtype=np.int64, count=nk)
        hi = np.repeat(cid << 32, lens)
        a_s = hi + a["start"]
        a_e = hi + a["end"]
        b_s, b_e, label = prep.b_start, prep.b_end, prep.label
        if na and len(b_s):
            j0 = np.searchsorted(b_e, a_s, side="right")
            j1 = np.maximum(np.searchsorted(b_s, a_e, side="left"), j0)
            cnt = j1 - j0
            n = int(cnt.sum())
            ai = np.repeat(np.arange(na, dtype=np.int64), cnt)
            bi = np.repeat(j0, cnt) + (np.arange(n, dtype=np.int64) - np.repeat(np.cumsum(cnt) - cnt, cnt))
            cls = label[bi]
        else:
            ai = bi = cls = np.empty(0, dtype=np.int64)
        if truncate:
            src = ai
            gid = rank[ai] * K + cls
            order = np.argsort(gid, kind="stable")                            # within a group: segment order = position order
            out = iv.make((np.maximum(a_s[ai], b_s[bi]) & 0xFFFFFFFF)[order], (np.minimum(a_e[ai], b_e[bi]) & 0xFFFFFFFF)[order])
        else:
            hit = np.zeros(na * K, dtype=bool)
            hit[ai * K + cls] = True
            idx = np.flatnonzero(hit)                                         # (segment, class) pairs, each once
            src = idx // K
            gid = rank[src] * K + (idx - src * K)
            order = np.argsort(gid, kind="stable")
            out = a[src[order]]
        off = np.zeros(nk * K + 1, dtype=np.int64)
        np.cumsum(np.bincount(gid, minlength=nk * K), out=off[1:])
        o = off.tolist()
        olds = [self.intervals[c] for c in contigs]
        keys, views = [], []
        self.intervals.clear()
        g = 0
        for old, contig in zip(olds, contigs):
            for other_track, _ in tracks:
                key = "%s.%s" % (contig, other_track)
                w = out[o[g]:o[g + 1]]
                self.intervals[key] = _new_list(w, old)
                keys.append(key)
                views.append(w)
                g += 1
        if len(keys) == len(self.intervals):                                  # (no two (contig, class) pairs share a key)
            self._flat_cache = _DictFlat(out, off, keys, views)
        return True

    def _install_split(self, keys, out, off, protos, per):
        """the lists of a toIsochores pass made elsewhere (IntervalCollection.toIsochores: one native call for the whole
        collection): `out` / `off` -- this dictionary's own array and offsets, keys -- contig.class in (contig, class) order"""
        lazy = _LazyLists(keys, out, off.tolist(), protos, per)
        if len(lazy) != len(keys):                       # two (contig, class) pairs share a key: the plain form
            return False
        self.intervals = lazy
        self._flat_cache = _DictFlat(out, off, keys, None)
        return True

    def fromIsochores(self):
        """gat/Engine.pyx:2857-2876."""
        new = collections.defaultdict(SegmentList)
        normalize = False
        for isochore, segmentlist in self.intervals.items():
            isochore = isochore.strip()
            if "." in isochore and isochore != ".":
                contig, _ = isochore.split(".")
                if getattr(segmentlist, "is_points", False):
                    raise _points_type_error()               # new[contig] is a SegmentList (gat/Engine.pyx:2866)
                new[contig].extend(segmentlist)
                normalize = True
            else:
                new[isochore] = segmentlist
        if normalize:
            for x in new.values():
                x.merge(0)
        self.intervals = new

    def asArrays(self):
        return collections.OrderedDict((k, v.asArray()) for k, v in self.intervals.items())


class IntervalCollection(object):
    """track -> IntervalDictionary (gat/Engine.pyx:2887)."""

    def __init__(self, name=None):
        self.intervals = collections.defaultdict(IntervalDictionary)
        self.name = name
        self._flat_cache = None

    def _flat(self, tracks=None, _have=None):
        """(data, bases, flats): the lists of the dictionaries of `tracks` (default: all) in ONE array -- dictionary t's
        flat form (IntervalDictionary._flat) starts at bases[t].  Kept while the dictionaries' own flat forms stay.
        _have: what an earlier call with the same tracks returned, from a caller that knows nothing was touched since (one
        run() looks at the annotations several times; finding 10^4 lists unchanged costs a millisecond each time)."""
        if _have is not None:
            return _have
        tracks = list(self.intervals.keys()) if tracks is None else list(tracks)
        flats = [self.intervals[t]._flat() for t in tracks]
        c = self._flat_cache
        if c is not None and len(c[2]) == len(flats) and all(x is y for x, y in zip(c[2], flats)):
            return c
        bases = np.zeros(len(flats) + 1, dtype=np.int64)
        if flats:
            np.cumsum([len(f.data) for f in flats], out=bases[1:])
        data = np.concatenate([f.data for f in flats]) if bases[-1] else iv.EMPTY.copy()
        c = self._flat_cache = (data, bases, flats)
        return c

    def setName(self, name):
        self.name = name

    def getName(self):
        return self.name

    @property
    def tracks(self):
        return self.intervals.keys()

    def keys(self):
        return self.intervals.keys()

    def items(self):
        return self.intervals.items()

    def __len__(self):
        return len(self.intervals)

    def __getitem__(self, key):
        return self.intervals[key]

    def __delitem__(self, key):
        del self.intervals[key]

    def __contains__(self, key):
        return key in self.intervals

    def add(self, track, contig, segmentlist):
        self.intervals[track][contig] = segmentlist

    def load(self, filenames, allow_multiple=False, ignore_tracks=False):
        """load segments from bed file(s) (gat/Engine.pyx:2918-2922)."""
        from . import io as _io
        self.intervals = _io.readFromBed(filenames, allow_multiple=allow_multiple, ignore_tracks=ignore_tracks)
        self._flat_cache = None

    def sum(self):  # noqa: A003
        return sum(v.sum() for v in self.intervals.values())

    def counts(self):
        return sum(v.counts() for v in self.intervals.values())

    def clone(self):
        new = IntervalCollection(self.name)
        for track, v in self.intervals.items():
            for contig, segmentlist in v.items():
                new.add(track, contig, segmentlist.clone())
        return new

    def normalize(self):
        for vv in self.intervals.values():
            vv.normalize(remove_empty=True)

    def sort(self):
        for vv in self.intervals.values():
            for s in vv.intervals.values():
                s.sort()

    def check(self):
        for vv in self.intervals.values():
            for s in vv.intervals.values():
                s.check()

    def toPositions(self, method="mid-point"):
        """every SegmentList becomes the PositionList of its segments' midpoints / starts / ends
        (gat/Engine.pyx:3103-3109; --annotations-to-points, gat/IO.py:134-135)."""
        for vv in self.intervals.values():
            for contig in list(vv.keys()):
                p = PositionList()
                p.fromSegmentList(vv[contig], method=method)
                vv[contig] = p

    _points_memo = None        # set for the duration of a gat_amd.run() call (the collection is not changed inside it)
    _plain_memo = None         # likewise: every list a normalized SegmentList (no points)
    _ranges_memo = None        # likewise: {keys: (begin, end)} of _ranges

    def _ranges(self, aflat, target_keys):
        """(begin, end): where the lists of `target_keys` lie in aflat's one array, dictionary after dictionary (a key a
        dictionary lacks gives (0, 0)): what the device calls and the input statistics take instead of the lists.  The
        observed counts, the intersections' sizes and the problem ask for the same arrays: kept for the length of a run()."""
        memo = self._ranges_memo
        key = tuple(target_keys)
        if memo is not None and key in memo and memo[key][0] is aflat:
            return memo[key][1], memo[key][2]
        _, bases, flats = aflat
        target_keys = list(target_keys)
        if flats and all(f.keys is flats[0].keys or f.keys == target_keys for f in flats) and flats[0].keys == target_keys:
            n = len(target_keys)
            off = np.concatenate([f.off for f in flats]).reshape(len(flats), n + 1) + bases[:-1, None]
            b, e = np.ascontiguousarray(off[:, :-1]).ravel(), np.ascontiguousarray(off[:, 1:]).ravel()
        elif flats:
            bb, ee = zip(*[f.ranges(target_keys, base) for f, base in zip(flats, bases)])
            b, e = np.concatenate(bb), np.concatenate(ee)
        else:
            b = e = np.zeros(0, dtype=np.int64)
        if memo is not None:
            memo[key] = (aflat, b, e)
        return b, e

    def hasPositions(self):
        if self._points_memo is not None:
            return self._points_memo
        return any(vv._has_points() for vv in self.intervals.values())

    def merge(self, delete=False):
        merged = IntervalDictionary()
        for track in list(self.intervals.keys()):
            for contig, segmentlist in self.intervals[track].items():
                if getattr(segmentlist, "is_points", False):
                    raise _points_type_error()               # merged[contig] is a SegmentList (gat/Engine.pyx:3007)
                merged[contig].extend(segmentlist)
            if delete:
                del self.intervals[track]
        self.intervals["merged"] = merged

    def collapse(self):
        """gat/Engine.pyx:3014-3038: intersection of all tracks -> track 'collapsed'."""
        result = IntervalDictionary()
        contigs = collections.defaultdict(int)
        for vv in self.intervals.values():
            for contig in vv.keys():
                contigs[contig] += 1
        ntracks = len(self.intervals)
        shared = set(x for x, y in contigs.items() if y == ntracks)
        for vv in list(self.intervals.values()):
            for contig, segmentlist in vv.items():
                if contig not in shared:
                    continue
                if contig not in result:
                    result[contig] = segmentlist.clone()
                else:
                    result[contig].intersect(segmentlist)
        self.intervals["collapsed"] = result

    def restrict(self, restrict):
        keep = set([restrict]) if not isinstance(restrict, (list, tuple, set)) else set(restrict)
        for track in [t for t in self.intervals.keys() if t not in keep]:
            del self.intervals[track]

    def intersect(self, other):
        for vv in self.intervals.values():
            vv.intersect(other)

    def filter(self, other):  # noqa: A003
        for vv in self.intervals.values():
            vv.filter(other)

    def toIsochores(self, isochores, truncate=False):
        dicts = list(self.intervals.values())
        if not dicts:
            return
        tracks = list(isochores.items())
        # a look-up of a contig an isochore track lacks adds an empty list to it (gat/Engine.pyx:2846: other_vv[contig] on a
        # defaultdict), in the order the reference meets them: dictionary by dictionary, contig by contig
        seen = set()
        for vv in dicts:
            for contig in vv.keys():
                if contig not in seen:
                    seen.add(contig)
                    for _, other_vv in tracks:
                        other_vv[contig]
        prep = _IsochorePrep(tracks)
        if self._to_isochores_native(dicts, tracks, truncate, prep):
            return
        for vv in dicts:
            vv.toIsochores(isochores, truncate, _prep=prep)

    def _to_isochores_native(self, dicts, tracks, truncate, prep):
        """every list of every dictionary split in ONE call of the library (gat_isochore_split: host threads over the lists,
        two passes -- count, fill -- into one array) where IntervalDictionary._to_isochores_flat's conditions hold for all of
        them: the classes partition the contigs, no points, normalized lists.  The dictionaries receive their lists as slices
        of that array, made on demand (_LazyLists).  False: nothing was done, the dictionaries split themselves."""
        if not prep.valid or len(tracks) < 1:
            return False
        arrays, cids, protos_of = [], [], []
        for vv in dicts:
            if not isinstance(vv, IntervalDictionary) or not vv.intervals or vv._has_points():
                return False
            lists = list(vv.intervals.values())
            for x in lists:
                a = x._a
                if a.dtype != iv.SEG or not a.flags["C_CONTIGUOUS"]:
                    return False
            if truncate and not all(map(_GET_NORMALIZED, lists)):
                return False
            arrays.extend(map(_GET_A, lists))
            cids.extend(prep.contig_id(c) for c in vv.intervals.keys())
            protos_of.append(lists)
        K = len(tracks)
        res = _lib.isochore_split(arrays, cids, prep.b_start, prep.b_end, prep.label, K, truncate)
        if res is None:
            return False
        out, off = res
        names = [t for t, _ in tracks]
        key_cache = {}
        l0 = 0
        plans = []
        for vv, protos in zip(dicts, protos_of):
            contigs = tuple(vv.intervals.keys())
            keys = key_cache.get(contigs)
            if keys is None:
                keys = key_cache[contigs] = ["%s.%s" % (c, t) for c in contigs for t in names]
                if len(set(keys)) != len(keys):
                    return False                              # (two (contig, class) pairs share a key: the plain form)
            n = len(contigs)
            o = off[l0 * K:(l0 + n) * K + 1]
            plans.append((vv, keys, out[o[0]:o[-1]], o - o[0], protos))
            l0 += n
        for vv, keys, data, o, protos in plans:
            vv._install_split(keys, data, o, protos, K)
        if len(plans) == len(self.intervals):
            bases = np.zeros(len(plans) + 1, dtype=np.int64)
            np.cumsum([len(p[2]) for p in plans], out=bases[1:])
            self._flat_cache = (out, bases, [vv._flat_cache for vv, _, _, _, _ in plans])
        return True

    def fromIsochores(self):
        for vv in self.intervals.values():
            vv.fromIsochores()

    def outputStats(self, outfile):
        """segments and bases per (track, contig) and per track (gat/Engine.pyx:2959-2981)."""
        outfile.write("section\ttrack\tcontig\tnsegments\tlength\n")
        for track, vv in self.intervals.items():
            total_length, total_segments = 0, 0
            for contig, segmentlist in vv.items():
                segments, length = len(segmentlist), segmentlist.sum()
                outfile.write("\t".join((str(self.name), track, contig, "%i" % segments, "%i" % length)) + "\n")
                total_length = (total_length + length) & 0xFFFFFFFF          # cdef Position accumulators
                total_segments = (total_segments + segments) & 0xFFFFFFFF
            outfile.write("\t".join((str(self.name), track, "total", "%i" % total_segments, "%i" % total_length)) + "\n")

    def outputOverlapStats(self, outfile, other):
        """overlap of every list with `other` (an IntervalDictionary) (gat/Engine.pyx:3152-3165)."""
        outfile.write("section\ttrack\tcontig\toverlap\tlength\tdensity\n")
        for track, vv in self.intervals.items():
            for contig, segmentlist in vv.items():
                length = segmentlist.sum()
                if length == 0:
                    continue
                overlap = segmentlist.overlapWithSegments(other[contig])
                outfile.write("\t".join((str(self.name), track, contig, "%i" % overlap, "%i" % length,
                                         "%f" % (float(overlap) / length))) + "\n")

    def save(self, outfile, prefix="", **kwargs):
        for track, vv in self.intervals.items():
            outfile.write("track name=%s%s %s\n" % (prefix, track, " ".join("%s=%s" % kv for kv in kwargs.items())))
            for contig, segmentlist in vv.items():
                for start, end in segmentlist:
                    outfile.write("%s\t%i\t%i\n" % (contig, start, end))


# ------------------------------------------------------------------------------------------------
class Sampler(object):
    pass


class SamplerAnnotator(Sampler):
    """gat/Engine.pyx:445: the default Monte-Carlo sampler, on the GPU.

    sample() places one pseudo-sample.  The reference draws from numpy's process-global legacy
    RandomState; here each call is one work unit of the per-unit stream contract
    (include/gat_mi355.h): it draws from RandomState(seed), with `seed` taken from the
    argument or, if None, from numpy.random.randint(0, 2**32) (so numpy.random.seed() still makes
    a script reproducible)."""

    kind = 0

    def __init__(self, bucket_size=1, nbuckets=100000, nunsuccessful_rounds=0):
        self.bucket_size = bucket_size
        self.nbuckets = nbuckets
        self.nunsuccessful_rounds = nunsuccessful_rounds

    def sample(self, segments, workspace, seed=None):
        assert segments.isNormalized, "segment list is not normalized"
        assert workspace.isNormalized, "workspace is not normalized"
        if seed is None:
            seed = int(np.random.randint(0, 2 ** 32))
        if len(segments) == 0 or len(workspace) == 0 or len(iv.filter(segments.asArray(), workspace.asArray())) == 0:
            return SegmentList()
        s, w = segments.asArray(), workspace.asArray()
        flat = dict(n_units=1, segs=s, seg_off=[0, len(s)], ws=w, ws_off=[0, len(w)], unit_contig=[0], n_contigs=1,
                    merge_contigs=0, n_tracks=0, annos=iv.EMPTY, anno_off=[0], cws_nseg=[len(w)],
                    bucket_size=self.bucket_size, nbuckets=self.nbuckets)
        P = _lib.Problem(get_context(), flat)
        try:
            seg, _ = P.sample(seed, 0, 1)
            self.nunsuccessful_rounds = P.last_stats["n_unsuccessful"]
        finally:
            P.close()
        r = SegmentList(array=seg)
        r.isNormalized = 1
        return r


class SamplerSegments(Sampler):
    """gat/Engine.pyx:653: places len(segments) segments drawn from the length distribution; sampled
    segments may overlap and the list is returned in placement order (not normalized), exactly as the
    reference does.  Same per-unit stream convention as SamplerAnnotator.sample."""

    kind = 1

    def __init__(self, bucket_size=1, nbuckets=100000):
        self.bucket_size = bucket_size
        self.nbuckets = nbuckets

    def sample(self, segments, workspace, seed=None):
        assert workspace.isNormalized, "workspace is not normalized"
        if seed is None:
            seed = int(np.random.randint(0, 2 ** 32))
        if len(segments) == 0 or len(workspace) == 0 or len(iv.filter(segments.asArray(), workspace.asArray())) == 0:
            return SegmentList()
        s, w = segments.asArray(), workspace.asArray()
        flat = dict(n_units=1, segs=s, seg_off=[0, len(s)], ws=w, ws_off=[0, len(w)], unit_contig=[0], n_contigs=1,
                    merge_contigs=0, n_tracks=0, annos=iv.EMPTY, anno_off=[0], cws_nseg=[len(w)],
                    bucket_size=self.bucket_size, nbuckets=self.nbuckets, sampler=1)
        P = _lib.Problem(get_context(), flat)
        try:
            seg, _ = P.sample(seed, 0, 1)
        finally:
            P.close()
        return SegmentList(array=seg)


class Counter(object):
    name = None

    def __call__(self, segments, annotations, workspace=None):
        n = len(workspace) if workspace is not None else 1
        v = _count_pair(self.name, segments, annotations, n)
        return float(v) if self.name == "nucleotide-density" else int(v)


class CounterNucleotideOverlap(Counter):
    name = "nucleotide-overlap"           # gat/Engine.pyx:1417


class CounterNucleotideDensity(Counter):
    name = "nucleotide-density"           # gat/Engine.pyx:1428


class CounterSegmentOverlap(Counter):
    name = "segment-overlap"              # gat/Engine.pyx:1443


class CounterSegmentMidpointOverlap(Counter):
    name = "segment-midoverlap"           # gat/Engine.pyx:1450


class CounterAnnotationOverlap(Counter):
    name = "annotation-overlap"           # gat/Engine.pyx:1458


class CounterAnnotationMidpointOverlap(Counter):
    name = "annotation-midoverlap"        # gat/Engine.pyx:1465


class UnconditionalWorkspace(object):
    """gat/Engine.pyx:2061."""
    is_conditional = False

    def __call__(self, segments, annotations, workspace):
        return segments, annotations, workspace

    def filter(self, segments, annotations, workspace):  # noqa: A003
        """restrict annotations and segments to a workspace (gat/Engine.pyx:2071-2091)."""
        temp_annotations = None
        if annotations:
            temp_annotations = annotations.clone()
            temp_annotations.filter(workspace)
        temp_segments = None
        if segments:
            temp_segments = segments.clone()
            temp_segments.filter(workspace)
        return temp_segments, temp_annotations, workspace


class ConditionalWorkspaceCooccurance(UnconditionalWorkspace):
    """only the workspace segments that hold both a segment and an annotation
    (gat/Engine.pyx:2093-2109)."""
    is_conditional = True

    def __call__(self, segments, annotations, workspace):
        temp_workspace = workspace.clone()
        temp_workspace.filter(annotations)
        temp_workspace.filter(segments)
        return self.filter(segments, annotations, temp_workspace)


class ConditionalWorkspaceCentered(UnconditionalWorkspace):
    """a workspace centered on segments or annotations (gat/Engine.pyx:2111-2134)."""
    is_conditional = True

    def __init__(self, extension=None, expansion=None):
        self.extension = extension
        self.expansion = expansion
        if self.extension is None and self.expansion is None:
            raise ValueError("need to specify either expansion or extension")

    def __call__(self, segments, annotations, workspace):
        temp_workspace = self.getCenter(segments, annotations).clone()
        if self.extension is not None:
            temp_workspace.extend(self.extension)
        else:
            temp_workspace.expand(self.expansion)
        temp_workspace.normalize()
        temp_workspace.intersect(workspace)
        return self.filter(segments, annotations, temp_workspace)


class ConditionalWorkspaceAnnotationCentered(ConditionalWorkspaceCentered):
    """gat/Engine.pyx:2136-2143."""

    def getCenter(self, segments, annotations):
        return annotations


class ConditionalWorkspaceSegmentCentered(ConditionalWorkspaceCentered):
    """gat/Engine.pyx:2145-2153: computed once per segment track, so not 'conditional'."""
    is_conditional = False

    def getCenter(self, segments, annotations):
        return segments


def computeCounts(counter, aggregator, segments, annotations, workspace, workspace_generator, append=False):
    """observed counts for all track x annotation pairs (gat/Engine.pyx:2164-2204): ONE device call over all
    (segment track, isochore) x (annotation, isochore) lists."""
    return computeCountsAll([counter], aggregator, segments, annotations, workspace, workspace_generator)[0]


def _lookup_every_key(segments, annotations, workspace):
    """the reference's computeCounts looks every (track, isochore) up (segments[track][isochore], gat/Engine.pyx:2196-2200):
    a dictionary that lacks the key GAINS an empty list -- and with it a work unit (and a place in the numbering of the
    unit streams) in the sampling that follows.  run() does this before it flattens anything, whenever the counts
    themselves are taken."""
    isochores = list(workspace.keys())
    for coll in (segments, annotations):
        for t in list(coll.tracks):
            d = coll[t]
            if list(d.keys()) != isochores:
                for i in isochores:
                    d[i]


def computeCountsAll(counters, aggregator, segments, annotations, workspace, workspace_generator=None, _aflat=None):
    """computeCounts for several counters at once: the lists cross to the device once, every counter is evaluated by
    the same launch (gat_count_list_ranges).  Returns one {track: {annotation: count}} per counter.
    _aflat: annotations._flat(tracks) from a caller that has it and has not touched the collection since."""
    if aggregator is not sum:
        raise NotImplementedError("only aggregator=sum is supported")
    all_counts = [collections.defaultdict(lambda: collections.defaultdict(float)) for _ in counters]
    isochores = list(workspace.keys())
    ctx = get_context()
    tracks = list(annotations.tracks)
    seg_tracks = list(segments.tracks)
    if not seg_tracks or not tracks or not counters:
        return all_counts
    if annotations.hasPositions() and any(c.name not in POINT_COUNTERS for c in counters):
        raise _points_type_error("annotations")              # gat/Engine.pyx:2200: counter(SegmentList, PositionList, ..)
    _lookup_every_key(segments, annotations, workspace)
    sdata, sbases, sflats = segments._flat(seg_tracks)
    lb, le = zip(*[f.ranges(isochores, base) for f, base in zip(sflats, sbases)])
    lb, le = np.concatenate(lb), np.concatenate(le)
    if len(lb) and bool(np.all(lb[1:] == le[:-1])) and lb[0] == 0:
        lcat, loff = sdata, np.append(lb, le[-1])
    else:
        lcat, loff = _problem._cat([sdata[b:e] for b, e in zip(lb.tolist(), le.tolist())])
    af = annotations._flat(tracks, _have=_aflat)
    ab, ae = annotations._ranges(af, isochores)
    ws_nseg = [len(workspace[i]) for i in isochores]
    names = [c.name for c in counters]
    r = ctx.count_lists(names, lcat, loff, len(seg_tracks), af[0], ab, len(tracks), ws_nseg, len(isochores), anno_end=ae)
    for k, name in enumerate(names):
        vals = r[k].tolist()                                 # [annotation][segment track]: Python floats / ints
        counts = all_counts[k]
        for l, track in enumerate(seg_tracks):
            per = counts[track]
            for a_i, annotation in enumerate(tracks):
                per[annotation] = vals[a_i][l]
    return all_counts


def overlap_sizes(track_segments, annotations, tracks=None, _aflat=None):
    """{annotation track: (segments, bases)} of `track_segments.clone().intersect(annotations[track])` for every track at
    once (gat_intersection_sizes): the overlap_* columns of the result rows (gat/Engine.pyx:1911-1928).  None when a list
    is not a normalized SegmentList (the row then takes the list-by-list path, which raises as the reference does)."""
    tracks = list(annotations.tracks) if tracks is None else list(tracks)
    if not tracks or track_segments._has_points() or not track_segments._all_normalized():
        return None
    plain = annotations._plain_memo                    # (run(): every list of the collection looked at once per call)
    if plain is None:
        plain = all(not annotations[t]._has_points() and annotations[t]._all_normalized() for t in tracks)
    if not plain or any(annotations[t] is track_segments for t in tracks):
        return None
    fs = track_segments._flat()
    af = annotations._flat(tracks, _have=_aflat)
    bb, be = annotations._ranges(af, fs.keys)
    pairs, bases = _lib.intersection_sizes(fs.data, fs.off, af[0], bb, be, len(tracks))
    return dict((t, (int(p), int(b))) for t, p, b in zip(tracks, pairs.tolist(), bases.tolist()))


# ------------------------------------------------------------------------------------------------
def getTwoSidedPValue(sorted_samples, expected, val):
    """gat/Engine.pyx:1543-1576 on the sorted sample values."""
    l = len(sorted_samples)  # noqa: E741
    idx = int(np.searchsorted(sorted_samples, val, side="left"))
    min_pval = 1.0 / l
    if idx == l:
        idx = 1
    elif val > expected:
        while idx > 0 and sorted_samples[idx] == val:
            idx -= 1
        idx = l - (idx + 1)
    else:
        while idx < l and sorted_samples[idx] == val:
            idx += 1
    return max(min_pval, float(idx) / l)


class AnnotatorResult(object):
    """gat/Engine.pyx:1725 / makeEnrichmentStatistics :1635-1718 (numpy on the host; identical
    IEEE arithmetic: numpy.mean/std, sorted-value lookups)."""

    format_observed = "%i"
    format_expected = "%6.4f"
    format_fold = "%6.4f"
    format_pvalue = "%6.4e"
    format_counts = "%i"
    format_density = "%6.4e"
    headers = ["track", "annotation", "observed", "expected", "CI95low", "CI95high", "stddev", "fold", "l2fold",
               "pvalue", "qvalue"]

    def __init__(self, track, annotation, counter, observed, samples, reference=None, pseudo_count=1.0, _stats=None):
        """_stats: (mean, std, value at the lower / upper interval position, samples < val, samples == val) of `samples`
        as gat_null_stats computed them on the device (val = observed, or observed / reference.fold); None = numpy here."""
        self.track, self.annotation, self.counter = track, annotation, counter
        # the reference builds a Python list of floats and sorts it; the same numbers come out of array operations
        # without the sort (two order statistics by selection, the p-value from two counts), which is what keeps
        # 1000 tracks x 100 000 samples from spending longer here than on the GPU
        # (with the statistics given the row is only kept; the float copy the reference holds is made when somebody asks)
        self._raw = samples if (_stats is not None and (isinstance(samples, np.ndarray) or hasattr(samples, "__array__"))) else None
        self._samples_cache = None if self._raw is not None else np.array(samples, dtype=np.float64)
        l = len(samples)  # noqa: E741
        if l < 1:
            raise ValueError("no samples")
        self.observed = float(observed)
        self.nsamples = l
        self._sorted_cache = None
        self._val_counts = None
        self.expected = float(_stats[0]) if _stats is not None else float(np.mean(self._samples))
        if reference is not None:
            self.expected *= reference.fold
        if self.expected != 0:
            self.fold = (self.observed + pseudo_count) / (self.expected + pseudo_count)
        else:
            self.fold = 1.0
        if _stats is not None:
            self.stddev, self.lower95, self.upper95 = float(_stats[1]), float(_stats[2]), float(_stats[3])
            self._val_counts = (self.observed if reference is None else self.observed / reference.fold if reference.fold > 0 else None,
                                int(_stats[4]), int(_stats[5]))
        else:
            self.stddev = float(np.std(self._samples))
            offset = int(0.05 * l)
            lo_i, hi_i = (min(offset, l - 1), max(l - offset, 0)) if offset > 0 else (0, l - 1)
            part = np.partition(self._samples, sorted(set((lo_i, hi_i))))
            self.lower95 = float(part[lo_i])
            self.upper95 = float(part[hi_i])
        if reference is None:
            self.pvalue = self._two_sided(self.observed)
        else:
            if reference.fold > 0:
                self.pvalue = self._two_sided(self.observed / reference.fold)
            else:
                raise ValueError("0 fold change not applicable")
            self.lower95 *= reference.fold
            self.upper95 *= reference.fold
        self.qvalue = 1.0

    def _two_sided(self, val):
        """getTwoSidedPValue (gat/Engine.pyx:1543-1576) from counts instead of a walk over the sorted values: with
        n_less values below val and n_eq equal to it, searchsorted gives n_less and the tie loops move to the other
        side of the run of equal values (downwards only while the index stays positive)."""
        l = self.nsamples  # noqa: E741
        if self._val_counts is not None and self._val_counts[0] == val:      # counted on the device for this very value
            n_less, n_eq = self._val_counts[1], self._val_counts[2]
        else:
            n_less = int(np.count_nonzero(self._samples < val))
            n_eq = int(np.count_nonzero(self._samples == val))
        idx = n_less
        if idx == l:
            idx = 1
        elif val > self.expected:
            if n_eq > 0 and idx > 0:
                idx -= 1
            idx = l - (idx + 1)
        else:
            idx += n_eq
        return max(1.0 / l, float(idx) / l)

    @property
    def _samples(self):
        if self._samples_cache is None:
            self._samples_cache = np.array(self._raw, dtype=np.float64)
            self._raw = None
        return self._samples_cache

    @property
    def _sorted(self):
        if self._sorted_cache is None:
            self._sorted_cache = np.sort(self._samples)
        return self._sorted_cache

    @property
    def samples(self):
        return self._samples.copy()

    def getSample(self, sample_id):
        return float(self._samples[sample_id])

    def getEmpiricalPValue(self, value):
        return self._two_sided(value)

    def _base_columns(self):
        logfold = self.format_fold % math.log(self.fold, 2) if self.fold > 0 else "-inf"
        return (self.track, self.annotation, self.format_observed % self.observed,
                self.format_expected % self.expected, self.format_expected % self.lower95,
                self.format_expected % self.upper95, self.format_expected % self.stddev,
                self.format_fold % self.fold, logfold, self.format_pvalue % self.pvalue,
                self.format_pvalue % self.qvalue)

    def __str__(self):
        return "\t".join(self._base_columns())


class AnnotatorResultExtended(AnnotatorResult):
    """gat/Engine.pyx:1854-1974: adds sizes/densities of track, annotation and overlap."""

    headers = AnnotatorResult.headers + [
        "track_nsegments", "track_size", "track_density", "annotation_nsegments", "annotation_size",
        "annotation_density", "overlap_nsegments", "overlap_size", "overlap_density",
        "percent_overlap_nsegments_track", "percent_overlap_size_track",
        "percent_overlap_nsegments_annotation", "percent_overlap_size_annotation"]

    def __init__(self, track, annotation, counter, observed, samples, track_segments, annotation_segments,
                 workspace, reference=None, pseudo_count=1.0, _sizes=None, _stats=None, _overlap=None):
        """_overlap: (segments, bases) of the intersection of the two dictionaries when the caller has them already
        (overlap_sizes: every annotation of a run in one call)."""
        AnnotatorResult.__init__(self, track, annotation, counter, observed, samples, reference=reference,
                                 pseudo_count=pseudo_count, _stats=_stats)
        sizes = _sizes if _sizes is not None else {}

        def cached(obj):                               # (counts, sum) of a dictionary, once per object and run()
            key = id(obj)
            if key not in sizes:
                sizes[key] = (obj.counts(), obj.sum())
            return sizes[key]
        self.track_nsegments, self.track_size = cached(track_segments)
        self.annotation_nsegments, self.annotation_size = cached(annotation_segments)
        stats = _overlap
        if stats is None and hasattr(track_segments, "intersect_stats"):
            stats = track_segments.intersect_stats(annotation_segments, _cache=sizes)
        if stats is None:
            overlap = track_segments.clone()
            try:
                overlap.intersect(annotation_segments)
            except TypeError:
                pass       # SegmentList x PositionList: "needs still to be implemented" (gat/Engine.pyx:1917-1923)
            stats = (overlap.counts(), overlap.sum())
        self.overlap_nsegments, self.overlap_size = stats
        self.workspace_size = cached(workspace)[1]

    def __str__(self):
        def _toFold(a, b):
            return self.format_fold % (100.0 * float(a) / b) if b > 0 else "na"

        def _toDensity(a, b):
            return self.format_density % (100.0 * float(a) / b) if b > 0 else "na"

        return "\t".join(self._base_columns() + (
            self.format_counts % self.track_nsegments, self.format_counts % self.track_size,
            _toDensity(self.track_size, self.workspace_size),
            self.format_counts % self.annotation_nsegments, self.format_counts % self.annotation_size,
            _toDensity(self.annotation_size, self.workspace_size),
            self.format_counts % self.overlap_nsegments, self.format_counts % self.overlap_size,
            _toDensity(self.overlap_size, self.workspace_size),
            _toFold(self.overlap_nsegments, self.track_nsegments), _toFold(self.overlap_size, self.track_size),
            _toFold(self.overlap_nsegments, self.annotation_nsegments), _toFold(self.overlap_size, self.annotation_size)))


def getNormedPValue(value, r):
    """p-value under a Gaussian fitted to the null distribution (gat/Engine.pyx:1979-1990)."""
    absval = abs(value - r.expected)
    if r.stddev == 0:
        return 1.0
    import scipy.stats
    return 1.0 - scipy.stats.norm.cdf(absval, 0, r.stddev)


def getEmpiricalPValue(value, r):
    """gat/Engine.pyx:1995."""
    return r.getEmpiricalPValue(value)


def updatePValues(annotator_results, method="empirical"):
    """gat/Engine.pyx:2001-2020."""
    if method == "norm":
        methodf = getNormedPValue
    elif method == "empirical":
        methodf = getEmpiricalPValue
    else:
        raise ValueError("unknown method '%s'" % method)
    for r in annotator_results:
        r.pvalue = methodf(r.observed, r)


def getQValues(pvalues, method="storey", **kwargs):
    """gat/Engine.pyx:2025-2040."""
    from . import stats
    return stats.getQValues(pvalues, method=method, **kwargs)


def updateQValues(annotator_results, method="storey", **kwargs):
    """gat/Engine.pyx:2044-2054."""
    pvalues = [r.pvalue for r in annotator_results]
    for r, qvalue in zip(annotator_results, getQValues(pvalues, method, **kwargs)):
        r.qvalue = qvalue

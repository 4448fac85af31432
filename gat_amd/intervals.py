"""Host-side interval algebra on packed (start, end) uint32 arrays (numpy, vectorised).

Used only for one-time input preparation (what gat/IO.py:88-293 does after parsing: normalize,
filter/intersect with the workspace, isochore split).  The per-sample work of the hot path
(placement, sort/merge, fromIsochores, counting) never runs here; it runs in the HIP kernels.
Semantics follow gat/SegmentList.pyx; coordinates must be < 2^31 (see include/gat_mi355.h).
"""
import numpy as np

SEG = np.dtype([("start", "<u4"), ("end", "<u4")])
EMPTY = np.empty(0, dtype=SEG)


def as_segments(x):
    """any iterable of (start, end) pairs or SEG array -> SEG array (copy)."""
    if isinstance(x, np.ndarray) and x.dtype == SEG:
        return x.copy()
    a = np.asarray(list(x), dtype=np.int64).reshape(-1, 2)
    if len(a) and (a.min() < 0 or a.max() >= (1 << 32)):
        raise OverflowError("segment coordinate out of range for unsigned int")
    out = np.empty(len(a), dtype=SEG)
    out["start"] = a[:, 0]
    out["end"] = a[:, 1]
    return out


def make(start, end):
    out = np.empty(len(start), dtype=SEG)
    out["start"] = start
    out["end"] = end
    return out


def _merge_sorted(start, end, distance, adjacent_rule):
    """shared body of normalize (start >= max_end starts a new segment, gat/SegmentList.pyx:736)
    and merge(distance) (start - distance > max_end, :801)."""
    keep = end != start
    start, end = start[keep].astype(np.int64), end[keep].astype(np.int64)
    if len(start) == 0:
        return EMPTY.copy()
    order = np.argsort(start, kind="stable")
    start, end = start[order], end[order]
    run = np.maximum.accumulate(end)
    head = np.ones(len(start), dtype=bool)
    if adjacent_rule == "normalize":
        head[1:] = start[1:] >= run[:-1]
    else:
        head[1:] = (start[1:] - distance) > run[:-1]
    idx = np.flatnonzero(head)
    last = np.append(idx[1:] - 1, len(start) - 1)
    return make(start[idx], run[last])


def normalize(a):
    """SegmentList.normalize (gat/SegmentList.pyx:697-754): merge overlapping, keep adjacent apart."""
    return _merge_sorted(a["start"], a["end"], 0, "normalize")


def merge(a, distance=0):
    """SegmentList.merge(distance) (gat/SegmentList.pyx:756-816)."""
    return _merge_sorted(a["start"], a["end"], int(distance), "merge")


def is_normalized(a):
    """SegmentList.check (gat/SegmentList.pyx:818-851)."""
    if len(a) == 0:
        return True
    if np.any(a["start"] >= a["end"]):
        return False
    return bool(np.all(a["end"][:-1] <= a["start"][1:]))


def total(a):
    """SegmentList.sum (gat/SegmentList.pyx:1607): uint32 accumulate."""
    return int((a["end"].astype(np.int64) - a["start"].astype(np.int64)).sum() & 0xFFFFFFFF)


def _overlap_ranges(a, b):
    """for each segment of a: [j0, j1) = indices of the segments of normalized b overlapping it."""
    j0 = np.searchsorted(b["end"], a["start"], side="right")
    j1 = np.searchsorted(b["start"], a["end"], side="left")
    return j0, np.maximum(j1, j0)


def filter(a, b):  # noqa: A001 - mirrors SegmentList.filter
    """SegmentList.filter (gat/SegmentList.pyx:1401-1467): whole segments of a touching b."""
    if len(a) == 0 or len(b) == 0:
        return EMPTY.copy()
    j0, j1 = _overlap_ranges(a, b)
    return a[j1 > j0].copy()


def intersect(a, b):
    """SegmentList.intersect (gat/SegmentList.pyx:1469-1549): one piece per overlapping pair."""
    if len(a) == 0 or len(b) == 0:
        return EMPTY.copy()
    j0, j1 = _overlap_ranges(a, b)
    cnt = j1 - j0
    n = int(cnt.sum())
    if n == 0:
        return EMPTY.copy()
    ai = np.repeat(np.arange(len(a)), cnt)
    first = np.repeat(np.cumsum(cnt) - cnt, cnt)
    bi = np.repeat(j0, cnt) + (np.arange(n) - first)
    return make(np.maximum(a["start"][ai], b["start"][bi]), np.minimum(a["end"][ai], b["end"][bi]))


def overlap(a, b):
    """bases shared by two normalized lists (overlapWithSegments, gat/SegmentList.pyx:1026) -- host
    convenience for input statistics only."""
    return total(intersect(a, b))

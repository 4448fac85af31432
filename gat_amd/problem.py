"""Flatten host-side interval dictionaries into the CSR arrays of gat_problem_desc
(include/gat_mi355.h), walking them exactly as gat.computeSample does
(gat/__init__.py:494-591): unit order = list(segs.keys()), units whose workspace or segment
list is empty are skipped, contig order = first appearance among the units that are not.
"""
import collections

import numpy as np

from . import intervals as iv

SEG = iv.SEG


def _cat(lists):
    lists = [x for x in lists]
    if not lists:
        return iv.EMPTY.copy(), np.zeros(1, dtype=np.int64)
    off = np.concatenate([[0], np.cumsum([len(x) for x in lists])]).astype(np.int64)
    return (np.concatenate(lists) if off[-1] else iv.EMPTY.copy()), off


def split_key(key):
    """contig of an isochore key, as IntervalDictionary.fromIsochores does (gat/Engine.pyx:2862-2868)."""
    k = key.strip()
    if "." in k and k != ".":
        parts = k.split(".")
        if len(parts) != 2:
            raise ValueError("isochore key %r: expected exactly one '.' (gat/Engine.pyx:2864)" % key)
        return parts[0], True
    return k, False


def from_isochores(d):
    """IntervalDictionary.fromIsochores (gat/Engine.pyx:2857-2876) on a dict key -> SEG array."""
    parts = collections.OrderedDict()
    merged = False
    for key, a in d.items():
        contig, dotted = split_key(key)
        if dotted:
            parts.setdefault(contig, []).append(a)
            merged = True
        else:
            parts[contig] = [a]                      # new[isochore] = segmentlist (replaces)
    new = collections.OrderedDict()
    for contig, lst in parts.items():
        a = lst[0] if len(lst) == 1 else np.concatenate(lst)
        new[contig] = iv.merge(a, 0) if merged else a
    return new


def contig_list_lengths(dictionary):
    """{contig: len(dictionary.fromIsochores()[contig])} (IntervalDictionary.fromIsochores, gat/Engine.pyx:2857-2876: the
    lists of a contig's isochore keys concatenated, sorted and merge(0)d -- empty segments dropped, a segment starting at or
    before the running end joins it, gat/SegmentList.pyx:756-816) from the dictionary's flat form, all contigs in one pass:
    only the LENGTHS are wanted (len(workspace[contig]): the density counter's divisor, gat/Engine.pyx:1437)."""
    f = dictionary._flat()
    contigs, index, cid, dotted_any, plain_any = [], {}, [], False, False
    for k in f.keys:
        contig, dotted = split_key(k)
        dotted_any |= dotted
        plain_any |= not dotted
        if contig not in index:
            index[contig] = len(contigs)
            contigs.append(contig)
        cid.append(index[contig])
    if dotted_any and plain_any:
        # keys with and without a dot in one dictionary: a plain key REPLACES what its contig has gathered so far
        # (new[isochore] = segmentlist, gat/Engine.pyx:2857-2876) -- rare enough for the list-by-list form
        return collections.OrderedDict((c, len(a)) for c, a in from_isochores(dictionary.asArrays()).items())
    if not dotted_any:
        # a key IS its contig (a later list of the same contig replaces an earlier one: dictionary assignment)
        out = collections.OrderedDict()
        for k, c, n in zip(f.keys, cid, np.diff(f.off).tolist()):
            out[contigs[c]] = n
        return out
    counts = np.zeros(len(contigs), dtype=np.int64)
    if len(f.data):
        hi = np.repeat(np.asarray(cid, dtype=np.int64) << 32, np.diff(f.off))
        start, end = f.data["start"].astype(np.int64), f.data["end"].astype(np.int64)
        keep = end > start                               # merge() drops empty segments
        ks, ke = (hi + start)[keep], (hi + end)[keep]
        order = np.argsort(ks, kind="stable")
        ks, ke = ks[order], ke[order]
        if len(ks):
            run = np.maximum.accumulate(ke)
            head = np.ones(len(ks), dtype=bool)
            head[1:] = ks[1:] > run[:-1]                 # (another contig's keys lie above every end of this one)
            counts = np.bincount((ks[head] >> 32), minlength=len(contigs)).astype(np.int64)
    return collections.OrderedDict((c, int(n)) for c, n in zip(contigs, counts.tolist()))


def to_isochores(d, isochores, truncate):
    """IntervalDictionary.toIsochores (gat/Engine.pyx:2837-2855)."""
    out = collections.OrderedDict()
    for contig, a in d.items():
        for track, per in isochores.items():
            other = per.get(contig, iv.EMPTY)
            out["%s.%s" % (contig, track)] = iv.intersect(a, other) if truncate else iv.filter(a, other)
    return out


def flatten_units(segs, workspace, annotations, bucket_size=0, nbuckets=100000, count_workspace=None):
    """segs / workspace: OrderedDict unit key -> SEG array (isochore level);
    annotations: list of (track, OrderedDict unit key -> SEG array) (isochore level);
    count_workspace: the workspace the counters see (gat/__init__.py:720 contig_workspace) when a
    workspace generator made the sampling workspace a different one."""
    units = list(segs.keys())
    contig_annotations = [(t, from_isochores(per)) for t, per in annotations]
    contig_workspace = from_isochores(workspace if count_workspace is None else count_workspace)
    seg_arrays, ws_arrays, unit_contig, contigs = [], [], [], []
    contig_index = {}
    merge = 0
    dotted_any, plain_any = False, False
    for u in units:
        sa = segs[u]
        wa = workspace.get(u, iv.EMPTY)
        seg_arrays.append(sa)
        ws_arrays.append(wa)
        contig, dotted = split_key(u)
        dotted_any |= dotted
        plain_any |= not dotted
        if dotted:
            merge = 1
        if len(sa) == 0 or len(wa) == 0:          # gat/__init__.py:536-538
            unit_contig.append(-1)
            continue
        if contig not in contig_index:
            contig_index[contig] = len(contigs)
            contigs.append(contig)
        unit_contig.append(contig_index[contig])
    if dotted_any and plain_any:
        raise ValueError("mixing keys with and without isochores is not supported")
    anno_arrays = []
    for _, per in contig_annotations:
        for c in contigs:
            anno_arrays.append(per.get(c, iv.EMPTY))
    segs_cat, seg_off = _cat(seg_arrays)
    ws_cat, ws_off = _cat(ws_arrays)
    annos_cat, anno_off = _cat(anno_arrays)
    return dict(n_units=len(units), unit_names=list(units), segs=segs_cat, seg_off=seg_off, ws=ws_cat, ws_off=ws_off,
                unit_contig=np.array(unit_contig, dtype=np.int32), n_contigs=len(contigs), contig_names=list(contigs),
                merge_contigs=merge, n_tracks=len(annotations), track_names=[t for t, _ in annotations],
                annos=annos_cat, anno_off=anno_off,
                cws_nseg=np.array([len(contig_workspace.get(c, iv.EMPTY)) for c in contigs], dtype=np.int64),
                bucket_size=int(bucket_size), nbuckets=int(nbuckets))


def flatten_dictionaries(segs, workspace, annotations, tracks, bucket_size=0, nbuckets=100000, count_workspace=None, _aflat=None,
                         _shared=None):
    """flatten_units from the host classes themselves (IntervalDictionary / IntervalCollection, gat_amd/engine.py) without a
    numpy call per list: segments and workspace per unit from the dictionaries' flat forms, and the annotation lists
    handed over as they are -- one per (track, key), each with the contig it belongs to -- for the library to form the
    contig-level lists (gat_problem_desc::anno_group; IntervalDictionary.fromIsochores on its host threads).  Same result as
    flatten_units(segs.asArrays(), workspace.asArrays(), [(t, annotations[t].asArrays()) ...]); None when the shortcut
    does not apply (a track mixing keys with and without isochores).
    _shared: {annotations_key: device annotation object} of the run -- when the key of this problem (its contigs in order,
    whether fromIsochores merges) is in there the annotation lists are not walked again (flat["annos"] is None)."""
    fs, fw = segs._flat(), workspace._flat()
    units = fs.keys
    seg_len = np.diff(fs.off)
    wb, we = fw.ranges(units)
    ws_len = we - wb
    unit_contig, contigs, contig_index = [], [], {}
    dotted_any = plain_any = False
    for u, ns, nw in zip(units, seg_len.tolist(), ws_len.tolist()):
        contig, dotted = split_key(u)
        dotted_any |= dotted
        plain_any |= not dotted
        if ns == 0 or nw == 0:                      # gat/__init__.py:536-538
            unit_contig.append(-1)
            continue
        if contig not in contig_index:
            contig_index[contig] = len(contigs)
            contigs.append(contig)
        unit_contig.append(contig_index[contig])
    if dotted_any and plain_any:
        raise ValueError("mixing keys with and without isochores is not supported")
    merge = 1 if dotted_any else 0
    if fw.keys == units:
        ws_cat, ws_off = fw.data, fw.off
    else:
        ws_cat, ws_off = _cat([fw.data[b:e] for b, e in zip(wb.tolist(), we.tolist())])
    n_contigs = len(contigs)
    akey = (tuple(contigs), merge)
    if _shared is not None and akey in _shared:
        cws = contig_list_lengths(workspace if count_workspace is None else count_workspace)
        return dict(n_units=len(units), unit_names=list(units), segs=fs.data, seg_off=fs.off, ws=ws_cat, ws_off=ws_off,
                    unit_contig=np.array(unit_contig, dtype=np.int32), n_contigs=n_contigs, contig_names=list(contigs),
                    merge_contigs=merge, n_tracks=len(tracks), track_names=list(tracks),
                    annos=None, anno_off=None, anno_end=None, anno_group=None, annotations_key=akey,
                    cws_nseg=np.array([cws.get(c, 0) for c in contigs], dtype=np.int64),
                    bucket_size=int(bucket_size), nbuckets=int(nbuckets))
    aflat3 = annotations._flat(tracks, _have=_aflat)
    adata, abases, aflats = aflat3
    groups, begins, ends = [], [], []
    last_keys, last_group = None, None
    same_keys = bool(aflats) and all(f.keys is aflats[0].keys or f.keys == aflats[0].keys for f in aflats)
    if same_keys and hasattr(annotations, "_ranges"):
        # every track holds the same keys: one group table, the ranges from the collection (kept over a run())
        f = aflats[0]
        g, dotted_t = [], False
        for k in f.keys:
            contig, dotted = split_key(k)
            dotted_t |= dotted
            g.append(contig_index.get(contig, -1))
        if f.keys and dotted_t != bool(merge) and len(adata):
            return None
        g = np.array(g, dtype=np.int64)
        gt = np.where(g >= 0, g[None, :] + (np.arange(len(aflats), dtype=np.int64) * n_contigs)[:, None], -1).ravel()
        rb, re_ = annotations._ranges(aflat3, f.keys)
        groups, begins, ends, aflats = [gt], [rb], [re_], []
    for t, (f, base) in enumerate(zip(aflats, abases.tolist())):
        if f.keys is not last_keys and f.keys != last_keys:
            g, dotted_t = [], False
            for k in f.keys:
                contig, dotted = split_key(k)
                dotted_t |= dotted
                g.append(contig_index.get(contig, -1))
            if f.keys and dotted_t != bool(merge) and len(f.data):
                return None                          # this track's fromIsochores would not do what the segments' does
            last_keys, last_group = f.keys, np.array(g, dtype=np.int64)
        groups.append(np.where(last_group >= 0, last_group + t * n_contigs, -1))
        begins.append(f.off[:-1] + base)
        ends.append(f.off[1:] + base)
    cws = contig_list_lengths(workspace if count_workspace is None else count_workspace)
    cat = (lambda xs, dt: np.concatenate(xs).astype(dt, copy=False) if xs else np.zeros(0, dtype=dt))
    return dict(n_units=len(units), unit_names=list(units), segs=fs.data, seg_off=fs.off, ws=ws_cat, ws_off=ws_off,
                unit_contig=np.array(unit_contig, dtype=np.int32), n_contigs=n_contigs, contig_names=list(contigs),
                merge_contigs=merge, n_tracks=len(tracks), track_names=list(tracks),
                annos=adata, anno_off=cat(begins, np.int64), anno_end=cat(ends, np.int64), anno_group=cat(groups, np.int32),
                annotations_key=akey,
                cws_nseg=np.array([cws.get(c, 0) for c in contigs], dtype=np.int64),
                bucket_size=int(bucket_size), nbuckets=int(nbuckets))


def apply_isochores(segments, annotations, workspace, isochores=None, truncate_segments=False):
    """the array form of IO.applyIsochores (gat/IO.py:188-248).

    segments / workspace: OrderedDict contig -> normalized SEG array; annotations: list of
    (track, OrderedDict contig -> SEG array); isochores: OrderedDict track -> OrderedDict contig ->
    SEG array, or None.  Returns (segs, annotations, workspace) at isochore level."""
    if isochores:
        iso = collections.OrderedDict()
        for track, per in isochores.items():        # isochores.intersect(workspace), gat/IO.py:176
            iso[track] = collections.OrderedDict((c, iv.intersect(a, workspace[c])) for c, a in per.items() if c in workspace)
        ws = to_isochores(workspace, iso, True)
        annos = [(t, to_isochores(per, iso, True)) for t, per in annotations]
        segs = to_isochores(segments, iso, truncate_segments)
        return segs, annos, ws
    segs = collections.OrderedDict()
    for c, a in segments.items():                    # segments.filter / intersect (gat/IO.py:243-246)
        if c in workspace:
            segs[c] = iv.intersect(a, workspace[c]) if truncate_segments else iv.filter(a, workspace[c])
    annos = []
    for t, per in annotations:                       # annotations.intersect (gat/IO.py:248)
        annos.append((t, collections.OrderedDict((c, iv.intersect(a, workspace[c])) for c, a in per.items() if c in workspace)))
    return segs, annos, workspace


def flatten_arrays(segments, annotations, workspace, isochores=None, bucket_size=0, nbuckets=100000,
                   truncate_segments=False):
    segs, annos, ws = apply_isochores(segments, annotations, workspace, isochores, truncate_segments)
    return flatten_units(segs, ws, annos, bucket_size, nbuckets)

"""ctypes binding of libgat_mi355.so (C ABI: include/gat_mi355.h).

The library is the product's only compute path.  If it is missing, or no gfx950 device is
usable, everything here raises -- there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# (GAT_LIB_PATH: a diagnostic build of the same library, tools/diag_sampler.sh)
LIB_PATH = os.environ.get("GAT_LIB_PATH") or os.path.join(HERE, "libgat_mi355.so")

SEG = np.dtype([("start", "<u4"), ("end", "<u4")])

COUNTER_IDS = {
    "nucleotide-overlap": 0,
    "nucleotide-density": 1,
    "segment-overlap": 2,
    "segment-midoverlap": 3,
    "annotation-overlap": 4,
    "annotation-midoverlap": 5,
}

# every symbol include/gat_mi355.h declares
SYMBOLS = [
    "gat_ctx_create", "gat_ctx_destroy", "gat_last_error", "gat_version", "gat_ctx_synchronize", "gat_ctx_stream", "gat_ctx_set_kernel_times",
    "gat_dev_alloc", "gat_dev_free", "gat_memcpy_d2h", "gat_memcpy_h2d",
    "gat_problem_create", "gat_problem_destroy", "gat_sample_and_count", "gat_sample", "gat_sample_units",
    "gat_count_lists", "gat_count_list_ranges", "gat_intersection_sizes", "gat_problem_info",
    "gat_comm_unique_id", "gat_comm_create", "gat_comm_destroy", "gat_allgather_counts", "gat_null_stats",
    "gat_sample_and_count_serial", "gat_mt19937_seed", "gat_sample_and_count_enqueue", "gat_wait",
    "gat_annotations_create", "gat_annotations_destroy", "gat_annotations_wait", "gat_list_sums", "gat_problem_rng_rows",
    "gat_isochore_split", "gat_comm_library_preloaded", "gat_ctx_set_option", "gat_ctx_get_option",
]

MT_STATE_WORDS = 625          # GAT_MT_STATE_WORDS: 624 state words + numpy's position


def mt19937_seed(seed):
    """the state numpy.random.seed(seed) leaves for an integer seed (624 words + position 624)"""
    st = np.zeros(MT_STATE_WORDS, dtype=np.uint32)
    lib().gat_mt19937_seed(int(seed) & 0xFFFFFFFF, _p(st))
    return st


COUNT_KERNELS = {0: "none", 1: "k_count_seg", 2: "k_count_swap", 3: "k_count_merged"}


class GatError(RuntimeError):
    pass


class ProblemDesc(C.Structure):
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
        ("n_anno_lists", C.c_int64),
        ("anno_end", C.c_void_p),
        ("anno_group", C.c_void_p),
        ("annotations", C.c_void_p),
    ]


class AnnotationsDesc(C.Structure):
    _fields_ = [
        ("n_tracks", C.c_int32),
        ("n_contigs", C.c_int32),
        ("merge_contigs", C.c_int32),
        ("annos", C.c_void_p),
        ("anno_off", C.c_void_p),
        ("n_anno_lists", C.c_int64),
        ("anno_end", C.c_void_p),
        ("anno_group", C.c_void_p),
        ("mean_segment_length", C.c_double),
        ("flags", C.c_int32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("ms_sampler", C.c_float),
        ("ms_contig", C.c_float),
        ("ms_count", C.c_float),
        ("ms_total", C.c_float),
        ("n_placed", C.c_int64),
        ("n_draws", C.c_int64),
        ("n_sampled_segments", C.c_int64),
        ("n_unsuccessful", C.c_int64),
        ("n_retried", C.c_int64),
        ("n_full_units", C.c_int64),
        ("ms_count_main", C.c_float),
        ("ms_rng", C.c_float),
        ("ms_place", C.c_float),
        ("ms_merge", C.c_float),
        ("ms_tail", C.c_float),
        ("count_kernel", C.c_int32),
        ("ms_ktail", C.c_float),
        ("ms_finalize", C.c_float),
        ("n_tail_units", C.c_int64),
        ("lists_from_records", C.c_int64),
        ("n_index_entries", C.c_int64),
        ("n_index_lookups", C.c_int64),
        ("n_batches", C.c_int64),
        ("merged_form", C.c_int64),
        ("n_resumed_units", C.c_int64),
        ("n_straddle_candidates", C.c_int64),
        ("n_unit_overlaps", C.c_int64),
        ("kernel_times", C.c_int64),
        ("n_queued_units", C.c_int64),
    ]

    def asdict(self):
        return dict((f, getattr(self, f)) for f, _ in self._fields_)


_LIB = None


def lib():
    """load libgat_mi355.so; raises GatError if it has not been built (no fallback)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise GatError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); gat_amd has no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32
    L.gat_ctx_create.restype = C.c_int
    L.gat_ctx_create.argtypes = [C.POINTER(vp), C.c_int, vp]
    L.gat_ctx_destroy.restype = None
    L.gat_ctx_destroy.argtypes = [vp]
    L.gat_last_error.restype = C.c_char_p
    L.gat_last_error.argtypes = [vp]
    L.gat_version.restype = C.c_char_p
    L.gat_version.argtypes = []
    L.gat_ctx_synchronize.restype = C.c_int
    L.gat_ctx_synchronize.argtypes = [vp]
    L.gat_ctx_stream.restype = vp
    L.gat_ctx_stream.argtypes = [vp]
    L.gat_ctx_set_kernel_times.restype = C.c_int
    L.gat_ctx_set_kernel_times.argtypes = [vp, C.c_int]
    L.gat_dev_alloc.restype = C.c_int
    L.gat_dev_alloc.argtypes = [vp, C.POINTER(vp), C.c_size_t]
    L.gat_dev_free.restype = C.c_int
    L.gat_dev_free.argtypes = [vp, vp]
    L.gat_memcpy_d2h.restype = C.c_int
    L.gat_memcpy_d2h.argtypes = [vp, vp, vp, C.c_size_t]
    L.gat_memcpy_h2d.restype = C.c_int
    L.gat_memcpy_h2d.argtypes = [vp, vp, vp, C.c_size_t]
    L.gat_problem_create.restype = C.c_int
    L.gat_problem_create.argtypes = [vp, C.POINTER(ProblemDesc), C.POINTER(vp)]
    L.gat_problem_destroy.restype = None
    L.gat_problem_destroy.argtypes = [vp]
    L.gat_annotations_create.restype = C.c_int
    L.gat_annotations_create.argtypes = [vp, C.POINTER(AnnotationsDesc), C.POINTER(vp)]
    L.gat_annotations_wait.restype = C.c_int
    L.gat_annotations_wait.argtypes = [vp, vp]
    L.gat_annotations_destroy.restype = None
    L.gat_annotations_destroy.argtypes = [vp]
    L.gat_sample_and_count.restype = C.c_int
    L.gat_sample_and_count.argtypes = [vp, vp, vp, C.c_int, u32, i64, i64, vp, C.POINTER(Stats)]
    L.gat_sample_and_count_enqueue.restype = C.c_int
    L.gat_sample_and_count_enqueue.argtypes = [vp, vp, vp, C.c_int, u32, i64, i64, vp]
    L.gat_wait.restype = C.c_int
    L.gat_wait.argtypes = [vp, vp, C.POINTER(Stats)]
    L.gat_sample_and_count_serial.restype = C.c_int
    L.gat_sample_and_count_serial.argtypes = [vp, vp, vp, C.c_int, vp, i64, vp, C.POINTER(Stats)]
    L.gat_mt19937_seed.restype = None
    L.gat_mt19937_seed.argtypes = [u32, vp]
    L.gat_sample.restype = C.c_int
    L.gat_sample.argtypes = [vp, vp, u32, i64, i64, vp, i64, vp, C.POINTER(Stats)]
    L.gat_sample_units.restype = C.c_int
    L.gat_sample_units.argtypes = [vp, vp, u32, i64, i64, vp, i64, vp, C.POINTER(Stats)]
    L.gat_count_lists.restype = C.c_int
    L.gat_count_lists.argtypes = [vp, vp, C.c_int, vp, vp, i64, vp, vp, i32, vp, i32, vp]
    L.gat_count_list_ranges.restype = C.c_int
    L.gat_count_list_ranges.argtypes = [vp, vp, C.c_int, vp, vp, i64, vp, vp, vp, i32, vp, i32, vp]
    L.gat_intersection_sizes.restype = C.c_int
    L.gat_intersection_sizes.argtypes = [vp, vp, i32, vp, vp, vp, i32, vp, vp]
    L.gat_problem_rng_rows.restype = i64
    L.gat_problem_rng_rows.argtypes = [vp]
    L.gat_list_sums.restype = C.c_int
    L.gat_list_sums.argtypes = [vp, vp, vp, i64, vp]
    L.gat_isochore_split.restype = C.c_int
    L.gat_isochore_split.argtypes = [vp, vp, vp, i64, vp, vp, vp, i64, i32, i32, vp, vp, vp]
    L.gat_problem_info.restype = C.c_int
    L.gat_problem_info.argtypes = [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]
    L.gat_null_stats.restype = C.c_int
    L.gat_null_stats.argtypes = [vp, vp, i64, i64, vp, vp, i64, i64, vp]
    L.gat_comm_unique_id.restype = C.c_int
    L.gat_comm_unique_id.argtypes = [vp]
    L.gat_comm_library_preloaded.restype = C.c_int
    L.gat_comm_library_preloaded.argtypes = []
    L.gat_ctx_set_option.restype = C.c_int
    L.gat_ctx_set_option.argtypes = [vp, C.c_char_p, C.c_char_p]
    L.gat_ctx_get_option.restype = C.c_char_p
    L.gat_ctx_get_option.argtypes = [vp, C.c_char_p]
    L.gat_comm_create.restype = C.c_int
    L.gat_comm_create.argtypes = [vp, C.POINTER(vp), C.c_int, C.c_int, vp]
    L.gat_comm_destroy.restype = None
    L.gat_comm_destroy.argtypes = [vp]
    L.gat_allgather_counts.restype = C.c_int
    L.gat_allgather_counts.argtypes = [vp, vp, vp, vp, i64]
    _LIB = L
    return L


_ERRORS = {-1: ValueError, -2: AssertionError, -3: GatError, -4: MemoryError, -5: GatError, -6: ValueError}


def _check(rc, ctx=None):
    if rc == 0:
        return
    msg = lib().gat_last_error(ctx).decode("utf-8", "replace")
    raise _ERRORS.get(rc, GatError)("gat_mi355 error %d: %s" % (rc, msg))


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class _Options(object):
    """a context's tuning / testing knobs (gat_ctx_set_option) as a mapping: options["GAT_X"] = "1" sets, del / pop go back to
    the process's value (the environment as the library first saw it), "" means "not set whatever the environment says".
    pytest's monkeypatch.setitem(ctx.options, key, value) undoes itself."""

    def __init__(self, ctx):
        self._ctx = ctx
        self._mine = {}

    def __setitem__(self, key, value):
        _check(lib().gat_ctx_set_option(self._ctx._h, key.encode(), str(value).encode()), self._ctx._h)
        self._mine[key] = str(value)

    def __delitem__(self, key):
        _check(lib().gat_ctx_set_option(self._ctx._h, key.encode(), None), self._ctx._h)
        del self._mine[key]

    def __getitem__(self, key):
        return self._mine[key]                      # (KeyError: not set on this context -- what monkeypatch.setitem asks)

    def __contains__(self, key):
        return key in self._mine

    def get(self, key, default=None):
        """the value in force: this context's, else the process's; None: not set"""
        v = lib().gat_ctx_get_option(self._ctx._h, key.encode())
        return v.decode() if v is not None else default

    def pop(self, key, default=None):
        if key in self._mine:
            v = self._mine[key]
            del self[key]
            return v
        return default

    def update(self, **kw):
        for k, v in kw.items():
            self[k] = v


class Context(object):
    """one HIP device + stream (gat_ctx)."""

    def __init__(self, device=0, stream=None):
        """stream: a hipStream_t handle to run on (e.g. torch.cuda.Stream(...).cuda_stream); None: the context makes a
        private non-blocking stream; 0 -- the handle torch reports for its DEFAULT stream -- means that stream (it is
        passed on as hipStreamLegacy: a NULL handle is the C ABI's "make a private stream")."""
        self._h = C.c_void_p()
        if stream is None:
            handle = None
        else:
            handle = C.c_void_p(int(stream) if int(stream) != 0 else 1)      # hipStreamLegacy == (hipStream_t)1
        _check(lib().gat_ctx_create(C.byref(self._h), int(device), handle))
        self.device = device
        self.options = _Options(self)           # the knobs (GAT_*): ctx.options["GAT_NO_SPLIT"] = "1"

    def close(self):
        if self._h:
            lib().gat_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(lib().gat_ctx_synchronize(self._h), self._h)

    def stream_handle(self):
        """the hipStream_t the context enqueues on, as an integer (torch.cuda.ExternalStream(handle) orders torch's work
        against the library's with events, no device-wide synchronisation)"""
        return int(lib().gat_ctx_stream(self._h) or 0)

    def set_kernel_times(self, on):
        """per-kernel device times in the statistics of a call (ms_rng, ms_place, ...): off by default, an event behind
        every kernel costs a call 50-60 us"""
        _check(lib().gat_ctx_set_kernel_times(self._h, 1 if on else 0), self._h)

    def alloc(self, nbytes):
        p = C.c_void_p()
        _check(lib().gat_dev_alloc(self._h, C.byref(p), nbytes), self._h)
        return p.value

    def free(self, ptr):
        _check(lib().gat_dev_free(self._h, C.c_void_p(ptr)), self._h)

    def d2h(self, host_array, dev_ptr):
        _check(lib().gat_memcpy_d2h(self._h, _p(host_array), C.c_void_p(dev_ptr), host_array.nbytes), self._h)

    def null_stats(self, counts_dev_ptr, n_rows, n_samples, is_double, vals):
        """per row of a device count matrix: (mean, std, lower95 value, upper95 value, n < val, n == val) -- what
        AnnotatorResult reads off a null distribution (gat/Engine.pyx:1635-1718), numpy.mean / numpy.std bit for bit."""
        l = int(n_samples)  # noqa: E741
        offset = int(0.05 * l)
        lo_i, hi_i = (min(offset, l - 1), max(l - offset, 0)) if offset > 0 else (0, l - 1)      # gat/Engine.pyx:1689-1696
        hi_i = min(hi_i, l - 1)
        is_double = np.ascontiguousarray(is_double, dtype=np.uint8)
        vals = np.ascontiguousarray(vals, dtype=np.float64)
        out = np.zeros((int(n_rows), 8), dtype=np.float64)
        _check(lib().gat_null_stats(self._h, C.c_void_p(counts_dev_ptr), int(n_rows), l, _p(is_double), _p(vals), lo_i, hi_i,
                                    _p(out)), self._h)
        out[:, 1] = np.sqrt(out[:, 1] / l)                # numpy.std's last two steps (sum of squares from the device)
        return out

    def count_lists(self, counters, lists, list_off, n_lists, annos, anno_off, n_tracks, ws_nseg, n_groups, anno_end=None):
        """Counter*(list, annotation, workspace) for n_lists x n_groups lists (observed counts).  anno_end given: the
        annotation lists are the ranges annos[anno_off[l]:anno_end[l]] (gat_count_list_ranges), else anno_off is a CSR."""
        ids = np.array([COUNTER_IDS[c] for c in counters], dtype=np.int32)
        lists = np.ascontiguousarray(lists, dtype=SEG)
        annos = np.ascontiguousarray(annos, dtype=SEG)
        list_off = np.ascontiguousarray(list_off, dtype=np.int64)
        anno_off = np.ascontiguousarray(anno_off, dtype=np.int64)
        ws_nseg = np.ascontiguousarray(ws_nseg, dtype=np.int64)
        out = np.zeros((len(ids), n_tracks, n_lists), dtype=np.int64)
        if anno_end is None:
            _check(lib().gat_count_lists(self._h, _p(ids), len(ids), _p(lists), _p(list_off), n_lists, _p(annos),
                                         _p(anno_off), n_tracks, _p(ws_nseg), n_groups, _p(out)), self._h)
        else:
            anno_end = np.ascontiguousarray(anno_end, dtype=np.int64)
            assert len(anno_off) == len(anno_end) == n_tracks * n_groups
            _check(lib().gat_count_list_ranges(self._h, _p(ids), len(ids), _p(lists), _p(list_off), n_lists, _p(annos),
                                               _p(anno_off), _p(anno_end), n_tracks, _p(ws_nseg), n_groups, _p(out)), self._h)
        return [out[k].view(np.float64).copy() if c == "nucleotide-density" else out[k].copy()
                for k, c in enumerate(counters)]


def intersection_sizes(a, a_off, b, b_begin, b_end, n_tracks):
    """(pairs, bases) per track of b: segments and bases of the intersection of the dictionary a (len(a_off) - 1 normalized
    lists) with each of n_tracks dictionaries given as ranges of b (gat_intersection_sizes: the overlap_* columns)."""
    a = np.ascontiguousarray(a, dtype=SEG)
    b = np.ascontiguousarray(b, dtype=SEG)
    a_off = np.ascontiguousarray(a_off, dtype=np.int64)
    b_begin = np.ascontiguousarray(b_begin, dtype=np.int64)
    b_end = np.ascontiguousarray(b_end, dtype=np.int64)
    n_groups = len(a_off) - 1
    assert len(b_begin) == len(b_end) == n_tracks * n_groups
    pairs = np.zeros(n_tracks, dtype=np.int64)
    bases = np.zeros(n_tracks, dtype=np.int64)
    _check(lib().gat_intersection_sizes(_p(a), _p(a_off), n_groups, _p(b), _p(b_begin), _p(b_end), n_tracks, _p(pairs), _p(bases)))
    return pairs, bases


def list_sums(a, begin, end):
    """SegmentList.sum() (a uint32 accumulator, gat/SegmentList.pyx:1607) of every list a[begin[l]:end[l]] (gat_list_sums)"""
    a = np.ascontiguousarray(a, dtype=SEG)
    begin = np.ascontiguousarray(begin, dtype=np.int64)
    end = np.ascontiguousarray(end, dtype=np.int64)
    assert len(begin) == len(end)
    out = np.zeros(len(begin), dtype=np.int64)
    _check(lib().gat_list_sums(_p(a), _p(begin), _p(end), len(begin), _p(out)))
    return out


def isochore_split(arrays, contig_ids, cls_start, cls_end, cls_label, n_classes, truncate):
    """toIsochores of many lists at once (gat_isochore_split, host threads): arrays -- the lists' SEG arrays (contiguous,
    kept alive by the caller), contig_ids[l] -- the contig of list l in the numbering of the class segments' coordinates.
    Returns (out, out_off): list (l, k) = out[out_off[l * n_classes + k]:out_off[l * n_classes + k + 1]]; None if a list is
    not normalized (the caller's list-by-list form raises what the reference raises)."""
    n = len(arrays)
    # (the address of a list's data: the buffer protocol is the cheapest way to it -- 1 us; __array_interface__ of a structured
    #  array builds its descriptor every time, 7 us; an empty list has no buffer to speak of and is never read)
    addr, from_buffer = C.addressof, C.c_char.from_buffer
    ptr = np.fromiter((addr(from_buffer(a)) if len(a) and a.flags["WRITEABLE"] else a.ctypes.data for a in arrays),
                      dtype=np.uint64, count=n)
    lens = np.fromiter((len(a) for a in arrays), dtype=np.int64, count=n)
    cid = np.ascontiguousarray(contig_ids, dtype=np.int64)
    cls_start = np.ascontiguousarray(cls_start, dtype=np.int64)
    cls_end = np.ascontiguousarray(cls_end, dtype=np.int64)
    cls_label = np.ascontiguousarray(cls_label, dtype=np.int64)
    off = np.zeros(n * n_classes + 1, dtype=np.int64)
    total = C.c_int64(0)
    args = (_p(ptr), _p(lens), _p(cid), n, _p(cls_start), _p(cls_end), _p(cls_label), len(cls_start), int(n_classes), 1 if truncate else 0)
    rc = lib().gat_isochore_split(*args, None, _p(off), C.byref(total))
    if rc == 1:
        return None
    _check(rc)
    out = np.empty(max(1, total.value), dtype=SEG)
    _check(lib().gat_isochore_split(*args, _p(out), _p(off), C.byref(total)))
    return out[:total.value], off


COMM_ID_BYTES = 128


def comm_unique_id():
    """128 bytes that rank 0 hands to the other ranks (ncclGetUniqueId through the C ABI)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(lib().gat_comm_unique_id(buf))
    return buf.raw


class Comm(object):
    """RCCL communicator of the C ABI (gat_comm): the all-gather of the per-rank count blocks without torch."""

    def __init__(self, ctx, n_ranks, rank, unique_id):
        self.ctx, self.n_ranks, self.rank = ctx, n_ranks, rank
        self._h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        _check(lib().gat_comm_create(ctx._h, C.byref(self._h), int(n_ranks), int(rank), buf), ctx._h)

    def allgather_counts(self, send_dev_ptr, recv_dev_ptr, n_slots):
        _check(lib().gat_allgather_counts(self.ctx._h, self._h, C.c_void_p(send_dev_ptr), C.c_void_p(recv_dev_ptr), int(n_slots)),
               self.ctx._h)

    def close(self):
        if self._h:
            lib().gat_comm_destroy(self._h)
            self._h = C.c_void_p()


class Annotations(object):
    """the annotation side of a problem as a device-resident object of its own (gat_annotations): made once per run(),
    shared by the problems of every segment track whose contigs are `flat`'s (same names, same order)."""

    def __init__(self, ctx, flat, mean_segment_length=0.0, asynchronous=False, nucleotide_only=False):
        """asynchronous: the tables are built by a thread of the library while the caller goes on (GAT_ANNOTATIONS_ASYNC);
        the arrays handed over stay referenced by this object.  nucleotide_only: only the nucleotide counters will be asked
        for (GAT_ANNOTATIONS_NUCLEOTIDE_ONLY: where the merged index is built the per-track tables are left out)."""
        self.ctx = ctx
        keep = self._keep = {}

        def arr(name, dtype):
            keep[name] = np.ascontiguousarray(flat[name], dtype=dtype)
            return _p(keep[name])

        d = AnnotationsDesc()
        d.n_tracks, d.n_contigs, d.merge_contigs = int(flat["n_tracks"]), int(flat["n_contigs"]), int(flat["merge_contigs"])
        d.annos = arr("annos", SEG)
        d.anno_off = arr("anno_off", np.int64)
        if flat.get("anno_group") is not None:
            d.n_anno_lists = len(flat["anno_group"])
            d.anno_group = arr("anno_group", np.int32)
            d.anno_end = arr("anno_end", np.int64)
            assert len(keep["anno_off"]) == len(keep["anno_end"]) == d.n_anno_lists
        else:
            assert len(keep["anno_off"]) == d.n_tracks * d.n_contigs + 1
        d.mean_segment_length = float(mean_segment_length)
        d.flags = (1 if asynchronous else 0) | (2 if nucleotide_only else 0)
        self.n_tracks, self.n_contigs, self.merge_contigs = d.n_tracks, d.n_contigs, d.merge_contigs
        self._h = C.c_void_p()
        _check(lib().gat_annotations_create(ctx._h, C.byref(d), C.byref(self._h)), ctx._h)

    def wait(self):
        """an asynchronous build has finished (raises its error, if any)"""
        _check(lib().gat_annotations_wait(self.ctx._h, self._h), self.ctx._h)

    def close(self):
        if self._h:
            if getattr(self.ctx, "_h", None):
                lib().gat_annotations_wait(self.ctx._h, self._h)     # (a build still running has the arrays below in use)
            lib().gat_annotations_destroy(self._h)
            self._h = C.c_void_p()
        self._keep = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Problem(object):
    """device-resident inputs of one segment track (gat_problem).

    `flat` is a mapping with the fields of gat_problem_desc (numpy arrays / ints); annotations: an Annotations object made
    for the same tracks and contigs (flat's annotation lists are then not looked at and may be missing)."""

    def __init__(self, ctx, flat, annotations=None):
        self.ctx = ctx
        keep = {}

        def arr(name, dtype):
            keep[name] = np.ascontiguousarray(flat[name], dtype=dtype)
            return _p(keep[name])

        d = ProblemDesc()
        d.n_units = int(flat["n_units"])
        d.segs = arr("segs", SEG)
        d.seg_off = arr("seg_off", np.int64)
        d.ws = arr("ws", SEG)
        d.ws_off = arr("ws_off", np.int64)
        d.unit_contig = arr("unit_contig", np.int32)
        d.n_contigs = int(flat["n_contigs"])
        d.merge_contigs = int(flat["merge_contigs"])
        d.n_tracks = int(flat["n_tracks"])
        if annotations is None:
            d.annos = arr("annos", SEG)
            d.anno_off = arr("anno_off", np.int64)
        else:
            d.annotations = annotations._h
        d.cws_nseg = arr("cws_nseg", np.int64)
        d.bucket_size = int(flat.get("bucket_size", 0))
        d.nbuckets = int(flat.get("nbuckets", 100000))
        d.sampler = int(flat.get("sampler", 0))
        assert len(keep["seg_off"]) == d.n_units + 1 and len(keep["ws_off"]) == d.n_units + 1
        assert len(keep["cws_nseg"]) == d.n_contigs
        if annotations is not None:
            pass
        elif flat.get("anno_group") is not None:
            # the annotation lists as the host holds them (one per track and key), grouped into contigs by the library
            d.n_anno_lists = len(flat["anno_group"])
            d.anno_group = arr("anno_group", np.int32)
            d.anno_end = arr("anno_end", np.int64)
            assert len(keep["anno_off"]) == len(keep["anno_end"]) == d.n_anno_lists
        else:
            assert len(keep["anno_off"]) == d.n_tracks * d.n_contigs + 1
        self.n_units, self.n_contigs, self.n_tracks = d.n_units, d.n_contigs, d.n_tracks
        self._h = C.c_void_p()
        _check(lib().gat_problem_create(ctx._h, C.byref(d), C.byref(self._h)), ctx._h)
        self.last_stats = None

    def close(self):
        if self._h:
            lib().gat_problem_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        v = [C.c_int64() for _ in range(5)]
        _check(lib().gat_problem_info(self._h, *[C.byref(x) for x in v]))
        return dict(n_units=v[0].value, n_contigs=v[1].value, n_tracks=v[2].value,
                    slab_segments_per_sample=v[3].value, algorithmic_bytes_per_sample=v[4].value)

    def rows_per_sample(self):
        """raw MT19937 outputs generated ahead per sample, over all units (the rows of k_rng)"""
        return int(lib().gat_problem_rng_rows(self._h))

    def sample_and_count_device(self, counters, seed, sample_begin, sample_end, counts_dev_ptr):
        """the batch seam, results left on the device ([counter][track][sample] 8-byte slots)."""
        ids = np.array([COUNTER_IDS[c] for c in counters], dtype=np.int32)
        st = Stats()
        _check(lib().gat_sample_and_count(self.ctx._h, self._h, _p(ids), len(ids), int(seed) & 0xFFFFFFFF,
                                          int(sample_begin), int(sample_end), C.c_void_p(counts_dev_ptr), C.byref(st)),
               self.ctx._h)
        self.last_stats = st.asdict()
        return self.last_stats

    def enqueue(self, counters, seed, sample_begin, sample_end, counts_dev_ptr):
        """first half of the batch seam (gat_sample_and_count_enqueue): the call's batches are put on the context's stream
        and the host goes on; counts_dev_ptr must stay valid until wait()."""
        ids = np.array([COUNTER_IDS[c] for c in counters], dtype=np.int32)
        _check(lib().gat_sample_and_count_enqueue(self.ctx._h, self._h, _p(ids), len(ids), int(seed) & 0xFFFFFFFF,
                                                  int(sample_begin), int(sample_end), C.c_void_p(counts_dev_ptr)), self.ctx._h)

    def wait(self):
        """second half (gat_wait): blocks until the call has completed, raises what sample_and_count_device would have"""
        st = Stats()
        _check(lib().gat_wait(self.ctx._h, self._h, C.byref(st)), self.ctx._h)
        self.last_stats = st.asdict()
        return self.last_stats

    def sample_and_count(self, counters, seed, sample_begin, sample_end):
        """returns a list (per counter) of [n_tracks, n_samples] arrays (int64 / float64 for density)."""
        ns = sample_end - sample_begin
        nslots = max(1, len(counters) * self.n_tracks * ns)
        dev = self.ctx.alloc(nslots * 8)
        try:
            self.sample_and_count_device(counters, seed, sample_begin, sample_end, dev)
            host = np.empty((len(counters), self.n_tracks, ns), dtype=np.int64)
            if host.size:
                self.ctx.d2h(host, dev)
        finally:
            self.ctx.free(dev)
        return [host[k].view(np.float64) if c == "nucleotide-density" else host[k] for k, c in enumerate(counters)]

    def sample_and_count_serial(self, counters, mt_state, n_samples):
        """samples 0 .. n_samples-1 drawn from ONE MT19937 stream in the reference's order (an unpatched gat-run.py
        --random-seed): mt_state (mt19937_seed(seed), or what an earlier call left) is advanced in place.  Returns the
        count matrices like sample_and_count."""
        assert mt_state.dtype == np.uint32 and mt_state.size == MT_STATE_WORDS and mt_state.flags["C_CONTIGUOUS"]
        ids = np.array([COUNTER_IDS[c] for c in counters], dtype=np.int32)
        ns = int(n_samples)
        dev = self.ctx.alloc(max(1, len(counters) * self.n_tracks * ns) * 8)
        st = Stats()
        try:
            _check(lib().gat_sample_and_count_serial(self.ctx._h, self._h, _p(ids), len(ids), _p(mt_state), ns,
                                                     C.c_void_p(dev), C.byref(st)), self.ctx._h)
            host = np.zeros((len(counters), self.n_tracks, ns), dtype=np.int64)
            if host.size:
                self.ctx.d2h(host, dev)
        finally:
            self.ctx.free(dev)
        self.last_stats = st.asdict()
        return [host[k].view(np.float64).copy() if c == "nucleotide-density" else host[k].copy()
                for k, c in enumerate(counters)]

    def sample(self, seed, sample_begin, sample_end, unit_level=False):
        """sampled contig-level segment lists: (segments SEG array, offsets[(n_samples*n_contigs)+1]); with
        unit_level the lists of every (sample, unit) as the sampler returned them (offsets per n_units)."""
        ns = sample_end - sample_begin
        off = np.zeros(ns * (self.n_units if unit_level else self.n_contigs) + 1, dtype=np.int64)
        cap = max(1024, self.info()["slab_segments_per_sample"] * ns // 2)
        st = Stats()
        fn = lib().gat_sample_units if unit_level else lib().gat_sample
        while True:
            out = np.empty(cap, dtype=SEG)
            rc = fn(self.ctx._h, self._h, int(seed) & 0xFFFFFFFF, int(sample_begin), int(sample_end),
                    _p(out), cap, _p(off), C.byref(st))
            if rc == -3 and off[-1] > cap:
                cap = int(off[-1])
                continue
            _check(rc, self.ctx._h)
            break
        self.last_stats = st.asdict()
        return out[: off[-1]].copy(), off

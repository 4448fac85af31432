"""GPU: the batch seam in two halves (gat_sample_and_count_enqueue / gat_wait, include/gat_mi355.h) -- what the reference does
with map_async over its process pool (gat/__init__.py:681-700): the host goes on while the samples are computed.  Same
columns as the blocking call and as the oracle, whatever is enqueued ahead, repeated or interleaved."""
import numpy as np
import pytest

from gat_amd import _lib
from oracle import oracle as O
from test_hip_parity import _big_problem, _random_problem

pytestmark = pytest.mark.gpu
COUNTERS = ["nucleotide-overlap", "nucleotide-density", "segment-overlap", "annotation-overlap"]


@pytest.fixture(scope="module")
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def _enqueue_wait(ctx, P, counters, seed, lo, hi, between=None):
    n = max(1, len(counters) * P.n_tracks * (hi - lo))
    dev = ctx.alloc(n * 8)
    try:
        P.enqueue(counters, seed, lo, hi, dev)
        if between is not None:
            between()
        st = P.wait()
        host = np.empty((len(counters), P.n_tracks, hi - lo), dtype=np.int64)
        if host.size:
            ctx.d2h(host, dev)
    finally:
        ctx.free(dev)
    return [host[k].view(np.float64) if c == "nucleotide-density" else host[k] for k, c in enumerate(counters)], st


@pytest.mark.parametrize("seed", [11, 12, 13, 14])
def test_enqueue_wait_equals_the_blocking_call_and_the_oracle(ctx, seed):
    rs = np.random.RandomState(seed)
    flat = _random_problem(rs, n_contigs=int(rs.randint(1, 6)), n_segs=int(rs.randint(20, 600)),
                           n_tracks=int(rs.randint(1, 6)), isochores=bool(seed % 2))
    S = 48
    want, _ = O.run_samples(flat, COUNTERS, 500 + seed, 1, 7, 7 + S)
    P = _lib.Problem(ctx, flat)
    # the host does device work of its own on the same context between the two halves (run() computes observed counts there)
    other = _lib.Problem(ctx, flat)
    between = lambda: other.sample_and_count(["nucleotide-overlap"], 1, 0, 5)      # noqa: E731
    got, st = _enqueue_wait(ctx, P, COUNTERS, 500 + seed, 7, 7 + S, between)
    blocking = P.sample_and_count(COUNTERS, 500 + seed, 7, 7 + S)
    for k, c in enumerate(COUNTERS):
        assert np.array_equal(got[k], want[k]), c
        assert np.array_equal(blocking[k], want[k]), c
    assert st["n_batches"] == 1 and st["ms_total"] > 0
    other.close()
    P.close()


def test_more_batches_than_one_flight_holds(ctx, monkeypatch):
    """a scratch budget of a few samples: the call is dozens of batches, enqueued eight at a time"""
    rs = np.random.RandomState(77)
    flat = _big_problem(rs, 700, 5)
    counters = ["nucleotide-overlap", "segment-overlap", "annotation-overlap"]
    S = 150
    want, _ = O.run_samples(flat, counters, 31, 1, 2, 2 + S)
    monkeypatch.setitem(ctx.options, "GAT_SLAB_BYTES", "400000")
    P = _lib.Problem(ctx, flat)
    got, st = _enqueue_wait(ctx, P, counters, 31, 2, 2 + S)
    assert st["n_batches"] > 8, st
    for k, c in enumerate(counters):
        assert np.array_equal(got[k], want[k]), c
    P.close()


def test_overflow_in_flight_repeats_the_batch_and_what_was_behind_it(ctx, monkeypatch):
    """tiny slab regions: a batch in the middle of a flight overflows, the slab is laid out again (the batch that fits the
    budget shrinks) and everything from that batch on is redone inside gat_wait"""
    rs = np.random.RandomState(78)
    flat = _big_problem(rs, 700, 5)
    counters = ["nucleotide-overlap", "segment-overlap", "annotation-overlap"]
    S = 90
    want, wsamples = O.run_samples(flat, counters, 32, 1, 0, S, want_samples=True)
    monkeypatch.setitem(ctx.options, "GAT_TEST_SMALL_CAPS", "1")
    monkeypatch.setitem(ctx.options, "GAT_SLAB_BYTES", "300000")
    P = _lib.Problem(ctx, flat)
    got, st = _enqueue_wait(ctx, P, counters, 32, 0, S)
    assert st["n_retried"] > 0 and st["n_batches"] > 2, st
    for k, c in enumerate(counters):
        assert np.array_equal(got[k], want[k]), c
    seg, off = P.sample(32, 0, S)
    assert np.array_equal(off, wsamples[1]) and np.array_equal(seg, wsamples[0])
    P.close()


def test_two_problems_in_flight_on_one_context(ctx):
    """run() keeps the next segment track's call enqueued behind the current one's"""
    rs = np.random.RandomState(5)
    flats = [_random_problem(rs, 3, 200, 3, True), _random_problem(rs, 2, 300, 2, False)]
    S = 40
    wants = [O.run_samples(f, COUNTERS, 9 + i, 1, 0, S)[0] for i, f in enumerate(flats)]
    Ps = [_lib.Problem(ctx, f) for f in flats]
    devs = [ctx.alloc(len(COUNTERS) * P.n_tracks * S * 8) for P in Ps]
    for i, (P, d) in enumerate(zip(Ps, devs)):
        P.enqueue(COUNTERS, 9 + i, 0, S, d)
    for i in (1, 0):                                       # (waited for in the other order)
        Ps[i].wait()
        host = np.empty((len(COUNTERS), Ps[i].n_tracks, S), dtype=np.int64)
        ctx.d2h(host, devs[i])
        for k, c in enumerate(COUNTERS):
            got = host[k].view(np.float64) if c == "nucleotide-density" else host[k]
            assert np.array_equal(got, wants[i][k]), (i, c)
    for P, d in zip(Ps, devs):
        ctx.free(d)
        P.close()


def test_steps_taking_turns_on_two_problems_over_the_same_inputs(ctx, monkeypatch):
    """bench.py's pipeline (and run()'s tracks): step i+1 is enqueued before step i is waited for.  gat_wait returns at the
    call's OWN end (its event), with the other problem's kernels still running behind it -- the statistics, the status words
    and the counts it hands out are the finished call's; one of the steps overflows its slab on the way and is redone"""
    import time
    rs = np.random.RandomState(91)
    flat = _big_problem(rs, 700, 3)
    counters = ["nucleotide-overlap", "segment-overlap"]
    S, steps = 64, 6
    wants = [O.run_samples(flat, counters, 77, 1, i * S, (i + 1) * S)[0] for i in range(steps)]
    A = _lib.Annotations(ctx, flat)
    monkeypatch.setitem(ctx.options, "GAT_TEST_SMALL_CAPS", "1")
    monkeypatch.setitem(ctx.options, "GAT_SLAB_BYTES", "300000")            # (tiny slabs: a call is several batches, some overflow)
    Ps = [_lib.Problem(ctx, flat, annotations=A) for _ in range(2)]
    small = _lib.Problem(ctx, flat, annotations=A)
    devs = [ctx.alloc(len(counters) * flat["n_tracks"] * S * 8) for _ in range(2)]

    def check(i):
        host = np.empty((len(counters), flat["n_tracks"], S), dtype=np.int64)
        ctx.d2h(host, devs[i & 1])
        for k, c in enumerate(counters):
            assert np.array_equal(host[k], wants[i][k]), (i, c)

    order = [Ps[0], Ps[1], small, Ps[1], Ps[0], small]         # (a problem's next call only after its last one was waited for)
    retried = 0
    for i in range(steps):
        order[i].enqueue(counters, 77, i * S, (i + 1) * S, devs[i & 1])
        if i > 0:
            retried += order[i - 1].wait()["n_retried"]
            check(i - 1)
    retried += order[steps - 1].wait()["n_retried"]
    check(steps - 1)
    assert retried > 0
    monkeypatch.delitem(ctx.options, "GAT_TEST_SMALL_CAPS")
    monkeypatch.delitem(ctx.options, "GAT_SLAB_BYTES")
    for P in Ps:
        P.close()
    Ps = [_lib.Problem(ctx, flat, annotations=A) for _ in range(2)]
    # the wait is for the call, not for the stream: a short call's wait returns while a long one enqueued behind it runs
    big = ctx.alloc(len(counters) * flat["n_tracks"] * 40000 * 8)
    Ps[0].enqueue(counters, 5, 0, 8, devs[0])
    Ps[1].enqueue(counters, 5, 0, 40000, big)      # (tens of milliseconds: a stalled host thread does not decide the test)
    t0 = time.perf_counter()
    Ps[0].wait()
    t1 = time.perf_counter()
    Ps[1].wait()
    t2 = time.perf_counter()
    assert (t1 - t0) < 0.5 * (t2 - t0), (t1 - t0, t2 - t0)
    for d in devs + [big]:
        ctx.free(d)
    for P in Ps + [small]:
        P.close()
    A.close()


def test_seam_errors(ctx):
    rs = np.random.RandomState(6)
    flat = _random_problem(rs, 2, 100, 2, False)
    P = _lib.Problem(ctx, flat)
    with pytest.raises(ValueError):
        P.wait()                                           # nothing in flight
    dev = ctx.alloc(2 * P.n_tracks * 16 * 8)
    P.enqueue(["nucleotide-overlap"], 1, 0, 16, dev)
    with pytest.raises(ValueError):
        P.enqueue(["nucleotide-overlap"], 1, 0, 16, dev)   # one call per problem
    with pytest.raises(ValueError):
        P.sample(1, 0, 4)                                  # the scratch is in use
    P.wait()
    with pytest.raises(ValueError):
        P.enqueue(["nucleotide-overlap", "nucleotide-overlap"], 1, 0, 16, dev)   # (arguments are checked at enqueue)
    with pytest.raises(ValueError):
        P.wait()
    P.enqueue(["nucleotide-overlap"], 1, 0, 0, dev)        # an empty range is a call like any other
    assert P.wait()["n_batches"] == 0
    P.enqueue(["nucleotide-overlap"], 1, 0, 16, dev)
    P.close()                                              # destroyed with a call in flight: dropped, nothing leaks
    ctx.free(dev)


def test_context_closed_before_its_problem():
    """a host may close the handles in either order (the context lives until the last problem made on it is gone)"""
    rs = np.random.RandomState(7)
    flat = _random_problem(rs, 2, 100, 2, False)
    want, _ = O.run_samples(flat, ["nucleotide-overlap"], 3, 1, 0, 8)
    c = _lib.Context(0)
    P = _lib.Problem(c, flat)
    got = P.sample_and_count(["nucleotide-overlap"], 3, 0, 8)
    assert np.array_equal(got[0], want[0])
    c.close()
    P.close()


@pytest.mark.parametrize("n_tracks,isochores", [(6, True), (5, False), (2, True), (1, False), (3, False)])
def test_annotation_tables_built_while_the_device_samples(ctx, n_tracks, isochores):
    """gat_annotations_create with GAT_ANNOTATIONS_ASYNC: the problem is sampled at once, the count kernels follow when the
    tables are there; fewer than four tracks: the build is synchronous (the count kernel's route is not known beforehand).
    Two segment tracks share the object."""
    rs = np.random.RandomState(100 + n_tracks)
    flat = _random_problem(rs, 4, 400, n_tracks, isochores)
    counters = ["nucleotide-overlap", "nucleotide-density"]
    S = 64
    want, _ = O.run_samples(flat, counters, 17, 1, 0, S)
    A = _lib.Annotations(ctx, flat, mean_segment_length=60.0, asynchronous=True)
    units = dict(flat)
    for k in ("annos", "anno_off", "anno_end", "anno_group"):
        units[k] = None
    P1, P2 = _lib.Problem(ctx, units, annotations=A), _lib.Problem(ctx, units, annotations=A)
    got1, st = _enqueue_wait(ctx, P1, counters, 17, 0, S)
    A.close()                                               # (the problems keep it alive)
    got2 = P2.sample_and_count(counters + ["segment-overlap"], 17, 0, S)
    for k, c in enumerate(counters):
        assert np.array_equal(got1[k], want[k]), c
        assert np.array_equal(got2[k], want[k]), c
    assert P1.info()["algorithmic_bytes_per_sample"] == _lib.Problem(ctx, flat).info()["algorithmic_bytes_per_sample"]
    P1.close()
    P2.close()


@pytest.mark.parametrize("n_tracks,isochores", [(6, True), (5, False), (2, False)])
def test_annotation_tables_for_the_nucleotide_counters_only(ctx, n_tracks, isochores):
    """GAT_ANNOTATIONS_NUCLEOTIDE_ONLY: with four tracks or more only the merged index is built -- the nucleotide counters
    equal the oracle, any other counter is refused (GAT_ERR_ARG -> ValueError); with fewer tracks (no merged index for lists
    that fit the per-track kernel) the flag changes nothing."""
    rs = np.random.RandomState(300 + n_tracks)
    flat = _random_problem(rs, 4, 400, n_tracks, isochores)
    counters = ["nucleotide-overlap", "nucleotide-density"]
    S = 48
    want, _ = O.run_samples(flat, counters + ["segment-overlap"], 23, 1, 0, S)
    A = _lib.Annotations(ctx, flat, mean_segment_length=60.0, asynchronous=True, nucleotide_only=True)
    units = dict(flat)
    for k in ("annos", "anno_off", "anno_end", "anno_group"):
        units[k] = None
    P = _lib.Problem(ctx, units, annotations=A)
    try:
        got = P.sample_and_count(counters, 23, 0, S)
        for k, c in enumerate(counters):
            assert np.array_equal(got[k], want[k]), c
        if n_tracks >= 4:
            with pytest.raises(ValueError):
                P.sample_and_count(counters + ["segment-overlap"], 23, 0, S)
            assert np.array_equal(P.sample_and_count(counters, 23, 0, S)[0], want[0])      # (and the object is still good)
        else:
            got = P.sample_and_count(counters + ["segment-overlap"], 23, 0, S)
            assert np.array_equal(got[2], want[2])
    finally:
        P.close()
        A.close()


def test_an_error_of_the_asynchronous_build_is_reported_by_the_call_that_needs_the_tables(ctx):
    rs = np.random.RandomState(3)
    flat = dict(_random_problem(rs, 3, 200, 5, False))
    annos = np.array(flat["annos"], copy=True)
    lo = int(flat["anno_off"][2])
    assert flat["anno_off"][3] - lo >= 2
    annos[lo + 1]["start"] = annos[lo]["start"]            # list 2 is no longer normalized
    flat["annos"] = annos
    A = _lib.Annotations(ctx, flat, asynchronous=True)
    units = dict(flat, annos=None, anno_off=None)
    P = _lib.Problem(ctx, units, annotations=A)
    dev = ctx.alloc(P.n_tracks * 8 * 8)
    with pytest.raises(AssertionError):                     # (by the enqueue already if the build has finished by then)
        P.enqueue(["nucleotide-overlap"], 1, 0, 8, dev)     # the sampler's kernels are on their way
        P.wait()
    with pytest.raises(AssertionError):
        A.wait()
    ctx.free(dev)
    P.close()
    A.close()

"""GPU parity on the BASELINE.json workloads themselves (configs 2-5 at their full interval counts).

Every configuration is compared with the CPU oracle on its own inputs: the count matrix of the configuration's
counter and the sampled lists for a few samples (what the oracle finishes in seconds), then -- at the sample counts
a GPU of the configuration actually runs in one call -- columns picked from a full-size launch against the oracle's
result for the same sample ids (the per-unit stream contract makes any column computable by itself), invariance
under splitting the sample range, and the reference's own sampler invariants (test/benchmark_gat.py:773-780,
:828-837: normalized lists inside the workspace that cover exactly the observed number of workspace bases).
Bit-exact throughout: int64 counts, IEEE doubles for the density counter, uint32 coordinates."""
import os

import numpy as np
import pytest

from gat_amd import _lib, problem, synthetic
from oracle import oracle as O

pytestmark = pytest.mark.gpu

# configuration -> (samples compared against the oracle incl. sampled lists, samples of the full-size launch,
#                   columns of that launch checked against the oracle)
CONFIGS = {
    "config2": (16, 10000, 6),      # 1 x MI355X: 10k segments x 1 track, 10 000 samples
    "config3": (16, 10000, 4),      # 1 x MI355X: 10k segments x 100 tracks, 192 isochore units, 10 000 samples
    "config4": (3, 12500, 2),       # 8 x MI355X: 100k segments x 1000 tracks; a rank's whole 12 500-sample shard, one call
    "config5": (16, 125000, 4),     # 8 x MI355X: density, 1M-interval annotation; a rank's whole 125 000-sample shard, one call
}
_CACHE = {}


@pytest.fixture(scope="module")
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def _flat(name):
    if name not in _CACHE:
        _CACHE.clear()                       # config4 alone is 80 MB of annotations: keep one at a time
        cfg = synthetic.config(name)
        _CACHE[name] = (cfg, problem.flatten_arrays(cfg["segments"], cfg["annotations"], cfg["workspace"], cfg["isochores"]))
    return _CACHE[name]


@pytest.mark.parametrize("name", list(CONFIGS))
def test_config_workloads_vs_oracle(ctx, name):
    """counts of the configuration's counter (plus the other nucleotide counter) and the sampled lists == oracle."""
    cfg, flat = _flat(name)
    S = CONFIGS[name][0]
    counters = [cfg["counter"], "nucleotide-density" if cfg["counter"] == "nucleotide-overlap" else "nucleotide-overlap"]
    seed, begin = 2024, 5
    want, wsamples = O.run_samples(flat, counters, seed, 1, begin, begin + S, want_samples=True)
    P = _lib.Problem(ctx, flat)
    try:
        got = P.sample_and_count(counters, seed, begin, begin + S)
        for k, c in enumerate(counters):
            assert got[k].dtype == want[k].dtype and np.array_equal(got[k], want[k]), (name, c)
        seg, off = P.sample(seed, begin, begin + S)
        assert np.array_equal(off, wsamples[1]), name
        assert np.array_equal(seg, wsamples[0]), name
        assert (want[0] != 0).any()              # the comparison is not about zeros
        # the kernels these shapes normally do not take: k_place with compiler-managed loads (instead of k_place_pipe), and
        # on the long lists of the config-4 shape k_sampler running the placement rounds itself (instead of k_tail_big) / finishing
        # the units itself (instead of k_resume_big)
        for knob in ("GAT_PLACE_NO_PIPE", "GAT_NO_TAIL_BIG", "GAT_NO_RESUME_BIG"):
            ctx.options[knob] = "1"
            try:
                again = P.sample_and_count(counters, seed, begin, begin + S)
            finally:
                ctx.options.pop(knob)
            for k, c in enumerate(counters):
                assert np.array_equal(again[k], want[k]), (name, c, knob)
    finally:
        P.close()


@pytest.mark.parametrize("name", list(CONFIGS))
def test_config_workloads_full_size_properties(ctx, name):
    """one launch of the size a GPU runs for the configuration: picked columns == oracle for those sample ids, the
    matrix does not depend on how the range is split over calls, and the sampled lists keep the sampler's invariants."""
    cfg, flat = _flat(name)
    _, S, ncheck = CONFIGS[name]
    counters = [cfg["counter"]]
    seed = 31337
    P = _lib.Problem(ctx, flat)
    try:
        full = P.sample_and_count(counters, seed, 0, S)[0]
        assert full.shape == (flat["n_tracks"], S)
        if name in ("config2", "config3", "config5"):
            # counts alone: no final unit lists are written, k_count_seg (config 2, 5) / k_contig (config 3) take the
            # merged lists and k_tail's records; the same matrix must come from the final lists
            assert P.last_stats["lists_from_records"] > 0, name
            ctx.options["GAT_COUNT_FINAL_LISTS"] = "1"
            ctx.options["GAT_CONTIG_FINAL_LISTS"] = "1"
            try:
                again = P.sample_and_count(counters, seed, 0, S)[0]
                assert P.last_stats["lists_from_records"] == 0
            finally:
                ctx.options.pop("GAT_COUNT_FINAL_LISTS", None)
                ctx.options.pop("GAT_CONTIG_FINAL_LISTS", None)
            assert np.array_equal(again, full), name
        cols = sorted(set([0, S - 1] + [int(x) for x in np.random.RandomState(1).randint(0, S, ncheck)]))[:max(2, ncheck)]
        for s in cols:
            want, _ = O.run_samples(flat, counters, seed, 1, s, s + 1)
            assert np.array_equal(full[:, s], want[0][:, 0]), (name, s)
        # splitting the range (what sharding over GPUs and batching rely on)
        cut = S // 3 + 1
        a = P.sample_and_count(counters, seed, 0, cut)[0]
        b = P.sample_and_count(counters, seed, cut, S)[0]
        assert np.array_equal(np.concatenate([a, b], axis=1), full), name
        # sampler invariants at unit level, last samples of the range
        n_inv = 4
        seg, off = P.sample(seed, S - n_inv, S, unit_level=True)
        U = flat["n_units"]
        for u in range(U):
            if flat["unit_contig"][u] < 0:
                continue
            us = flat["segs"][flat["seg_off"][u]:flat["seg_off"][u + 1]]
            uw = flat["ws"][flat["ws_off"][u]:flat["ws_off"][u + 1]]
            ltotal = O.total(O.intersect(O.filter(us, uw), uw))
            for i in range(n_inv if U <= 48 else 1):
                x = seg[off[i * U + u]:off[i * U + u + 1]]
                if ltotal == 0:
                    assert len(x) == 0
                    continue
                assert O.check(x), (name, u)
                assert O.total(O.intersect(x, uw)) == ltotal, (name, u)
                assert len(O.filter(x, uw)) == len(x), (name, u)
        # the counts of one sample recomputed from its sampled contig lists with the oracle's counter
        cseg, coff = P.sample(seed, S - 1, S)
        C = flat["n_contigs"]
        for t in sorted(set([0, flat["n_tracks"] - 1])):
            vals = [O.counter(cfg["counter"], cseg[coff[c]:coff[c + 1]],
                              flat["annos"][flat["anno_off"][t * C + c]:flat["anno_off"][t * C + c + 1]],
                              int(flat["cws_nseg"][c])) for c in range(C)]
            if cfg["counter"] == "nucleotide-density":
                acc = 0.0
                for v in vals:
                    acc += v
                assert full[t, S - 1] == acc
            else:
                assert full[t, S - 1] == int(sum(vals))
    finally:
        P.close()


def test_config4_shard_across_batches(ctx):
    """the config-4 shape cut into batches by the scratch budget (a rank's 12 500-sample shard takes more than one batch
    under a small budget; here 96 samples under a budget that holds fewer than 40): the matrix equals the one-batch matrix
    column for column, and the columns on both sides of the batch seams equal the oracle's."""
    cfg, flat = _flat("config4")
    counters = [cfg["counter"]]
    seed, S = 777, 96
    P = _lib.Problem(ctx, flat)
    try:
        one = P.sample_and_count(counters, seed, 0, S)[0]
        stride = P.info()["slab_segments_per_sample"]
    finally:
        P.close()
    ctx.options["GAT_SLAB_BYTES"] = str(stride * 8 * 40)      # (the slab alone: rows, records and partials make a sample larger)
    Q = _lib.Problem(ctx, flat)
    try:
        cut = Q.sample_and_count(counters, seed, 0, S)[0]
    finally:
        ctx.options.pop("GAT_SLAB_BYTES", None)
        Q.close()
    assert np.array_equal(cut, one)
    assert Q.last_stats["n_batches"] >= 3 and P.last_stats["n_batches"] == 1
    for s in (0, 39, 40, S - 1):                             # (a batch holds fewer than 40: these straddle at least one seam)
        want, _ = O.run_samples(flat, counters, seed, 1, s, s + 1)
        assert np.array_equal(cut[:, s], want[0][:, 0]), s

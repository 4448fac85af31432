"""one-off sweep of the fuzz generator of tests/test_hip_parity.py over many seeds (GPU vs oracle, bit-exact);
usage: tools/fuzz_sweep.py first_seed n_seeds [edge]"""
import importlib.util, os, sys, time
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root)
spec = importlib.util.spec_from_file_location("thp", os.path.join(root, "tests", "test_hip_parity.py"))
m = importlib.util.module_from_spec(spec)
spec.loader.exec_module(m)
from gat_amd import _lib


class MP(object):                      # minimal monkeypatch stand-in
    def setenv(self, k, v):
        os.environ[k] = v


def edge_case(ctx, seed):
    """the corners: segments longer than workspace pieces, one-base pieces, units of one to three segments, dense units
    (overshoot trims, unsuccessful rounds), odd bucket sizes"""
    import collections
    import numpy as np
    from gat_amd import problem, synthetic, intervals as iv
    from oracle import oracle as O
    rs = np.random.RandomState(seed)
    contigs = collections.OrderedDict(("e%d" % i, int(rs.randint(3000, 200000))) for i in range(int(rs.randint(1, 4))))
    segs, ws = collections.OrderedDict(), collections.OrderedDict()
    for c, size in contigs.items():
        nseg = int(rs.choice([1, 2, 3, 10, 60, 300]))
        mean = int(rs.choice([1, 5, 50, 500, 3000]))
        st = rs.randint(0, size, nseg)
        ln = 1 + rs.geometric(1.0 / mean, nseg)
        segs[c] = iv.normalize(iv.make(st, np.minimum(st + ln, size + 5000)))
        npieces = int(rs.choice([1, 2, 7, 40]))
        edges = np.sort(rs.choice(np.arange(1, size), size=min(2 * npieces, size - 1), replace=False))
        ws[c] = iv.normalize(iv.make(edges[0::2][:npieces], edges[1::2][:npieces] + int(rs.choice([0, 0, 1]))))
        ws[c] = ws[c][ws[c]["end"] > ws[c]["start"]]
    annos = [("t0", synthetic.random_segments(contigs, int(rs.randint(5, 200)), int(rs.randint(20, 2000)), int(rs.randint(1 << 30))))]
    bucket_size = int(rs.choice([0, 1, 3, 64]))
    nbuckets = int(rs.choice([100000, 5000]))
    try:
        flat = problem.flatten_arrays(segs, annos, ws, None, bucket_size=bucket_size, nbuckets=nbuckets)
    except Exception:                  # noqa: BLE001  (degenerate generator output)
        return "skipped"
    if flat["n_contigs"] == 0:
        return "skipped"
    counters = list(_lib.COUNTER_IDS.keys())
    S = 10
    try:
        want, wsamples = O.run_samples(flat, counters, seed, 1, 0, S, want_samples=True)
    except (ValueError, AssertionError) as e:
        try:
            P = _lib.Problem(ctx, flat)
            P.sample_and_count(counters, seed, 0, S)
        except type(e):
            return "both raised %s" % type(e).__name__
        raise AssertionError("oracle raised %s, the device path did not" % type(e).__name__)
    P = _lib.Problem(ctx, flat)
    got = P.sample_and_count(counters, seed, 0, S)
    for k, c in enumerate(counters):
        assert np.array_equal(got[k], want[k]), c
    seg, off = P.sample(seed, 0, S)
    assert np.array_equal(off, wsamples[1]) and np.array_equal(seg, wsamples[0])
    P.close()
    return "compared"


ctx = _lib.Context(0)
first, n = int(sys.argv[1]), int(sys.argv[2])
edge = len(sys.argv) > 3 and sys.argv[3] == "edge"
bad = 0
outcomes = {}
t0 = time.time()
for seed in range(first, first + n):
    os.environ.pop("GAT_TEST_HUGE", None)
    try:
        if edge:
            r = edge_case(ctx, seed)
            outcomes[r] = outcomes.get(r, 0) + 1
        else:
            m.test_fuzz_shapes_vs_oracle(ctx, seed, MP())
    except Exception as e:             # noqa: BLE001
        bad += 1
        print("seed %d: %s: %s" % (seed, type(e).__name__, str(e)[:300]))
print("%d seeds, %d failures, %.1f s %s" % (n, bad, time.time() - t0, outcomes if edge else ""))

"""k_place (and the other sampler kernels) against the number of samples in a call: a kernel bound by throughput scales
with it, one bound by its longest serial chain does not.  usage: tools/place_scaling.py [config] [S ...]"""
import os, sys
os.environ.setdefault("GAT_KERNEL_TIMES", "1")      # the per-kernel times of gat_stats (off by default)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gat_amd import _lib, problem, synthetic

name = sys.argv[1] if len(sys.argv) > 1 else "config2"
sizes = [int(x) for x in sys.argv[2:]] or [1250, 2500, 5000, 10000, 20000]
cfg = synthetic.config(name)
flat = problem.flatten_arrays(cfg["segments"], cfg["annotations"], cfg["workspace"], cfg["isochores"])
ctx = _lib.Context(0)
P = _lib.Problem(ctx, flat)
for S in sizes:
    P.sample_and_count([cfg["counter"]], 1, 0, S)
    best = None
    for rep in range(3):
        P.sample_and_count([cfg["counter"]], 1, 0, S)
        st = P.last_stats
        if best is None or st["ms_place"] < best["ms_place"]:
            best = st
    print("%s S=%6d  rng %.3f place %.3f consolidate %.3f behind-it %.3f count %.3f (main kernel %.3f) total %.3f ms" % (
        name, S, best["ms_rng"], best["ms_place"], best["ms_merge"], best["ms_tail"], best["ms_count"], best["ms_count_main"],
        best["ms_total"]))

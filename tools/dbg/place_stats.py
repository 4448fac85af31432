"""debug: the sampler's statistics of one call with k_place's written-out step and with the compiler's (GAT_PLACE_NO_CM=1)"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from gat_amd import _lib, problem, synthetic

name, S = sys.argv[1], int(sys.argv[2])
cfg = synthetic.config(name, 1.0)
flat = problem.flatten_arrays(cfg["segments"], cfg["annotations"], cfg["workspace"], cfg["isochores"])
ctx = _lib.Context(0)
res = {}
for env in ("0", "1"):
    if env == "1":
        os.environ["GAT_PLACE_NO_CM"] = "1"
    else:
        os.environ.pop("GAT_PLACE_NO_CM", None)
    P = _lib.Problem(ctx, flat)
    got = P.sample_and_count([cfg["counter"]], 7, 0, S)
    st = P.last_stats
    print("NO_CM=" + env, {k: st[k] for k in ("n_tail_units", "n_full_units", "n_resumed_units", "n_retried", "n_batches", "n_unsuccessful", "n_sampled_segments")})
    res[env] = got[0]
    P.close()
print("equal:", np.array_equal(res["0"], res["1"]))

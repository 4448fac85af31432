"""what crossing the boundary costs beside the timed step: gat_problem_create (host-side preparation of the unit records, length
histograms and look-up structures + upload of the inputs) and the read-back of the count matrix, per BASELINE configuration.
usage: tools/time_create.py [config ...]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from gat_amd import _lib, problem, synthetic

S = {"config2": 10000, "config3": 10000, "config5": 16384, "config4": 4096}
ctx = _lib.Context(0)
for name in sys.argv[1:] or ["config2", "config3", "config5", "config4"]:
    cfg = synthetic.config(name)
    flat = problem.flatten_arrays(cfg["segments"], cfg["annotations"], cfg["workspace"], cfg["isochores"])
    nbytes = sum(np.asarray(flat[k]).nbytes for k in ("segs", "ws", "annos"))
    _lib.Problem(ctx, flat).close()
    t = time.time(); P = _lib.Problem(ctx, flat); create = time.time() - t
    P.sample_and_count([cfg["counter"]], 1, 0, S[name])
    t = time.time(); out = P.sample_and_count([cfg["counter"]], 1, 0, S[name]); call = time.time() - t
    dev = P.last_stats["ms_total"] / 1e3
    print("%s: inputs %.1f MB, gat_problem_create %.1f ms (once per problem); one call of %d samples: %.2f ms on the stream, "
          "%.2f ms wall incl. the read-back of %d bytes -> %.0f samples/s with creation and read-back counted" %
          (name, nbytes / 1e6, create * 1e3, S[name], dev * 1e3, call * 1e3, out[0].nbytes, S[name] / (create + call)))
    P.close()

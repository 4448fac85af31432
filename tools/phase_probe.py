import ctypes, sys, numpy as np
sys.path.insert(0, ".")
import gat_amd
from gat_amd import _lib, synthetic, problem
cfg = synthetic.config(sys.argv[1] if len(sys.argv) > 1 else "config2")
flat = problem.flatten_arrays(cfg["segments"], cfg["annotations"], cfg["workspace"], cfg.get("isochores"))
ctx = _lib.Context(0)
P = _lib.Problem(ctx, flat)
S = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
for it in range(3):
    P.sample_and_count(["nucleotide-overlap"], 7, 0, S)
out = (ctypes.c_ulonglong * 16)()
_lib.lib().gat_debug_phases(out)
v = np.array(list(out)[:8], dtype=float)
names = ["resume copy", "loop: draws/placements/fast paths", "sort/insert", "merge0", "coverage", "trim", "final filter+write", "-"]
for n, x in zip(names, v):
    print("%-36s %6.2f%%" % (n, 100 * x / v.sum()))

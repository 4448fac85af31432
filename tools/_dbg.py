import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_hip_parity as m
from gat_amd import _lib
ctx = _lib.Context(0)
rs = np.random.RandomState(5)
flat = m._big_problem(rs, int(sys.argv[1]) if len(sys.argv) > 1 else 6000, 3)
res = {}
for mode in sys.argv[2:] or ["old", "new"]:
    if mode == "old": ctx.options["GAT_MERGE_OLD"] = "1"
    else: ctx.options.pop("GAT_MERGE_OLD", None)
    P = _lib.Problem(ctx, flat)
    seg, off = P.sample(99, 0, 8)
    print(mode, seg.shape, off[:4], P.last_stats.get("n_tail_units"), flush=True)
    res[mode] = (seg, off)
    P.close()
if "old" in res and "new" in res:
    print("equal", np.array_equal(res["old"][0], res["new"][0]) and np.array_equal(res["old"][1], res["new"][1]))

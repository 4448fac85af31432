import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from gat_amd import _lib, problem, synthetic
cfg = synthetic.config("config2")
flat = problem.flatten_arrays(cfg["segments"], cfg["annotations"], cfg["workspace"], cfg["isochores"])
ctx = _lib.Context(0)
P = _lib.Problem(ctx, flat)
n = 1 << 20
dev = ctx.alloc(n * 8)
host = np.empty(n, np.int64)
def lap(label, f):
    t0 = time.perf_counter(); f(); print("%-50s %.2f ms" % (label, 1e3 * (time.perf_counter() - t0)), flush=True)
for rep in range(2):
    lap("sample_and_count_device 2000", lambda: P.sample_and_count_device(["nucleotide-overlap"], 1, 0, 2000, dev))
    lap("d2h right behind it", lambda: ctx.d2h(host, dev))
    lap("d2h again", lambda: ctx.d2h(host, dev))
    time.sleep(0.05)
    lap("d2h after 50 ms idle", lambda: ctx.d2h(host, dev))
    time.sleep(0.5)
    lap("d2h after 500 ms idle", lambda: ctx.d2h(host, dev))
    lap("sample_and_count_device 2000 after idle", lambda: P.sample_and_count_device(["nucleotide-overlap"], 1, 0, 2000, dev))
    lap("sync", lambda: ctx.synchronize())
    big = ctx.alloc(64 << 20)
    lap("alloc+free 64MB", lambda: ctx.free(ctx.alloc(64 << 20)))
    ctx.free(big)

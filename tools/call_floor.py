"""the host side of one gat_sample_and_count call: wall time per call of a tiny problem (config 1's shape, 64 samples: some
0.15 ms of kernels), with and without the per-kernel events; usage: tools/call_floor.py [calls]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from gat_amd import _lib, problem, synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cfg = synthetic.config("config1")
flat = problem.flatten_arrays(cfg["segments"], cfg["annotations"], cfg["workspace"], cfg["isochores"])
ctx = _lib.Context(0)
P = _lib.Problem(ctx, flat)
S = 64
dev = ctx.alloc(flat["n_tracks"] * S * 8)
for times in (False, True):
    ctx.set_kernel_times(times)
    for _ in range(50):
        P.sample_and_count_device([cfg["counter"]], 1, 0, S, dev)
    t0 = time.perf_counter()
    tot = 0.0
    for i in range(n):
        st = P.sample_and_count_device([cfg["counter"]], 1, i * S, (i + 1) * S, dev)
        tot += st["ms_total"]
    dt = time.perf_counter() - t0
    print("per-kernel events %-5s: %.1f us per call, %.1f us of it between the call's two events on the stream" %
          (times, dt / n * 1e6, tot / n * 1e3))
ctx.free(dev)
P.close()

"""does the step gain from two half-batches on two streams?  Two contexts (private streams), two problems, two host threads,
half the samples each, against one call with all of them.  usage: tools/two_streams.py [config] [samples]"""
import os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from gat_amd import _lib, problem, synthetic
name = sys.argv[1] if len(sys.argv) > 1 else "config2"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
cfg = synthetic.config(name)
flat = problem.flatten_arrays(cfg["segments"], cfg["annotations"], cfg["workspace"], cfg["isochores"])
counters = [cfg["counter"]]
ctxs = [_lib.Context(0), _lib.Context(0)]
Ps = [_lib.Problem(c, flat) for c in ctxs]
devs = [c.alloc(flat["n_tracks"] * S * 8) for c in ctxs]
def one(k, b, e):
    Ps[k].sample_and_count_device(counters, 1, b, e, devs[k])
for rep in range(3):
    one(0, 0, S); one(1, 0, S // 2)
def timeit(f, n=20):
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0) / n * 1e3
def both(parts):
    ts = [threading.Thread(target=one, args=(k, k * S // parts, (k + 1) * S // parts)) for k in range(parts)]
    [t.start() for t in ts]; [t.join() for t in ts]
print("%s %d samples: one call %.3f ms; two halves one after the other %.3f ms; two halves on two streams at once %.3f ms" % (
    name, S, timeit(lambda: one(0, 0, S)), timeit(lambda: (one(0, 0, S // 2), one(0, S // 2, S))), timeit(lambda: both(2))))

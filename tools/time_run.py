"""gat_amd.run() end to end on a BASELINE configuration (the whole drop-in: host classes, observed counts, sampling +
counting on the device, null-distribution statistics, result rows), wall clock with the stages of the host side.
usage: tools/time_run.py [config] [num_samples] [--profile]"""
import cProfile, gc, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gat_amd
from gat_amd import synthetic

args = [a for a in sys.argv[1:] if not a.startswith("--")]
name = args[0] if len(args) > 0 else "config2"
S = int(args[1]) if len(args) > 1 else 10000
cfg = synthetic.config(name)
counters = [gat_amd.COUNTERS[cfg["counter"]]()]


def once(inputs=None):
    t0 = time.perf_counter()
    segments, annotations, workspace, t_iso = inputs or synthetic.as_collections(cfg)
    t1 = time.perf_counter()
    results = gat_amd.run(segments, annotations, workspace, gat_amd.SamplerAnnotator(bucket_size=1, nbuckets=100000),
                          counters, gat_amd.UnconditionalWorkspace(), num_samples=S, random_seed=1)
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1, len(results)


print("first call (context, library load)", once())
gc.disable()
for _ in range(3):
    prep, run, n = once()
    print("%s S=%d: inputs %.1f ms, run() %.1f ms (%d result rows) -> %.0f samples/s end to end" % (name, S, 1e3 * prep, 1e3 * run, n, S / run))
inputs = synthetic.as_collections(cfg)
for _ in range(3):
    prep, run, n = once(inputs)
    print("%s S=%d: the same collections again: run() %.1f ms" % (name, S, 1e3 * run))
if "--profile" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    once(inputs)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)

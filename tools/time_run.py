"""gat_amd.run() end to end on a BASELINE configuration (the whole drop-in: host classes, observed counts, sampling +
counting on the device, null-distribution statistics, result rows), wall clock with the stages of the host side.
usage: tools/time_run.py [config] [num_samples]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gat_amd
from gat_amd import synthetic

name = sys.argv[1] if len(sys.argv) > 1 else "config2"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
cfg = synthetic.config(name)


def coll(tracks):
    c = gat_amd.IntervalCollection()
    for t, per in tracks:
        for contig, a in per.items():
            s = gat_amd.SegmentList(array=a)
            s.isNormalized = 1
            c.add(t, contig, s)
    return c


def once():
    t0 = time.time()
    segments = coll([("merged", cfg["segments"])])
    annotations = coll(cfg["annotations"])
    workspaces = coll([("ws", cfg["workspace"])])
    workspaces.collapse()
    workspaces.restrict("collapsed")
    if cfg["isochores"]:
        isochores = coll(list(cfg["isochores"].items()))
        isochores.intersect(workspaces["collapsed"])
        workspaces.toIsochores(isochores, truncate=True)
        annotations.toIsochores(isochores, truncate=True)
        segments.toIsochores(isochores, truncate=False)
    t1 = time.time()
    counters = [gat_amd.COUNTERS[cfg["counter"]]()]
    results = gat_amd.run(segments, annotations, workspaces["collapsed"], gat_amd.SamplerAnnotator(bucket_size=1, nbuckets=100000),
                          counters, gat_amd.UnconditionalWorkspace(), num_samples=S, random_seed=1)
    t2 = time.time()
    n = len(results)
    return t1 - t0, t2 - t1, n


print("first call (context, library load)", once())
for _ in range(2):
    prep, run, n = once()
    print("%s S=%d: inputs %.3f s, run() %.3f s (%d result rows) -> %.0f samples/s end to end" % (name, S, prep, run, n, S / run))
pr = cProfile.Profile()
pr.enable()
once()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

"""wall time of the whole drop-in call gat.run() (host flattening, observed counts, device sampling + counting,
read-back, statistics, table) on a BASELINE configuration; usage: tools/bench_run.py [config] [num_samples]"""
import io, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import gat_amd as gat
from gat_amd import synthetic, IO

name = sys.argv[1] if len(sys.argv) > 1 else "config3"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
cfg = synthetic.config(name)
t0 = time.time()
segments = gat.IntervalCollection("segments")
for c, a in cfg["segments"].items():
    segments.add("merged", c, gat.SegmentList(array=a, normalize=True))
annotations = gat.IntervalCollection("annotations")
for t, per in cfg["annotations"]:
    for c, a in per.items():
        annotations.add(t, c, gat.SegmentList(array=a, normalize=True))
workspaces = gat.IntervalCollection("workspace")
for c, a in cfg["workspace"].items():
    workspaces.add("collapsed", c, gat.SegmentList(array=a, normalize=True))
isochores = None
if cfg.get("isochores"):
    isochores = gat.IntervalCollection("isochores")
    for t, per in cfg["isochores"].items():
        for c, a in per.items():
            isochores.add(t, c, gat.SegmentList(array=a, normalize=True))
opts, _ = gat.buildParser().parse_args([])
workspace = IO.applyIsochores(segments, annotations, workspaces, opts, isochores)
t1 = time.time()
gat.get_context(0)
t2 = time.time()
counters = [gat.COUNTERS[cfg["counter"]]()]
for rep in range(2):
    t3 = time.time()
    results = gat.run(segments, annotations, workspace, gat.SamplerAnnotator(bucket_size=0, nbuckets=100000), counters,
                      gat.UnconditionalWorkspace(), num_samples=S, random_seed=7)
    t4 = time.time()
    opts.stdout = io.StringIO()
    IO.outputResults(results, opts, gat.AnnotatorResultExtended.headers)
    t5 = time.time()
    print("%s, %d samples, %d rows: build collections %.2f s, context %.2f s, gat.run %.3f s (%.0f samples/s end to end), table %.3f s"
          % (name, S, len(results), t1 - t0, t2 - t1, t4 - t3, S / (t4 - t3), t5 - t4))
if os.environ.get("GAT_PROFILE_RUN"):
    import cProfile, pstats
    pr = cProfile.Profile()
    pr.enable()
    gat.run(segments, annotations, workspace, gat.SamplerAnnotator(bucket_size=0, nbuckets=100000), counters,
            gat.UnconditionalWorkspace(), num_samples=S, random_seed=7)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)

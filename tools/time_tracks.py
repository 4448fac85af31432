"""gat_amd.run() over several segment tracks against config 3's annotations: wall clock per run (the annotation tables are made
once, a track's samples are enqueued while the previous track's rows are made).  usage: tools/time_tracks.py [tracks] [repeats]"""
import gc, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gat_amd
from gat_amd import synthetic

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = synthetic.config("config3")
cfg = dict(cfg, segment_tracks=[("track%02d" % i, synthetic.random_segments(synthetic.HG19, 10000, 500, 11 + i)) for i in range(n)])
segments, annotations, workspace, _ = synthetic.as_collections(cfg)
counters = [gat_amd.COUNTERS[cfg["counter"]]()]
for r in range(reps):
    t = time.perf_counter()
    rows = gat_amd.run(segments, annotations, workspace, gat_amd.SamplerAnnotator(bucket_size=1, nbuckets=100000), counters,
                       gat_amd.UnconditionalWorkspace(), num_samples=10000, random_seed=1)
    dt = time.perf_counter() - t
    print("%d tracks: run() %.1f ms, %.2f ms per track, %d rows" % (n, dt * 1e3, dt * 1e3 / n, len(rows)), flush=True)

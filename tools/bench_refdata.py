"""samples/s on the reference's own test dataset (tests/golden/refdata), per segment track."""
import os, sys, time
os.environ.setdefault("GAT_KERNEL_TIMES", "1")      # the per-kernel times of gat_stats (off by default)
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import gat_amd
from gat_amd import IO, _lib, problem

d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "refdata")
opts, _ = gat_amd.buildParser().parse_args(["--segments=%s" % os.path.join(d, "segments_single.bed.gz"),
                                           "--annotations=%s" % os.path.join(d, "annotations.bed.gz"),
                                           "--workspace=%s" % os.path.join(d, "workspace.bed.gz"), "--with-segment-tracks"])
t0 = time.time()
segments, annotations, workspaces, isochores = IO.buildSegments(opts)
workspace = IO.applyIsochores(segments, annotations, workspaces, opts, isochores)
print("load+prepare %.2f s" % (time.time() - t0))
ctx = gat_amd.get_context()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
for track in segments.tracks:
    flat = problem.flatten_units(segments[track].asArrays(), workspace.asArrays(),
                                 [(t, annotations[t].asArrays()) for t in annotations.tracks])
    P = _lib.Problem(ctx, flat)
    P.sample_and_count(["nucleotide-overlap"], 1, 0, 64)
    t0 = time.time()
    P.sample_and_count(["nucleotide-overlap"], 1, 0, S)
    dt = time.time() - t0
    st = P.last_stats
    print("%s: %d segments, %d units, %d ws segs: %.0f samples/s  (sampler %.1f ms, count %.1f ms, full units %d)" % (
        track, len(flat["segs"]), flat["n_units"], len(flat["ws"]), S / dt, st["ms_sampler"], st["ms_count"], st["n_full_units"]))
    P.close()

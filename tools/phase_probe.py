"""temporary: k_sampler phase shares (needs the instrumented build, /tmp/instrument.py)"""
import ctypes, os, sys, numpy as np
sys.path.insert(0, ".")
import gat_amd
from gat_amd import _lib, synthetic, problem, IO
which = sys.argv[1] if len(sys.argv) > 1 else "config2"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
ctx = _lib.Context(0)
if which == "refdata":
    d = os.path.join("tests", "golden", "refdata")
    opts, _ = gat_amd.buildParser().parse_args(["--segments=%s" % os.path.join(d, "segments_single.bed.gz"),
        "--annotations=%s" % os.path.join(d, "annotations.bed.gz"), "--workspace=%s" % os.path.join(d, "workspace.bed.gz"), "--with-segment-tracks"])
    segments, annotations, workspaces, isochores = IO.buildSegments(opts)
    workspace = IO.applyIsochores(segments, annotations, workspaces, opts, isochores)
    track = list(segments.tracks)[int(sys.argv[3]) if len(sys.argv) > 3 else 3]
    flat = problem.flatten_units(segments[track].asArrays(), workspace.asArrays(), [(t, annotations[t].asArrays()) for t in annotations.tracks])
else:
    cfg = synthetic.config(which)
    flat = problem.flatten_arrays(cfg["segments"], cfg["annotations"], cfg["workspace"], cfg.get("isochores"))
P = _lib.Problem(ctx, flat)
for it in range(2):
    P.sample_and_count(["nucleotide-overlap"], 7, 0, S)
out = (ctypes.c_ulonglong * 16)()
_lib.lib().gat_debug_phases(out)
v = np.array(list(out)[:10], dtype=float)
names = ["resume copy", "loop: draws/placements", "sort/insert", "merge0", "coverage", "trim", "final filter+write", "fast-path consolidations", "dirty compaction", "big counting sort"]
for n, x in zip(names, v):
    print("%-36s %6.2f%%" % (n, 100 * x / v.sum()))
print("consolidations per unit-sample: %.2f" % (out[8] / (2.0 * S * flat["n_units"])))
print("raw:", [int(x) for x in list(out)[:10]])

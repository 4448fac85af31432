"""gat_amd.run() on a BASELINE configuration under cProfile: where the host side of the seam spends its time.
usage (GPU box): python tools/profile_run.py [config3] [repeats]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import gat_amd
from gat_amd import synthetic

name = sys.argv[1] if len(sys.argv) > 1 else "config3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = synthetic.config(name)
segments, annotations, workspace, _ = synthetic.as_collections(cfg)
counters = [gat_amd.COUNTERS[cfg["counter"]]()]


def call():
    t = time.perf_counter()
    gat_amd.run(segments, annotations, workspace, gat_amd.SamplerAnnotator(bucket_size=1, nbuckets=100000), counters,
                gat_amd.UnconditionalWorkspace(), num_samples=10000, random_seed=7)
    return time.perf_counter() - t


call(); call()
print("ms per run:", ["%.2f" % (call() * 1e3) for _ in range(reps)])
pr = cProfile.Profile()
pr.enable()
for _ in range(reps):
    call()
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(22)
print(out.getvalue()[:6000])

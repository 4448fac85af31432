"""scripts/gat-run.py end to end on the reference's own integration-test data (tests/golden/refdata: 279 844 workspace
segments, 4 segment x 7 annotation tracks), wall clock and where the host side spends it.
usage: tools/time_cli.py [num_samples]"""
import cProfile, io, os, pstats, sys, time
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root)
import importlib.util
spec = importlib.util.spec_from_file_location("gat_run", os.path.join(root, "scripts", "gat-run.py"))
gat_run = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gat_run)

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
d = os.path.join(root, "tests", "golden", "refdata")
argv = ["gat-run.py", "--segments=%s" % os.path.join(d, "segments_single.bed.gz"),
        "--annotations=%s" % os.path.join(d, "annotations.bed.gz"), "--workspace=%s" % os.path.join(d, "workspace.bed.gz"),
        "--num-samples=%d" % S, "--random-seed=1", "--with-segment-tracks", "--log=/dev/null", "--stdout=/dev/null"]


def once():
    t0 = time.time()
    gat_run.main(list(argv))
    return time.time() - t0


print("first run (library load, context): %.2f s" % once())
print("second run: %.2f s for %d samples" % (once(), S))
pr = cProfile.Profile()
pr.enable()
once()
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(30)
print(out.getvalue()[:6000])

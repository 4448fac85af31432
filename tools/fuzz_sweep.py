"""one-off sweep of the fuzz generator of tests/test_hip_parity.py over many seeds (GPU vs oracle, bit-exact);
usage: tools/fuzz_sweep.py first_seed n_seeds [edge|merged|long|scan|frag|units]"""
import importlib.util, os, sys, time
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, root)
spec = importlib.util.spec_from_file_location("thp", os.path.join(root, "tests", "test_hip_parity.py"))
m = importlib.util.module_from_spec(spec)
spec.loader.exec_module(m)
from gat_amd import _lib


class MP(object):                      # minimal monkeypatch stand-in: the knobs go through the context's options
    def setenv(self, k, v):
        ctx.options[k] = v

    def setitem(self, d, k, v):
        d[k] = v

    def delenv(self, k):
        ctx.options.pop(k, None)

    def delitem(self, d, k):
        d.pop(k, None)


ctx = _lib.Context(0)
first, n = int(sys.argv[1]), int(sys.argv[2])
edge = len(sys.argv) > 3 and sys.argv[3] == "edge"
merged = len(sys.argv) > 3 and sys.argv[3] == "merged"
long_lists = len(sys.argv) > 3 and sys.argv[3] == "long"
scan = len(sys.argv) > 3 and sys.argv[3] == "scan"
frag = len(sys.argv) > 3 and sys.argv[3] == "frag"
units = len(sys.argv) > 3 and sys.argv[3] == "units"
uo = [0, 0, 0]
handed = 0
bad = 0
outcomes = {}
t0 = time.time()
for seed in range(first, first + n):
    for k in ("GAT_TEST_HUGE", "GAT_PLACE_NO_WIDE", "GAT_PLACE_NO_CM", "GAT_PLACE_SCAN_SEQ", "GAT_PLACE_SCAN_TILES", "GAT_GRID_CELL_SEGS",
              "GAT_PLACE_NO_GRID", "GAT_TAIL_NO_LONG_WS", "GAT_COUNT_VIA_CONTIGS", "GAT_MERGED_MIN_TRACKS"):       # (what a seed's test sets stays set here: MP does not undo)
        ctx.options.pop(k, None)
    if (merged or long_lists or edge) and seed % 4 >= 2:
        ctx.options["GAT_PLACE_NO_CM"] = "1"           # k_place's steps as the compiler writes them (the shapes test picks by itself)
    try:
        if units:
            r = m._units_direct_case(ctx, seed)
            uo = [uo[0] + r[0], uo[1] + r[1], uo[2] + (1 if r[2] else 0)]
        elif frag:
            handed += m._frag_ws_case(ctx, seed)
        elif scan:
            m._scan_case(ctx, seed, MP().setenv)
        elif merged:
            for k in ("GAT_MERGED_MIN_TRACKS", "GAT_COUNT_NO_MERGED", "GAT_MERGED_BLOCK"):
                ctx.options.pop(k, None)
            m.test_merged_track_index_vs_oracle(ctx, seed, MP())
        elif long_lists:
            handed += m._long_list_case(ctx, seed)
        elif edge:
            r = m._edge_case(ctx, seed)
            outcomes[r] = outcomes.get(r, 0) + 1
        else:
            m.test_fuzz_shapes_vs_oracle(ctx, seed, MP())
    except Exception as e:             # noqa: BLE001
        bad += 1
        print("seed %d: %s: %s" % (seed, type(e).__name__, str(e)[:300]), flush=True)
    if (seed - first + 1) % 2000 == 0:  # (a sweep cut short by `timeout` still says how far it came)
        print("... %d seeds, %d failures, %.0f s" % (seed - first + 1, bad, time.time() - t0), flush=True)
print("%d seeds, %d failures, %.1f s %s" % (n, bad, time.time() - t0, outcomes if edge else ("units through k_tail_big: %d" % handed if long_lists else ("units finished by k_tail: %d" % handed if frag else ("candidates %d, overlaps taken off %d, problems repeated through k_contig %d" % tuple(uo) if units else "")))))

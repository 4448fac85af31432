"""k_place's tiles in time (diagnostic build, tools/diag_place.sh writes phases.jsonl): per unit -- largest first -- when its tiles began
and ended, microseconds from the kernel's first begin.  usage: tools/diag_tiles.py gpurun_out/diag_<tag>/phases.jsonl"""
import json, sys
last = None
for l in open(sys.argv[1]):
    d = json.loads(l)
    if "k_place_tiles" in d:
        last = d["k_place_tiles"]
if last is None:
    sys.exit("no k_place_tiles record")
print("%3s %8s %6s | begin: first median last | end: first median last   (us)" % ("a", "segments", "tiles"))
for u in last:
    print("%3d %8d %6d | %7.1f %7.1f %7.1f | %7.1f %7.1f %7.1f" % (u["a"], u["segments"], u["tiles"], u["begin_first"], u["begin_median"],
                                                                  u["begin_last"], u["end_first"], u["end_median"], u["end_last"]))
print("kernel: %.1f us from the first begin to the last end" % max(u["end_last"] for u in last))

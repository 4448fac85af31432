#!/bin/bash
# Diagnostic build of the library with in-kernel stamps (-DGAT_DIAG): where k_place's loop spends a wave's cycles -- waiting
# for its rows, the chunk's look-ups, the eight steps of the state machine, the flush of the ring, loop control -- over all
# tiles and for the largest unit's tiles (the ones the kernel ends with).  s_memtime counts shader-clock cycles; the stamps
# fence the schedule (every stamp waits for the scalar and LDS queues): read the SHARES and the per-row figures as bounds.
# usage (GPU box): bash tools/diag_place.sh <tag> [config:samples ...]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
TAG=$1; shift
SHAPES=${@:-"config2:10000 config2:1250 config3:10000"}
OUT=$R/gpurun_out/diag_$TAG; mkdir -p $OUT
LIB=$R/build/diag/libgat_mi355_diag.so   # (built in the build container when it is there: make ... EXTRA=-DGAT_DIAG BUILD=build/diag OUT=build/diag/libgat_mi355_diag.so)
[ -f $LIB ] || make -C gat_amd/csrc -s -j2 EXTRA=-DGAT_DIAG BUILD=$R/build/diag OUT=$LIB || exit 1
for SH in $SHAPES; do
  CFG=${SH%%:*}; S=${SH##*:}
  rm -f $OUT/phases.jsonl
  GAT_LIB_PATH=$LIB GAT_DIAG_OUT=$OUT/phases.jsonl python3 bench.py --no-cpu-baseline --no-api --no-strong --sustain-seconds 0 \
      --extra "" --config $CFG --samples $S --steps 2 --warmup 1 > $OUT/bench_${CFG}_${S}.log 2>&1
  python3 - $OUT/phases.jsonl $CFG $S <<'PY'
import json, sys
names = ["row_wait", "lookups", "steps", "flush", "loop_control"]
tot = {"all_tiles": {}, "largest_unit": {}}
for l in open(sys.argv[1]):
    d = json.loads(l)
    if "k_place" not in d: continue
    for which in tot:
        for k, v in d["k_place"][which].items(): tot[which][k] = tot[which].get(k, 0) + v
print("k_place, %s at %s samples per call (cycles)" % (sys.argv[2], sys.argv[3]))
for which in ("all_tiles", "largest_unit"):
    t = tot[which]; s = sum(t[n] for n in names)
    print("  %-13s %d tiles, %.0f rows per tile, %.0f cycles per row of 64 lanes" % (which, t["tiles"], t["rows"] / max(1, t["tiles"]), s / max(1, t["rows"])))
    for n in names: print("      %-13s %5.1f %%   %6.1f cycles per row" % (n, 100.0 * t[n] / s, t[n] / max(1, t["rows"])))
PY
done

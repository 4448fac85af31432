#!/bin/bash
# Diagnostic build (-DGAT_DIAG_CONS): where a k_consolidate work unit (config4: a k_merge_big workgroup) spends its cycles -- unit record + workspace,
# the list into registers, the sort, merge(0), coverage + write-back.  Every stamp drains the wave's memory queues (the shares
# are of a unit walked phase by phase: bounds, not the product kernel's timing).
# usage (GPU box): bash tools/diag_consolidate.sh <tag> [config:samples ...]   (the library is built beforehand where hipcc is:
#   make -C gat_amd/csrc EXTRA="-DGAT_DIAG_CONS" BUILD=build/diagc OUT=build/diagc/libgat_mi355_diagc.so)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
TAG=$1; shift
SHAPES=${@:-"config2:10000 config3:10000"}
OUT=$R/gpurun_out/diagc_$TAG; mkdir -p $OUT
LIB=$R/build/diagc/libgat_mi355_diagc.so
[ -f $LIB ] || make -C gat_amd/csrc -s -j2 EXTRA="-DGAT_DIAG_CONS" BUILD=$R/build/diagc OUT=$LIB || exit 1
for SH in $SHAPES; do
  CFG=${SH%%:*}; S=${SH##*:}
  rm -f $OUT/phases.jsonl
  GAT_LIB_PATH=$LIB GAT_DIAG_OUT=$OUT/phases.jsonl python3 bench.py --no-cpu-baseline --no-api --no-strong --sustain-seconds 0 \
      --extra "" --config $CFG --samples $S --steps 2 --warmup 1 --details /dev/null > $OUT/bench_${CFG}_${S}.log 2>&1
  python3 - $OUT/phases.jsonl $CFG $S <<'PY'
import json, sys
names = [("prologue", "unit record + workspace"), ("sort", "list into registers"), ("merge", "sort"), ("coverage", "merge(0)"),
         ("fast_paths", "coverage + write-back")]
kernel = "k_consolidate"
if sys.argv[2] == "config4":       # long lists: k_merge_big's stamps land in the same slots (a workgroup per unit)
    kernel = "k_merge_big"
    names = [("prologue", "record + workspace + zeroed histogram"), ("sort", "histogram pass over the slab"), ("merge", "prefix over the buckets"),
             ("coverage", "scatter pass over the slab into LDS"), ("fast_paths", "buckets sorted thread by thread"), ("trim", "merge(0)"),
             ("draws_placement", "coverage"), ("final_filter_write", "running lengths + record")]
tot, wu = {}, 0
for l in open(sys.argv[1]):
    d = json.loads(l)
    if "cycles" not in d: continue
    wu += d["work_units"]
    for k, v in d["cycles"].items(): tot[k] = tot.get(k, 0) + v
s = sum(tot.get(k, 0) for k, _ in names)
print("%s, %s at %s samples per call: %.0f cycles per work unit" % (kernel, sys.argv[2], sys.argv[3], s / max(1, wu)))
for k, what in names: print("  %-26s %5.1f %%  %8.0f cycles/unit" % (what, 100 * tot.get(k, 0) / max(1, s), tot.get(k, 0) / max(1, wu)))
PY
done

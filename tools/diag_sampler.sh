#!/bin/bash
# Diagnostic build of the library with in-kernel stamps (-DGAT_DIAG): where a k_sampler work unit spends its cycles
# (prologue / sort / merge / coverage / fast paths / trim / draws / final filter).  The stamps fence the schedule: read the
# SHARES, not the run time.  usage (GPU box): bash tools/diag_sampler.sh <tag> [bench args]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
TAG=$1; shift
OUT=$R/gpurun_out/diag_$TAG; mkdir -p $OUT
make -C gat_amd/csrc -s -j2 EXTRA=-DGAT_DIAG BUILD=$OUT/build OUT=$OUT/libgat_mi355_diag.so || exit 1
rm -f $OUT/phases.jsonl
GAT_LIB_PATH=$OUT/libgat_mi355_diag.so GAT_DIAG_OUT=$OUT/phases.jsonl python3 bench.py --no-cpu-baseline --extra "" --steps 2 --warmup 1 "$@" > $OUT/bench.log 2>&1
python3 - $OUT/phases.jsonl <<'PY'
import json, sys
tot = {}; wu = 0
for l in open(sys.argv[1]):
    d = json.loads(l); wu += d["work_units"]
    for k, v in d["cycles"].items(): tot[k] = tot.get(k, 0) + v
s = sum(tot.values())
print("k_sampler phases, %d work units, %.0f cycles per work unit" % (wu, s / max(1, wu)))
for k, v in tot.items(): print("  %-20s %5.1f %%  %8.0f cycles/unit" % (k, 100 * v / s, v / max(1, wu)))
PY

#!/bin/bash
# per-kernel PMC counters of one bench command under an environment tag; usage: tools/pmc2.sh <tag> "<bench args>" GROUP...
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
TAG=$1; ARGS=$2; shift; shift
OUT=$R/gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
i=0
for G in "$@"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $G --output-format csv -d $OUT/g$i -- python3 bench.py --no-cpu-baseline $ARGS > $OUT/log$i 2>&1
done
python3 - $OUT <<'PY' > $OUT/summary.txt
import csv, glob, sys, collections
res = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1] + "/g*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "gat::" not in n: continue
        res[n][r["Counter_Name"]] += float(r["Counter_Value"]); calls[n][r["Counter_Name"]] += 1
for n in res:
    print(n[:90])
    for c in sorted(res[n]): print("   %-28s %16.0f per launch" % (c, res[n][c] / max(1, calls[n][c])))
PY
cat $OUT/summary.txt

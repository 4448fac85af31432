#!/bin/bash
# per-kernel average durations of one bench run (rocprofv3 kernel trace); usage: tools/kstats.sh [bench args]
# KSTATS_SCRIPT=tools/bench_refdata.py tools/kstats.sh 10000   profiles another script instead of bench.py
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/kstats; rm -rf $OUT; mkdir -p $OUT
if [ -n "${KSTATS_SCRIPT:-}" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $KSTATS_SCRIPT "$@" > $OUT/log 2>&1
else
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --no-cpu-baseline "$@" > $OUT/log 2>&1
fi
python3 - $OUT <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "gat::" in r["Name"]:
            print("%-60s calls %3s avg %9.1f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY

#!/bin/bash
# rocprofv3 summaries behind DESIGN.md / bench.py's roofline: kernel-trace stats, then HBM traffic
# counters in separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).
# usage (on the GPU box): bash tools/collect_profiles.sh <tag> [bench args...]
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline "$@" > $OUT/bench_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --no-cpu-baseline "$@" > $OUT/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --no-cpu-baseline "$@" > $OUT/bench_write.log 2>&1
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(int)
for kind in ("fetch", "write"):
    for f in glob.glob(out + "/%s/*/*counter_collection.csv" % kind):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "gat::" not in n:
                continue
            res[n][r["Counter_Name"]] += float(r["Counter_Value"])
            if kind == "fetch":
                calls[n] += 1
summary = {}
for n, d in res.items():
    c = max(1, calls[n])
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB
    summary[n] = {"launches": c, "FETCH_SIZE_KiB_per_launch": d.get("FETCH_SIZE", 0.0) / c,
                  "WRITE_SIZE_KiB_per_launch": d.get("WRITE_SIZE", 0.0) / c}
json.dump(summary, open(out + "/traffic.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
grep -v "^W\|^E\|^I" $OUT/bench_trace.log | tail -1 > $OUT/bench.json

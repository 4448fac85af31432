#!/bin/bash
# A/B timing of library builds made beforehand (build/v/<name>.so travel with the snapshot): per build and shape one short
# bench.py run, one line of per-kernel times each.  usage (GPU box): bash tools/ab.sh "config2:10000 config3:10000" build/v/a.so build/v/b.so ...
# env AB_ENV="GAT_GRID_FACTOR=4" adds environment to every run; AB_STEPS (default 10)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/ab; mkdir -p $OUT
SHAPES=$1; shift
for rep in 1 2; do
for LIB in "$@"; do
  for SH in $SHAPES; do
    CFG=${SH%%:*}; S=${SH##*:}
    TAG=$(basename $LIB .so)_${CFG}_${S}
    env ${AB_ENV:-} GAT_LIB_PATH=$R/$LIB python3 bench.py --no-cpu-baseline --no-api --no-strong --sustain-seconds 0 --extra "" \
        --config $CFG --samples $S --steps ${AB_STEPS:-10} --warmup 3 --details $OUT/$TAG.json > /dev/null 2> $OUT/$TAG.err || { echo "FAILED $TAG"; tail -3 $OUT/$TAG.err; }
    echo -n "$(basename $LIB .so) ${AB_ENV:-} | "; python3 tools/show_bench.py $OUT/$TAG.json
  done
done
done

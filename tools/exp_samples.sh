#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
OUT=$R/gpurun_out/exp20; mkdir -p $OUT
i=0
for flags in "$@"; do
  i=$((i+1))
  make -C gat_amd/csrc -s -j2 EXTRA="$flags" BUILD=$OUT/build_$i OUT=$OUT/lib_$i.so || exit 1
  for S in 10000 20000; do
  GAT_LIB_PATH=$OUT/lib_$i.so python3 bench.py --no-cpu-baseline --no-api --no-strong --sustain-seconds 0 --extra "" --config config2 --samples $S --steps 10 --warmup 2 > $OUT/bench_$i.json 2>$OUT/err_$i.log
  echo "== $flags S=$S"; python3 tools/show_bench.py $OUT/bench_$i.json
  done
done

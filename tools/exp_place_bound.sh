#!/bin/bash
# What bounds k_place_pipe at 10 000 samples of config 2 -- its bytes or its instructions?  Three builds of
# the library, timed back to back on one box (tools/ab.sh): the product; GAT_EXP_SAME_ROWS -- every tile reads the SAME 64 rows
# (16 KB, cache resident: the kernel without the 2.0 GB of row traffic); GAT_EXP_NO_STORES -- the flush's stores left out (0.8 GB).
# The timing builds compute garbage; only k_place's time is read.
# usage (build container): bash tools/exp_place_bound.sh build ; (GPU box) bash tools/exp_place_bound.sh run
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
if [ "${1:-run}" = build ]; then
  mkdir -p build/v
  make -C gat_amd/csrc -s BUILD=$R/build/pb0 OUT=$R/build/v/place_product.so &
  make -C gat_amd/csrc -s EXTRA=-DGAT_EXP_SAME_ROWS BUILD=$R/build/pb1 OUT=$R/build/v/place_same_rows.so &
  make -C gat_amd/csrc -s EXTRA=-DGAT_EXP_NO_STORES BUILD=$R/build/pb2 OUT=$R/build/v/place_no_stores.so &
  make -C gat_amd/csrc -s EXTRA="-DGAT_EXP_SAME_ROWS -DGAT_EXP_NO_STORES" BUILD=$R/build/pb3 OUT=$R/build/v/place_neither.so &
  wait
  exit 0
fi
for S in 10000 20000 2500; do
  bash tools/ab.sh "config2:$S" build/v/place_product.so build/v/place_same_rows.so build/v/place_no_stores.so build/v/place_neither.so
done

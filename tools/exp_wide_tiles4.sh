# k_place_wide's tiles per workgroup on the config-4 shape; tuning builds as for tools/exp_wide_tiles.sh
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for W in default w4 w2; do
  echo "config-4 shape, k_place_wide tiles per workgroup: $W"
  if [ $W = default ]; then L=""; else L="GAT_LIB_PATH=$PWD/build/$W/libgat_$W.so"; fi
  env $L python bench.py --config config4 --samples 12500 --steps 4 --warmup 1 --no-api --no-strong --no-cpu-baseline --extra "" --sustain-seconds 0 2>/dev/null | python tools/show_bench.py /dev/stdin
done

# the merged index's piece bound (GAT_MERGED_BOUND x the mean interval length) on the config-4 shape
for B in 1 2 3; do
  echo "GAT_MERGED_BOUND=$B"
  GAT_MERGED_BOUND=$B python bench.py --config config4 --samples 12500 --steps 3 --warmup 1 --no-api --no-strong --no-cpu-baseline --extra "" --sustain-seconds 0 2>/dev/null | python tools/show_bench.py /dev/stdin
done

#!/bin/bash
# the row budget of k_rng (rows generated per stream = expected consumption + sigmas x spread + tail rows) against what the
# streams that run out cost: k_rng / k_sampler times and units run in full, per setting.  usage: bash tools/exp_rows.sh
run() {
  for CFG in config2:10000 config3:10000; do
    python bench.py --config ${CFG%%:*} --samples ${CFG##*:} --steps 10 --warmup 2 --no-api --no-strong --no-cpu-baseline --extra "" --sustain-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); k=d['kernels']; s=d['sampler']
print('  %-8s %9.0f samples/s %.3f ms | rng %.3f place %.3f merge %.3f tail %.3f sampler %.3f | full %d retried %d' % ('${CFG%%:*}', d['value'], d['ms_per_step'], k['k_rng_ms'], k['k_place_ms'], k['k_merge_ms'], k['k_tail_ms'], k['k_sampler_ms'], s['units_run_in_full'], s['units_retried']))"
  done
}
echo "default"; run
for SET in "4 6 64" "3 5 48" "3 4 32" "2.5 3.5 32" "2 3 24"; do
  set -- $SET
  echo "GAT_RNG_SIGMA_MIN=$1 GAT_RNG_SIGMA_MAX=$2 GAT_RNG_TAIL_ROWS=$3"
  GAT_RNG_SIGMA_MIN=$1 GAT_RNG_SIGMA_MAX=$2 GAT_RNG_TAIL_ROWS=$3 run
done

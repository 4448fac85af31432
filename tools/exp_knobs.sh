# quick sweeps of launch-shape knobs on the headline shapes (bench.py, 10 steps); usage: bash tools/exp_knobs.sh
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { env $1 python bench.py --config $2 --samples 10000 --steps 10 --warmup 2 --no-api --no-strong --no-cpu-baseline --extra "" --sustain-seconds 0 2>/dev/null | python tools/show_bench.py /dev/stdin | sed "s/^/$1  /"; }
for V in 8 16 32 64 128; do run GAT_COUNT_SAMPLES_PER_BLOCK=$V config2; done
for V in 1 3 6 8; do run GAT_SIZE_CLASSES=$V config2; done
for V in 2 4 8; do run GAT_MERGED_SAMPLES_PER_BLOCK=$V config3; done
for V in 3 6 8; do run GAT_SIZE_CLASSES=$V config3; done

# k_place's written-out steps (GAT_STEP_SIMPLE_ASM / GAT_STEP_TABLE_ASM) against the compiler's (GAT_PLACE_NO_CM=1): parity first,
# then the kernel's time on the headline shapes; usage: bash tools/exp_place_step.sh [quick]
cd ${GRAFT_REPO_ROOT:-$(pwd)}
[ "${1:-}" = quick ] || timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -5
run() { env $1 timeout 120 python bench.py --config $2 --samples $3 --steps 10 --warmup 2 --no-api --no-strong --no-cpu-baseline --extra "" --sustain-seconds 0 2>/dev/null | python tools/show_bench.py /dev/stdin | sed "s|^|$1 $3  |" | cut -c1-200; }
for E in GAT_X=0 GAT_PLACE_NO_CM=1; do run $E config3 10000; done
for E in GAT_X=0 GAT_PLACE_NO_CM=1; do run $E config3 1250; done
for E in GAT_X=0 GAT_PLACE_NO_CM=1; do run $E config5 16384; done
for E in GAT_X=0 GAT_PLACE_NO_CM=1; do run $E config2 10000; done
for E in GAT_X=0 GAT_PLACE_NO_CM=1; do run $E config2 1250; done
for E in GAT_X=0 GAT_PLACE_NO_CM=1; do run $E config4 12500; done

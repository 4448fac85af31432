# k_place's written-out step (GAT_STEP_SIMPLE_ASM) against the compiler's (GAT_PLACE_NO_CM=1): parity first, then the kernel's time
# on the headline shapes at 10 000 and 1 250 samples per call; usage: bash tools/exp_place_step.sh
cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_hip_parity.py -x -q -m gpu 2>&1 | tail -5
run() { env $1 python bench.py --config $2 --samples $3 --steps 10 --warmup 2 --no-api --no-strong --no-cpu-baseline --extra "" --sustain-seconds 0 2>/dev/null | python tools/show_bench.py /dev/stdin | sed "s/^/$1 $3  /"; }
for S in 10000 1250; do
  for E in GAT_X=0 GAT_PLACE_NO_CM=1; do run $E config2 $S; done
done
for E in GAT_X=0 GAT_PLACE_NO_CM=1; do run $E config5 16384; done
for E in GAT_X=0 GAT_PLACE_NO_CM=1; do run $E config4 12500; done

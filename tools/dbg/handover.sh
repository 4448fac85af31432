cd ${GRAFT_REPO_ROOT:-$(pwd)}
OUT=gpurun_out/dbg; mkdir -p $OUT; rm -f $OUT/*.jsonl
for E in GAT_X=0 GAT_PLACE_NO_CM=1; do
  env $E GAT_LIB_PATH=$PWD/build/diag/libgat_mi355_diag.so GAT_DIAG_OUT=$OUT/h_$E.jsonl python3 bench.py --no-cpu-baseline --no-api --no-strong --sustain-seconds 0 --extra "" --config config2 --samples 2000 --steps 1 --warmup 0 > $OUT/b.log 2>&1
  echo $E; grep handover $OUT/h_$E.jsonl | head -2 | cut -c1-1500
done

cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { env $1 python bench.py --config config2 --samples $2 --steps 10 --warmup 2 --no-api --no-strong --no-cpu-baseline --extra "" --sustain-seconds 0 2>/dev/null | python tools/show_bench.py /dev/stdin | sed "s|^|$3 $2  |"; }
for S in 10000 1250; do
  run GAT_X=0 $S base
  run GAT_LIB_PATH=$PWD/build/noflush/libgat_noflush.so $S noflush
done

cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { env $1 timeout 60 python bench.py --config config2 --samples $2 --steps 3 --warmup 1 --no-api --no-strong --no-cpu-baseline --extra "" --sustain-seconds 0 2>/dev/null | python tools/show_bench.py /dev/stdin | sed "s|^|$3 $2  |" | cut -c1-140; }
for S in 10000; do
  run GAT_X=0 $S base
  run GAT_LIB_PATH=$PWD/build/noflush/libgat_noflush.so $S noflush
  run GAT_LIB_PATH=$PWD/build/fl2/libgat_fl2.so $S lds_reads_no_stores
  run GAT_LIB_PATH=$PWD/build/fl3/libgat_fl3.so $S nt_stores
done

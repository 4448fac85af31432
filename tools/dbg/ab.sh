# A/B of builds of the library on the headline shapes: bash tools/dbg/ab.sh <label>=<.so> ...
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { env $1 timeout 120 python bench.py --config $3 --samples $4 --steps 10 --warmup 3 --no-api --no-strong --no-cpu-baseline --extra "" --sustain-seconds 0 2>/dev/null | python tools/show_bench.py /dev/stdin | sed "s|^|$2 $4  |" | cut -c1-210; }
for C in ${SHAPES:-config2:10000 config3:10000 config5:16384}; do
  S=${C##*:}; CFG=${C%%:*}
  run GAT_X=0 base $CFG $S
  for V in "$@"; do run GAT_LIB_PATH=$PWD/${V##*=} ${V%%=*} $CFG $S; done
done

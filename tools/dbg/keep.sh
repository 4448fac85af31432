cd ${GRAFT_REPO_ROOT:-$(pwd)}
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_async_seam.py tests/test_config_workloads.py -x -q -m gpu 2>&1 | tail -2
run() { env $1 timeout 120 python bench.py --config $2 --samples $3 --steps 10 --warmup 3 --no-api --no-strong --no-cpu-baseline --extra "" --sustain-seconds 0 2>/dev/null | python tools/show_bench.py /dev/stdin | sed "s|^|$1 $3  |" | cut -c1-205; }
for C in config2:10000 config3:10000 config5:16384 config2:1250 config3:1250; do S=${C##*:}; CFG=${C%%:*}; for E in GAT_X=0 GAT_PLACE_NO_KEEP_ROWS=1; do run $E $CFG $S; done; done

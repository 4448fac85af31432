export TMPDIR=/tmp
mkdir -p gpurun_out/r6c
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6c/pytest.log 2>&1
tail -3 gpurun_out/r6c/pytest.log
python bench.py --steps 20 --warmup 5 --details gpurun_out/r6c/bench_default.json > gpurun_out/r6c/bench_default.line 2> gpurun_out/r6c/bench_default.err
tail -c 600 gpurun_out/r6c/bench_default.line
python tools/show_bench.py gpurun_out/r6c/bench_default.json

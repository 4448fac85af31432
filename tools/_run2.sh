export TMPDIR=/tmp
mkdir -p gpurun_out/r6b
timeout 600 python -m pytest tests/test_hip_parity.py -x -q -k "fragmented or robustness or reference_dataset or golden_counts" > gpurun_out/r6b/pytest1.log 2>&1
tail -5 gpurun_out/r6b/pytest1.log
timeout 300 python tools/fuzz_sweep.py 1000 300 frag > gpurun_out/r6b/fuzz_frag.log 2>&1
tail -3 gpurun_out/r6b/fuzz_frag.log
python bench.py --config refdata --extra= --no-strong --no-api --no-cpu-baseline --steps 5 --warmup 2 --details gpurun_out/r6b/refdata.json > gpurun_out/r6b/refdata.line 2> gpurun_out/r6b/refdata.err
python tools/show_bench.py gpurun_out/r6b/refdata.json 2>/dev/null | head -30

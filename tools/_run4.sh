export TMPDIR=/tmp
mkdir -p gpurun_out/r6d
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_config_workloads.py -x -q -k "units_counted or golden_config or config3 or fuzz_shapes or merged_track or random_problems or golden_counts" > gpurun_out/r6d/pytest1.log 2>&1
tail -15 gpurun_out/r6d/pytest1.log
timeout 300 python tools/fuzz_sweep.py 10000 400 units 2>&1 | tail -5
python bench.py --config config3 --extra= --no-strong --no-api --no-cpu-baseline --steps 10 --warmup 3 --details gpurun_out/r6d/c3.json > /dev/null 2> gpurun_out/r6d/c3.err; python tools/show_bench.py gpurun_out/r6d/c3.json
GAT_COUNT_VIA_CONTIGS=1 python bench.py --config config3 --extra= --no-strong --no-api --no-cpu-baseline --steps 10 --warmup 3 --details gpurun_out/r6d/c3b.json > /dev/null 2> gpurun_out/r6d/c3b.err; python tools/show_bench.py gpurun_out/r6d/c3b.json

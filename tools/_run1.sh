export TMPDIR=/tmp
mkdir -p gpurun_out/r6a
python bench.py --config refdata --extra= --no-strong --no-api --no-cpu-baseline --steps 5 --warmup 2 --details gpurun_out/r6a/refdata.json > gpurun_out/r6a/refdata.line 2> gpurun_out/r6a/refdata.err
KSTATS_SCRIPT=tools/bench_refdata.py bash tools/kstats.sh 10000 > gpurun_out/r6a/refdata_kstats.txt 2>&1
cp gpurun_out/kstats/log gpurun_out/r6a/refdata_kstats.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r6a/pytest.log 2>&1
tail -3 gpurun_out/r6a/pytest.log

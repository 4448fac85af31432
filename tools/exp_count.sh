for v in "GAT_COUNT_SAMPLES_PER_BLOCK=16" "GAT_COUNT_SAMPLES_PER_BLOCK=32" "GAT_COUNT_SAMPLES_PER_BLOCK=64" "GAT_COUNT_SAMPLES_PER_BLOCK=128" "GAT_COUNT_SAMPLES_PER_BLOCK=64 GAT_COUNT_LDS_ENTRIES=2048"; do
env $v python bench.py --config config3 --samples 2000 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config3 $v', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
env $v python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config2 $v', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done

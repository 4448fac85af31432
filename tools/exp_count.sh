for v in "X=1"; do
env $v python bench.py --config config3 --samples 2000 --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config3 $v', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
env $v python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config2 $v', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
done

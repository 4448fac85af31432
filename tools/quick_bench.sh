#!/bin/bash
# quick look at the two bench shapes (no CPU baseline): value, per-kernel ms
show() { python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-8s %10.0f samples/s  %.3f ms/step  count %.3f ms frac %.3f  sampler %.3f ms' % (sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['sampler']['avg_launch_ms']))
" $1; }
python bench.py --no-cpu-baseline --steps 10 --warmup 3 | show config2
python bench.py --no-cpu-baseline --steps 10 --warmup 3 | show config2
python bench.py --no-cpu-baseline --config config3 --samples 2000 --steps 5 --warmup 2 | show config3

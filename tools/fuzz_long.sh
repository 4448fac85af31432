#!/bin/bash
# the seven generators of tools/fuzz_sweep.py side by side on one GPU, fresh seed ranges;
# usage: tools/fuzz_long.sh <seconds per generator> [offset added to every first seed]
T=${1:-3600}
O=${2:-0}
mkdir -p gpurun_out/fuzz_long
( timeout $T python tools/fuzz_sweep.py $((1000000 + O)) 100000        > gpurun_out/fuzz_long/shapes.txt 2>&1 ) &
( timeout $T python tools/fuzz_sweep.py $((2000000 + O)) 200000 edge   > gpurun_out/fuzz_long/edge.txt 2>&1 ) &
( timeout $T python tools/fuzz_sweep.py $((3000000 + O)) 80000  merged > gpurun_out/fuzz_long/merged.txt 2>&1 ) &
( timeout $T python tools/fuzz_sweep.py $((4000000 + O)) 60000  long   > gpurun_out/fuzz_long/long.txt 2>&1 ) &
( timeout $T python tools/fuzz_sweep.py $((5000000 + O)) 250000 scan   > gpurun_out/fuzz_long/scan.txt 2>&1 ) &
( timeout $T python tools/fuzz_sweep.py $((6000000 + O)) 150000 frag   > gpurun_out/fuzz_long/frag.txt 2>&1 ) &
( timeout $T python tools/fuzz_sweep.py $((7000000 + O)) 150000 units  > gpurun_out/fuzz_long/units.txt 2>&1 ) &
wait
tail -n 3 gpurun_out/fuzz_long/*.txt

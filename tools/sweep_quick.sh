# a few minutes of the fuzz generators (GPU vs oracle, bit-exact) behind a kernel change: bash tools/sweep_quick.sh
cd ${GRAFT_REPO_ROOT:-$(pwd)}
echo "shapes"; timeout 600 python tools/fuzz_sweep.py 120000 3000 2>&1 | tail -3
echo "shapes, GAT_RNG_SLACK=0.75"; GAT_RNG_SLACK=0.75 timeout 600 python tools/fuzz_sweep.py 150000 2000 2>&1 | tail -3
echo "edge"; timeout 600 python tools/fuzz_sweep.py 120000 3000 edge 2>&1 | tail -3
echo "merged"; timeout 600 python tools/fuzz_sweep.py 120000 1000 merged 2>&1 | tail -3
echo "long"; timeout 600 python tools/fuzz_sweep.py 120000 1000 long 2>&1 | tail -3

# the round's closing sweeps of the fuzz generators of tests/test_hip_parity.py over many seeds, GPU vs oracle, bit-exact (default
# row budget, and budgets cut so that streams run out and are resumed with the generator moved up): bash tools/sweep_r04.sh
cd ${GRAFT_REPO_ROOT:-$(pwd)}
echo "shapes, default budget"; timeout 1200 python tools/fuzz_sweep.py 20000 20000 2>&1 | tail -4
echo "shapes, GAT_RNG_SLACK=0.75"; GAT_RNG_SLACK=0.75 timeout 900 python tools/fuzz_sweep.py 50000 6000 2>&1 | tail -4
echo "shapes, GAT_RNG_SLACK=0.9"; GAT_RNG_SLACK=0.9 timeout 900 python tools/fuzz_sweep.py 60000 6000 2>&1 | tail -4
echo "edge"; timeout 1200 python tools/fuzz_sweep.py 20000 20000 edge 2>&1 | tail -4
echo "edge, slack 0.8"; GAT_RNG_SLACK=0.8 timeout 900 python tools/fuzz_sweep.py 50000 6000 edge 2>&1 | tail -4
echo "merged"; timeout 900 python tools/fuzz_sweep.py 20000 6000 merged 2>&1 | tail -4
echo "long"; timeout 1500 python tools/fuzz_sweep.py 20000 8000 long 2>&1 | tail -4
echo "long, slack 0.85"; GAT_RNG_SLACK=0.85 timeout 900 python tools/fuzz_sweep.py 40000 3000 long 2>&1 | tail -4

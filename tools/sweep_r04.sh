cd ${GRAFT_REPO_ROOT:-$(pwd)}
echo "shapes, default budget"; timeout 600 python tools/fuzz_sweep.py 1000 3000 2>&1 | tail -4
echo "shapes, GAT_RNG_SLACK=0.75 (streams run out: resumed with the generator moved up)"; GAT_RNG_SLACK=0.75 timeout 600 python tools/fuzz_sweep.py 5000 3000 2>&1 | tail -4
echo "shapes, GAT_RNG_SLACK=0.9"; GAT_RNG_SLACK=0.9 timeout 600 python tools/fuzz_sweep.py 9000 2000 2>&1 | tail -4
echo "edge"; timeout 600 python tools/fuzz_sweep.py 1000 3000 edge 2>&1 | tail -4
echo "edge, slack 0.8"; GAT_RNG_SLACK=0.8 timeout 600 python tools/fuzz_sweep.py 5000 2000 edge 2>&1 | tail -4
echo "merged"; timeout 600 python tools/fuzz_sweep.py 1000 1500 merged 2>&1 | tail -4
echo "long"; timeout 900 python tools/fuzz_sweep.py 1000 1200 long 2>&1 | tail -4
echo "long, slack 0.85"; GAT_RNG_SLACK=0.85 timeout 900 python tools/fuzz_sweep.py 3000 600 long 2>&1 | tail -4

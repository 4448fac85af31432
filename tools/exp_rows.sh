#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
run() { echo "== $1"; env $1 python3 bench.py --no-cpu-baseline --no-api --no-strong --sustain-seconds 0 --extra config3,config5 > /tmp/b.json 2>/dev/null; python3 tools/show_bench.py /tmp/b.json; python3 - <<'PY'
import json
d=[json.loads(l) for l in open('/tmp/b.json') if l.startswith('{')][-1]
print("   full units:", d["sampler"]["units_run_in_full"], [v["sampler"]["units_run_in_full"] for v in d["configs"].values()])
PY
}
run "GAT_NOP=1"
run "GAT_RNG_SIGMA_MIN=4.0 GAT_RNG_SIGMA_MAX=6.5 GAT_RNG_TAIL_ROWS=64"
run "GAT_RNG_SIGMA_MIN=3.5 GAT_RNG_SIGMA_MAX=6.0 GAT_RNG_TAIL_ROWS=64"
run "GAT_RNG_SIGMA_MIN=3.0 GAT_RNG_SIGMA_MAX=6.0 GAT_RNG_TAIL_ROWS=48"

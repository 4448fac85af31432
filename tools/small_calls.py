"""the small-call floor: one call of S samples (what each of N GPUs gets of a 10 000-sample job) with the lane-per-stream
front end (k_rng + k_place + split path) against the wave-per-unit sampler alone (GAT_SAMPLER_MODE=wave).
usage: tools/small_calls.py [config] [S ...]"""
import json, os, subprocess, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
cfg = sys.argv[1] if len(sys.argv) > 1 else "config2"
sizes = [int(x) for x in sys.argv[2:]] or [625, 1250, 2500, 5000, 10000]
for S in sizes:
    row = []
    for mode in ("lane", "wave"):
        env = dict(os.environ)
        if mode == "wave":
            env["GAT_SAMPLER_MODE"] = "wave"
        else:
            env.pop("GAT_SAMPLER_MODE", None)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--extra", "", "--config", cfg,
                              "--samples", str(S), "--steps", "20", "--warmup", "3"], env=env, capture_output=True, text=True).stdout
        d = [json.loads(l) for l in out.splitlines() if l.startswith("{")][-1]
        k = d["kernels"]
        row.append("%s %.3f ms/call (sampler %.3f: rng %.3f place %.3f merge %.3f tail %.3f k_sampler %.3f; contig %.3f count %.3f)" %
                   (mode, d["ms_per_step"], k["sampler_phase_ms"], k["k_rng_ms"], k["k_place_ms"], k["k_merge_ms"], k["k_tail_ms"],
                    k["k_sampler_ms"], k["k_contig_ms"], k["count_phase_ms"]))
    print("%s S=%d: %s | %s" % (cfg, S, row[0], row[1]), flush=True)

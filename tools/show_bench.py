#!/usr/bin/env python3
"""one line per configuration of a bench.py details file (bench_details.json): value, ms/step, per-kernel ms;
usage: tools/show_bench.py FILE..."""
import json
import sys


def show(name, r):
    k = r["kernels"]
    wu = max(1, r["sampler"]["work_units"])
    print("%-8s %10.0f samples/s %8.3f ms/step | rng %.3f place %.3f consol %.3f tail %.3f final %.3f sampler %.3f (tail did %.1f%%) contig %.3f count %.3f (%s) | frac %.2f"
          % (name, r["value"], r["ms_per_step"], k["k_rng_ms"], k["k_place_ms"], k["k_merge_ms"], k.get("k_tail_ms", 0), k.get("k_finalize_ms", 0),
             k["k_sampler_ms"], 100.0 * r["sampler"].get("units_finished_by_k_tail", 0) / wu,
             k["k_contig_ms"], k["count_main_ms"], r["roofline"]["kernel"].split()[0], r["roofline"]["frac"]))


for fn in sys.argv[1:]:
    text = open(fn).read()
    try:
        docs = [json.loads(text)]
    except ValueError:                                     # (a log with the report as one of its lines: earlier rounds)
        docs = [json.loads(line) for line in text.splitlines() if line.startswith("{")]
    for d in docs:
        if "kernels" not in d:
            continue
        show(d["config"]["workload"].split(":")[0], d)
        for k, v in d.get("configs", {}).items():
            show(k, v)

#!/usr/bin/env python3
"""one line per configuration of a bench.py JSON line: value, ms/step, per-kernel ms; usage: tools/show_bench.py FILE..."""
import json
import sys


def show(name, r):
    k = r["kernels"]
    print("%-8s %10.0f samples/s %8.3f ms/step | rng %.3f place %.3f merge %.3f sampler %.3f contig %.3f count %.3f (%s) | frac %.2f"
          % (name, r["value"], r["ms_per_step"], k["k_rng_ms"], k["k_place_ms"], k["k_merge_ms"], k["k_sampler_ms"],
             k["k_contig_ms"], k["count_main_ms"], r["roofline"]["kernel"].split()[0], r["roofline"]["frac"]))


for fn in sys.argv[1:]:
    for line in open(fn):
        if not line.startswith("{"):
            continue
        d = json.loads(line)
        show(d["config"]["workload"].split(":")[0], d)
        for k, v in d.get("configs", {}).items():
            show(k, v)

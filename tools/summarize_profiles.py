#!/usr/bin/env python3
"""Turn the rocprofv3 output of tools/collect_profiles.sh (gpurun_out/prof_<tag>/<config>/) into the summaries that are
committed under profiles/: per configuration the kernel-trace statistics (CSV, gat:: kernels), a text table of the
counters per kernel and launch, and <tag>_kernel_counters.json, which bench.py reads for the `traffic` / `valu_busy`
fields of its roofline block (keyed "config:samples per step").

Counters (MI355X_MICROARCH.md, HBM / rocprofv3 sections): FETCH_SIZE and WRITE_SIZE are KiB summed over the 8 XCDs,
collected in separate passes; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced streaming read (re-checked
in round 1 on k_place, whose reads are exactly the rows k_rng writes), other access shapes are uncalibrated.  So the
correction is applied per kernel: x 2 for the kernels whose reads are coalesced streams (STREAMING below), x 1 -- the raw
counter -- for the ones that gather (per-lane binary searches, index look-ups): doubling those gave rates above what a copy
achieves (6.29 TB/s), i.e. not evidence.  Both figures are written; a x 2 figure beyond 6.29 TB/s falls back to the raw
one and is flagged.  VALU issue share = SQ_INSTS_VALU x 2 cycles (a wave64 instruction occupies a SIMD-32 for two cycles) /
(4 SIMDs x 256 CUs x kernel cycles); kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the XCDs).
usage: tools/summarize_profiles.py gpurun_out/prof_<tag> <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

# kernels whose global reads are coalesced streams over whole lists / rows (the FETCH_SIZE x 2 correction applies)
STREAMING = ("k_place", "k_rng", "k_seed", "k_finalize", "k_consolidate", "k_merge_big", "k_count_seg<", "k_count_finish", "k_reduce_stats",
             "k_sampler", "k_count_swap", "k_null_stats")
HBM_ACHIEVABLE_GBPS = 6290.0

def kernel_sources_sha(root):
    """what the committed counters belong to: the kernels' sources at the time of the collection (bench.py compares it with
    the tree it runs from and quotes no counter of other kernels)"""
    import hashlib
    h = hashlib.sha256()
    for fn in ("gat_kernels.h", "gat_tail.h", "gat_device.h", "gat_stats.h", "gat_types.h"):
        h.update(open(os.path.join(root, "gat_amd", "csrc", fn), "rb").read())
    return h.hexdigest()[:16]


src, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
counters_json = {}
for d in sorted(glob.glob(os.path.join(src, "*"))):
    cfg = os.path.basename(d)
    if not os.path.isdir(d):
        continue
    res = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(os.path.join(d, "*", "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "gat::" not in n:
                continue
            res[n][r["Counter_Name"]] += float(r["Counter_Value"])
            calls[n][r["Counter_Name"]] += 1
    stats = {}
    ncalls = {}
    ks = os.path.join(d, "kernel_stats.csv")
    if os.path.exists(ks):
        rows = [r for r in csv.DictReader(open(ks)) if "gat::" in r["Name"]]
        with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, cfg)), "w") as out:
            w = csv.writer(out)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
                stats[r["Name"]] = float(r["AverageNs"])
                ncalls[r["Name"]] = int(r["Calls"])
    bj = os.path.join(d, "bench.json")
    S = None
    if os.path.exists(bj) and os.path.getsize(bj):
        b = json.load(open(bj))
        S = b["config"]["samples_per_step_per_gpu"]
        shutil.copy(bj, os.path.join(dst, "%s_%s_bench.json" % (tag, cfg)))
    with open(os.path.join(dst, "%s_%s_pmc.txt" % (tag, cfg)), "w") as out:
        out.write("# rocprofv3 --pmc, one pass per counter group (tools/collect_profiles.sh), bench.py --config %s --samples %s "
                  "--steps 3 --warmup 1; values per launch, summed over the 8 XCDs; FETCH_SIZE / WRITE_SIZE in KiB\n" % (cfg, S))
        per_kernel = {}
        for n in sorted(res):
            out.write(n + "\n")
            v = dict((c, res[n][c] / max(1, calls[n][c])) for c in res[n])
            for c in sorted(v):
                out.write("   %-28s %16.0f per launch\n" % (c, v[c]))
            rec = {"fetch_kib_per_launch": v.get("FETCH_SIZE", 0.0), "write_kib_per_launch": v.get("WRITE_SIZE", 0.0),
                   "avg_ns": stats.get(n)}
            if v.get("GRBM_GUI_ACTIVE") and v.get("SQ_INSTS_VALU") is not None:
                cycles = v["GRBM_GUI_ACTIVE"] / 8.0
                rec["valu_busy"] = v["SQ_INSTS_VALU"] * 2.0 / (4 * 256 * cycles)
                rec["waves_per_cu_avg"] = v.get("SQ_WAVE_CYCLES", 0.0) * 4.0 / cycles / 256.0 if v.get("SQ_WAVE_CYCLES") else None
                out.write("   %-28s %16.3f\n" % ("valu_issue_share", rec["valu_busy"]))
                if rec["waves_per_cu_avg"]:
                    out.write("   %-28s %16.2f\n" % ("avg_waves_per_cu", rec["waves_per_cu_avg"]))
            if v.get("TCC_REQ_sum"):
                # requests that reach the L2s (128-byte lines) and the share served there
                rec["l2_requests_per_launch"] = v["TCC_REQ_sum"]
                rec["l2_hit_rate"] = v.get("TCC_HIT_sum", 0.0) / max(1.0, v.get("TCC_HIT_sum", 0.0) + v.get("TCC_MISS_sum", 0.0))
                if stats.get(n):
                    rec["l2_requests_per_us"] = v["TCC_REQ_sum"] / (stats[n] / 1e3)
                    out.write("   %-28s %16.1f per us, hit rate %.3f\n" % ("l2_requests", rec["l2_requests_per_us"], rec["l2_hit_rate"]))
            raw = (rec["fetch_kib_per_launch"] + rec["write_kib_per_launch"]) * 1024.0
            x2 = (2.0 * rec["fetch_kib_per_launch"] + rec["write_kib_per_launch"]) * 1024.0
            factor = 2.0 if any(k in n for k in STREAMING) else 1.0
            note = "coalesced streaming reads" if factor == 2.0 else "gathers: raw counter"
            if factor == 2.0 and stats.get(n) and x2 / stats[n] > HBM_ACHIEVABLE_GBPS:
                factor, note = 1.0, "x 2 would exceed what a copy achieves: raw counter"
            hbm = x2 if factor == 2.0 else raw
            rec["fetch_factor"], rec["fetch_factor_why"] = factor, note
            rec["hbm_bytes_raw_per_launch"], rec["hbm_bytes_x2_per_launch"] = raw, x2
            rec["hbm_bytes_per_launch"] = hbm
            if stats.get(n):
                rec["hbm_GBps"] = hbm / stats[n]
                out.write("   %-28s %16.1f GB/s (%g x FETCH + WRITE over the average launch; %s; raw %.1f, x 2 %.1f)\n" %
                          ("hbm_rate", rec["hbm_GBps"], factor, note, raw / stats[n], x2 / stats[n]))
            per_kernel[n] = rec
    if S is not None:
        count = [n for n in per_kernel if ("k_count_merged" in n and "finish" not in n) or "k_count_seg<" in n or "k_count_swap" in n]
        entry = {"kernels": per_kernel}
        if count:
            main = max(count, key=lambda n: per_kernel[n].get("avg_ns") or 0)
            entry["count_kernel"] = dict(per_kernel[main], name=main)
            # the whole step: every kernel's bytes x its launches per step (launches of the main count kernel = batches = steps
            # here: one batch per step at these sizes)
            steps = float(max(1, ncalls.get(main, 1)))
            tot, tot_ns = 0.0, 0.0
            for n, rec in per_kernel.items():
                per_step = ncalls.get(n, 0) / steps
                rec["launches_per_step"] = per_step
                tot += rec["hbm_bytes_per_launch"] * per_step
                tot_ns += (rec.get("avg_ns") or 0.0) * per_step
            entry["step"] = {"hbm_bytes": tot, "kernel_ns": tot_ns, "hbm_GBps_over_kernel_time": tot / tot_ns if tot_ns else None}
        counters_json["%s:%d" % (cfg, S)] = entry
counters_json["_meta"] = {"tag": tag, "kernel_sources_sha": kernel_sources_sha(root),
                          "collected_by": "tools/collect_profiles.sh (rocprofv3 --kernel-trace --stats, then one --pmc pass per counter group)"}
with open(os.path.join(dst, "%s_kernel_counters.json" % tag), "w") as f:
    json.dump(counters_json, f, indent=1, sort_keys=True)
print("wrote", sorted(k for k in counters_json if k != "_meta"))

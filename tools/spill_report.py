#!/usr/bin/env python3
"""Where the register spills of the device code land: for every kernel of the device assembly (-save-temps=obj,
build/gat_mi355-hip-amdgcn-amd-amdhsa-gfx950.s) the metadata's spill counts and, instruction by instruction, the SGPR spills
(v_writelane_b32 vN, sM, <imm> / v_readlane_b32 sM, vN, <imm>: the backend parks scalar registers in the lanes of a vector
register) and scratch accesses by LOOP DEPTH -- depth = the number of backward branches whose [target, branch] range holds the
instruction.  A spill at depth 0 runs once per wave (prologue / epilogue / between the loops of a kernel that holds several);
one inside a loop runs per trip.  The hand-pipelined loops (GAT_PINNED_BEGIN / GAT_PINNED_END) are listed by themselves.
Last column: the instructions inside the loops of the marked regions and how many of them are spill traffic.
usage: tools/spill_report.py <device .s> [--max-marked-share PERCENT]   (exit 1 if spill traffic exceeds that share of a
kernel's marked loops: the Makefile runs it with 3)"""
import re
import subprocess
import sys

WRITELANE = re.compile(r"^\s*v_writelane_b32 v\d+, s\d+, \d+\s*$")
READLANE = re.compile(r"^\s*v_readlane_b32 s\d+, v\d+, \d+\s*$")
SCRATCH = re.compile(r"^\s*scratch_(load|store)")
BRANCH = re.compile(r"^\s*s_c?branch\w*\s+(\.LBB\d+_\d+)")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")


def kernels(path):
    """(mangled name, [(line number, code, in pinned loop)]) per kernel body"""
    out, cur, body, pinned = [], None, [], False
    for no, line in enumerate(open(path, errors="replace"), 1):
        m = re.match(r"^(_ZN3gat\w+):\s*(;.*)?$", line)
        if m:
            if cur:
                out.append((cur, body))
            cur, body, pinned = m.group(1), [], False
            continue
        if cur is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            out.append((cur, body))
            cur = None
            continue
        if "GAT_PINNED_BEGIN" in line:
            pinned = True
        if "GAT_PINNED_END" in line:
            pinned = False
        code = line.split("//")[0].rstrip()
        if LABEL.match(code) or (code.strip() and not code.lstrip().startswith((";", "."))):
            body.append((no, code, pinned))
    return out


def analyse(body):
    label_at = {}
    for i, (_, code, _) in enumerate(body):
        m = LABEL.match(code)
        if m:
            label_at[m.group(1)] = i
    loops = []
    for i, (_, code, _) in enumerate(body):
        m = BRANCH.match(code)
        if m and m.group(1) in label_at and label_at[m.group(1)] <= i:
            loops.append((label_at[m.group(1)], i))
    delta = [0] * (len(body) + 1)
    for a, b in loops:
        delta[a] += 1
        delta[b + 1] -= 1
    stats = {"instructions": 0, "loops": len(loops), "pinned_loop_insts": 0}
    d = 0
    for i, (_, code, pinned) in enumerate(body):
        d += delta[i]
        if LABEL.match(code):
            continue
        stats["instructions"] += 1
        if pinned and d >= 1:
            stats["pinned_loop_insts"] += 1
        kind = "sgpr_spill" if WRITELANE.match(code) else "sgpr_reload" if READLANE.match(code) else "scratch" if SCRATCH.match(code) else None
        if kind:
            key = "%s_d%s" % (kind, "0" if d == 0 else "1" if d == 1 else "2+")
            stats[key] = stats.get(key, 0) + 1
            if pinned and d >= 1:
                stats["pinned_" + kind] = stats.get("pinned_" + kind, 0) + 1
    return stats


def metadata(path):
    s = open(path, errors="replace").read()
    md = s[s.index("amdhsa.kernels:"):]
    out = {}
    for e in md.split("\n  - .agpr_count")[1:]:
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, e).group(1))      # noqa: E731
        out[re.search(r"\.name:\s+(\S+)", e).group(1)] = dict(
            sgpr=g("sgpr_count"), sgpr_spill=g("sgpr_spill_count"), vgpr=g("vgpr_count"), vgpr_spill=g("vgpr_spill_count"),
            scratch=g("private_segment_fixed_size"), lds=g("group_segment_fixed_size"))
    return out


def main(argv):
    path = argv[1]
    limit = float(argv[argv.index("--max-marked-share") + 1]) if "--max-marked-share" in argv else None
    md = metadata(path)
    ks = kernels(path)
    names = subprocess.run(["c++filt"] + [k for k, _ in ks], capture_output=True, text=True).stdout.splitlines()
    print("%-62s %5s %5s %5s %5s %5s | SGPR spills+reloads by loop depth 0 / 1 / 2+ | scratch 0 / 1 / 2+ | in marked loops"
          % ("kernel", "insts", "sgpr", "s.spl", "vgpr", "v.spl"))
    bad = []
    for (k, body), name in zip(ks, names):
        st, m = analyse(body), md.get(k, {})
        if not (m.get("sgpr_spill") or m.get("vgpr_spill") or m.get("scratch")):
            continue
        name = re.sub(r"^void ", "", name).replace("gat::", "").replace("(SamplerArgs)", "").replace("(TailArgs)", "")
        sp = [st.get("sgpr_spill_d" + d, 0) + st.get("sgpr_reload_d" + d, 0) for d in ("0", "1", "2+")]
        sc = [st.get("scratch_d" + d, 0) for d in ("0", "1", "2+")]
        hot = st.get("pinned_sgpr_spill", 0) + st.get("pinned_sgpr_reload", 0) + st.get("pinned_scratch", 0)
        n = st["pinned_loop_insts"]
        share = 100.0 * hot / n if n else 0.0
        print("%-62s %5d %5d %5d %5d %5d | %6d / %5d / %5d                     | %4d / %3d / %3d     | %s"
              % (name[:62], st["instructions"], m.get("sgpr", 0), m.get("sgpr_spill", 0), m.get("vgpr", 0), m.get("vgpr_spill", 0),
                 sp[0], sp[1], sp[2], sc[0], sc[1], sc[2], "%d of %d (%.1f %%)" % (hot, n, share) if n else "-"))
        if limit is not None and share > limit:
            bad.append("%s %.1f %%" % (name, share))
    if bad:
        sys.exit("spill_report: spill traffic above %.1f %% of the instructions of the marked loops of: %s" % (limit, "; ".join(bad)))


if __name__ == "__main__":
    main(sys.argv)

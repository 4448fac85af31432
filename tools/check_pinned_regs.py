#!/usr/bin/env python3
"""Build-time guard for k_place_pipe's hand-pipelined rows (gat_kernels.h, GAT_PLACE_LOOP_PIPE): the loop keeps loads in
flight into v96..v127 and nothing tells the register allocator so -- it stays below v96 only because the loop's own values fit
there (the amdgpu_num_vgpr attribute does not bind on gfx950).  This script reads the device assembly the compile leaves
behind (-save-temps=obj) and fails the build if, between the markers GAT_PINNED_BEGIN / GAT_PINNED_END the loop emits, any
COMPILER-GENERATED instruction names one of those registers (the loop's own asm statements -- the row loads, the takes
v_mov_b32 vN, v9x, the written-out steps that read a row register in place -- stand between ;;#ASMSTART / ;;#ASMEND and are
the author's business); a silent corruption of the random rows becomes a build error.
usage: tools/check_pinned_regs.py <device .s>"""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
LOAD = re.compile(r"^\s*global_load_dword v(\d+), v\[\d+:\d+\], off( offset:\d+)?( nt)?\s*$")
TAKE = re.compile(r"^\s*v_mov_b32(_e32)? v(\d+), v(\d+)\s*$")
FIRST = re.compile(r"^\s*\S+\s+v(\d+)\b")
BASE = 96          # the pinned rows of the region at hand: v[BASE, BASE + 32) -- 96 unless the marker says "BASE=n" (k_place_grid<8>: 192)


def pinned(n):
    return BASE <= n < BASE + 32


def touches(line):
    for m in REG.finditer(line):
        if m.group(1):
            if pinned(int(m.group(1))):
                return True
            continue
        lo, hi = int(m.group(2)), int(m.group(3))
        if lo <= BASE + 31 and hi >= BASE:
            return True
    return False


def main(path):
    global BASE
    kernel, inside, regions, bad, nloads, in_asm = None, False, 0, [], {}, False
    for no, line in enumerate(open(path, errors="replace"), 1):
        code = line.split("//")[0]
        m = re.match(r"^(_ZN3gat\w+):", line)
        if m:
            kernel, inside = m.group(1), False
        if "#ASMSTART" in line or "#ASMEND" in line:
            in_asm = "#ASMSTART" in line
            continue
        if "GAT_PINNED_BEGIN" in line:
            inside, regions = True, regions + 1
            m = re.search(r"BASE=(\d+)", line)
            BASE = int(m.group(1)) if m else 96
            continue
        if "GAT_PINNED_END" in line:
            inside = False
            continue
        if not inside or not code.strip() or code.lstrip().startswith((";", ".")):
            continue
        # the loop's wait counts (s_waitcnt vmcnt(24): three chunks of eight row loads on their way) assume that the pinned
        # loads are the loop's only LOADS: stores in between make a wait stricter, a spill's reload or any other load would too,
        # but a loop that spills no longer fits the registers below v96 anyway -- refused as well
        if re.match(r"^\s*(scratch_|buffer_)", code):
            bad.append((kernel, no, code.strip() + "   <- scratch / buffer access inside a pinned loop"))
            continue
        ld = LOAD.match(code)
        if ld and pinned(int(ld.group(1))):
            nloads[regions] = nloads.get(regions, 0) + 1
        if touches(code):
            t = TAKE.match(code)
            if (ld and pinned(int(ld.group(1)))) or (t and pinned(int(t.group(3))) and not pinned(int(t.group(2)))):
                continue
            f = FIRST.match(code)
            if in_asm and not re.match(r"^\s*(global_|flat_|buffer_|scratch_|ds_read|ds_load)", code) and \
                    not (f and pinned(int(f.group(1)))):
                continue                  # a hand-written instruction READING a row register (never a load into one, never its destination)
            bad.append((kernel, no, code.strip()))
    if regions == 0:
        sys.exit("check_pinned_regs: no GAT_PINNED_BEGIN marker in %s (wrong file?)" % path)
    if bad:
        for k, no, code in bad[:20]:
            sys.stderr.write("%s:%d: %s  [%s]\n" % (path, no, code, k))
        sys.exit("check_pinned_regs: %d instruction(s) inside a pinned-register loop use its pinned rows (v96..v127, or the 32 from "
                 "the marker's BASE): the loop's values no longer fit below them -- lower the pressure or take the loop off the pipe" % len(bad))
    odd = [r for r, n in nloads.items() if n % 8 != 0 or n == 0]
    if odd or len(nloads) != regions:
        sys.exit("check_pinned_regs: pinned loops with a number of row loads that is not a multiple of the chunk (8): %r" % (nloads,))
    print("check_pinned_regs: %d pinned-register loops clean" % regions)


if __name__ == "__main__":
    main(sys.argv[1])

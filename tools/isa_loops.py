#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a hipcc -S listing.

    python tools/isa_loops.py file.s 'k_flow_iterILi7ELi2'

Finds backward branches (loops), and prints for each loop body the number of VALU f32 / f64 / other
VALU, LDS, global memory, scalar and waitcnt instructions.  A development aid for the marching kernels."""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_"):
        if "f64" in op:
            return "valu_f64"
        if op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_writelane", "v_readfirstlane")):
            return "valu_mov"
        if "f32" in op or "f16" in op:
            return "valu_f32"
        return "valu_int"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN\S*" + re.escape(pat) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    labels = {}
    instrs = []
    for l in body:
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            labels[m.group(1)] = len(instrs)
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        instrs.append(t.split(";")[0].strip())
    print(f"{pat}: {len(instrs)} instructions")
    loops = []
    for i, ins in enumerate(instrs):
        m = re.match(r"s_cbranch\S*\s+(\.LBB\S+)|s_branch\s+(\.LBB\S+)", ins)
        if m:
            tgt = labels.get(m.group(1) or m.group(2))
            if tgt is not None and tgt <= i:
                loops.append((tgt, i))
    for a, b in sorted(loops, key=lambda t: t[0] - t[1]):
        c = Counter(classify(x.split()[0]) for x in instrs[a:b + 1])
        n = b - a + 1
        print(f"  loop [{a}, {b}] {n} instr: " + ", ".join(f"{k}={v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
        if len(sys.argv) > 3 and sys.argv[3] == "-v":
            ops = Counter(x.split()[0] for x in instrs[a:b + 1])
            print("     " + ", ".join(f"{k}:{v}" for k, v in ops.most_common(40)))


if __name__ == "__main__":
    main()

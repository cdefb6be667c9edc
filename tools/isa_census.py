#!/usr/bin/env python3
"""Block census of one kernel's gfx950 assembly (hipcc -S --cuda-device-only): per basic block the number of VALU, SALU,
LDS, VMEM and branch instructions, v_readlane / v_writelane (SGPR spill traffic and lane picks), and the backward
branches (loops).  usage: isa_census.py file.s kernel-substring [--blocks]"""
import re
import sys


def kind(op):
    if op.startswith(("v_readlane", "v_readfirstlane")):
        return "rdl"
    if op.startswith("v_writelane"):
        return "wrl"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_cbranch", "s_branch")):
        return "br"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_setprio")):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_")):
        return "vmem"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pat in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, cur = [], {"name": "entry", "n": {}, "line": start, "succ": []}
    for i in range(start + 1, end):
        l = lines[i].strip()
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(cur)
            cur = {"name": m.group(1), "n": {}, "line": i, "succ": []}
            continue
        if not l or l.startswith((";", ".")):
            continue
        op = l.split()[0]
        k = kind(op)
        cur["n"][k] = cur["n"].get(k, 0) + 1
        if k == "br":
            cur["succ"].append(l.split()[-1])
    blocks.append(cur)
    idx = {b["name"]: j for j, b in enumerate(blocks)}
    tot = {}
    for b in blocks:
        for k, v in b["n"].items():
            tot[k] = tot.get(k, 0) + v
    print("kernel", lines[start][:80], "blocks", len(blocks), "totals", tot)
    loops = []
    for j, b in enumerate(blocks):
        for s in b["succ"]:
            if s in idx and idx[s] <= j:
                loops.append((idx[s], j))
    for a, z in sorted(loops, key=lambda x: x[0] - x[1]):
        t = {}
        for b in blocks[a:z + 1]:
            for k, v in b["n"].items():
                t[k] = t.get(k, 0) + v
        print("loop %s .. %s (%d blocks, asm lines %d-%d): %s" % (blocks[a]["name"], blocks[z]["name"], z - a + 1,
                                                                 blocks[a]["line"] + 1, blocks[z]["line"] + 1, t))
    if "--blocks" in sys.argv:
        for b in blocks:
            print(b["name"], b["line"] + 1, b["n"], "->", ",".join(b["succ"]))


if __name__ == "__main__":
    main()

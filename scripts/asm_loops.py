#!/usr/bin/env python
"""Static instruction mix of the loops of one function in a gfx950 assembly listing (hipcc --cuda-device-only -S): every backward branch
closes a loop; per loop the number of vector / scalar / LDS / memory / scratch / lane-spill instructions in its body.  Used to see which
loop bodies the scalar unit pays for (HISTORY.md section 17).

    python scripts/asm_loops.py file.s <function-name-substring> [min-body-size]
"""
import re
import sys


def klass(op):
    if op.startswith(("v_writelane", "v_readlane", "v_readfirstlane")):
        return "lane"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("s_waitcnt", "s_nop", "s_branch", "s_cbranch", "s_barrier", "s_endpgm", "s_setpc", "s_swappc", "s_getpc", "s_sleep", "s_setprio")):
        return "ctl"
    if op.startswith(("s_load", "s_buffer_load", "s_store")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    return "other"


def main():
    path, fn = sys.argv[1], sys.argv[2]
    min_body = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^\S*%s\S*:\s*(;.*)?$" % re.escape(fn), l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    ins, labels = [], {}
    for i in range(start, end):
        l = lines[i].split(";")[0].strip()
        if not l or l.startswith("."):
            m = re.match(r"^(\.LBB\w+):", l)
            if m:
                labels[m.group(1)] = len(ins)
            continue
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        ins.append(l)
    tot = {}
    for l in ins:
        k = klass(l.split()[0]); tot[k] = tot.get(k, 0) + 1
    print("function %s: %d instructions, static mix %s" % (lines[start].rstrip(":"), len(ins), tot))
    loops = []
    for j, l in enumerate(ins):
        p = l.split()
        if p[0].startswith(("s_cbranch", "s_branch")) and p[-1] in labels and labels[p[-1]] <= j:
            loops.append((labels[p[-1]], j))
    loops.sort(key=lambda x: (x[0], -x[1]))
    for a, b in loops:
        if b - a < min_body:
            continue
        mix = {}
        for l in ins[a:b + 1]:
            k = klass(l.split()[0]); mix[k] = mix.get(k, 0) + 1
        depth = sum(1 for (c, d) in loops if c <= a and d >= b and (c, d) != (a, b))
        calls = sum(1 for l in ins[a:b + 1] if l.startswith("s_swappc"))
        print("%s loop @%d..%d (%d instr, %d call(s)): %s" % ("  " * depth, a, b, b - a + 1, calls, " ".join("%s=%d" % kv for kv in sorted(mix.items()))))


if __name__ == "__main__":
    main()

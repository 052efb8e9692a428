#!/usr/bin/env python3
"""Static instruction mix of one kernel by source line / phase.

    hipcc ... -gline-tables-only --cuda-device-only -S csrc/rollout_kernel.hip -o /tmp/r.s
    python tools/isa_by_line.py /tmp/r.s 'rollout_kernelILi20ELi20ELi2ELi16ELi25' [--lines]

Counts the instructions of the kernel's body between its label and .Lfunc_end, attributes each to the last `.loc`
(innermost inlined location), and prints totals by instruction class and by source-line bucket.  Static counts only:
loops and branches are not weighted (the step loop of the roll-out kernels is straight-line code inside one loop).
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64", "v_rcp_f64", "v_div", "v_cmp_", "v_cmpx")):
        return "valu_f64" if "f64" in op else "valu_other"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    show_lines = "--lines" in sys.argv
    files = {}
    inside = False
    cur = ("?", 0)
    by_line = collections.Counter()
    by_class = collections.Counter()
    by_op = collections.Counter()
    by_line_class = collections.defaultdict(collections.Counter)
    for ln in open(path):
        s = ln.strip()
        m = re.match(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
            continue
        if not inside:
            if re.match(r"^_Z\w*%s\w*:" % re.escape(pat), ln):
                inside = True
            continue
        if s.startswith(".Lfunc_end"):
            break
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        if not s or s.startswith((".", ";", "//")) or s.endswith(":"):
            continue
        op = s.split()[0]
        c = classify(op)
        by_line[cur] += 1
        by_class[c] += 1
        by_op[op] += 1
        by_line_class[cur][c] += 1
    tot = sum(by_class.values())
    print("total instructions: %d" % tot)
    for c, n in by_class.most_common():
        print("  %-12s %6d" % (c, n))
    print("top opcodes:")
    for o, n in by_op.most_common(40):
        print("  %-28s %6d" % (o, n))
    if show_lines:
        print("by source line:")
        for (f, l), n in sorted(by_line.items()):
            print("  %-22s %5d  %5d   %s" % (f, l, n, dict(by_line_class[(f, l)])))


if __name__ == "__main__":
    main()

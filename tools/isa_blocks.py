#!/usr/bin/env python3
"""tools/isa_blocks.py KERNEL_SUBSTRING [FIRST_LABEL LAST_LABEL] — vector-instruction counts per basic block of one kernel, from the ISA
text tools/kres2.py leaves in /tmp/ssd_kernels.s (no GPU needed).  A static view of where a loop's instructions are: per block the
vector ALU instructions (v_*, including v_readlane / v_cmp and those inside inline asm), LDS and memory instructions, and the
branch that ends it.  Used in round 6 to take K1's tile body from 42 to 25 vector instructions per point before anything was timed."""
import re
import sys

def main():
    want = sys.argv[1]
    text = open("/tmp/ssd_kernels.s").read()
    funcs = re.split(r"\n\t\.globl\t", text)
    for f in funcs:
        name = f.split("\n", 1)[0].strip()
        if want in name:
            break
    else:
        sys.exit("no kernel matches " + want)
    lines = f.split("\n")
    first = sys.argv[2] if len(sys.argv) > 2 else None
    last = sys.argv[3] if len(sys.argv) > 3 else None
    on = first is None
    label, valu, fast, lds, mem, total = "(entry)", 0, 0, 0, 0, [0, 0]
    def flush(end=""):
        nonlocal valu, fast, lds, mem
        if on and (valu or lds or mem):
            print("%-14s valu %4d (of them v_mov / f32 add / mul / int add / and: %3d)  lds %2d  mem %2d  %s" % (label, valu, fast, lds, mem, end))
            total[0] += valu; total[1] += fast
        valu = fast = lds = mem = 0
    for ln in lines:
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            flush()
            label = m.group(1)
            if first and label == first:
                on = True
            if last and label == last:
                on = False
            continue
        t = ln.strip()
        if re.match(r"^; %bb\.", t):
            flush()
            label = t.split()[1]
            continue
        op = t.split(" ")[0].split("\t")[0]
        if op.startswith("v_"):
            valu += 1
            if re.match(r"v_(mov_b32|mov_b64|add_f32|sub_f32|subrev_f32|mul_f32|add_u32|sub_u32|subrev_u32|and_b32|or_b32|xor_b32|lshlrev_b32|lshrrev_b32)", op):
                fast += 1
        elif op.startswith("ds_"):
            lds += 1
        elif op.startswith(("global_", "buffer_", "scratch_", "flat_")):
            mem += 1
        elif op.startswith(("s_cbranch", "s_branch")):
            flush(t)
    flush()
    print("total valu %d (fast %d)" % tuple(total))

main()

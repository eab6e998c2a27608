#!/usr/bin/env python3
"""Static per-basic-block instruction mix of one kernel in hipcc's -S output (VALU / SALU / LDS / VMEM / branch counts per block).
usage: isa_blocks.py file.s kernel_symbol [min_valu]
Blocks are true basic blocks: a label starts one, a branch ends one ("LABEL+n" = the part of a labelled region n lines below its label)."""
import re
import sys

path, sym = sys.argv[1], sys.argv[2]
minv = int(sys.argv[3]) if len(sys.argv) > 3 else 20
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
blocks, cur = [], ["entry", start, {}, []]
base_line = start
for i in range(start + 1, end):
    l = lines[i]
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = [m.group(1), i, {}, []]
        base_line = i
        continue
    t = l.strip().split()
    if not t or t[0].startswith(";") or t[0].startswith("."):
        continue
    op = t[0]
    k = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith("s_waitcnt") and not op.startswith("s_cbranch") and not op.startswith("s_branch") and not op.startswith("s_barrier")
         else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "wait" if op.startswith("s_waitcnt") else "br" if "branch" in op else "bar" if op.startswith("s_barrier") else "other")
    cur[2][k] = cur[2].get(k, 0) + 1
    cur[2].setdefault("ops", {})
    cur[2]["ops"][op] = cur[2]["ops"].get(op, 0) + 1
    if "branch" in op:
        # a branch ends the basic block: what follows it in the same labelled region runs only on fall-through (counting a whole labelled region as
        # one block once made 342 instructions of a skipped border path look like per-level work of k_lk_q)
        cur[3].append(t[1] if len(t) > 1 else "?")
        blocks.append(cur)
        cur = [cur[0].split("+")[0] + "+%d" % (i - (cur[1] if "+" not in cur[0] else base_line)), i, {}, []]
blocks.append(cur)
tot = {}
for b in blocks:
    for k, v in b[2].items():
        if k != "ops":
            tot[k] = tot.get(k, 0) + v
print("total", tot, "blocks", len(blocks))
for b in blocks:
    c = b[2]
    if c.get("valu", 0) >= minv:
        top = sorted(c["ops"].items(), key=lambda kv: -kv[1])[:8]
        print(f"{b[0]:12s} line {b[1]:6d} valu {c.get('valu',0):5d} salu {c.get('salu',0):4d} lds {c.get('lds',0):4d} vmem {c.get('vmem',0):3d} wait {c.get('wait',0):3d} bar {c.get('bar',0)} br->{b[3]} :: {top}")

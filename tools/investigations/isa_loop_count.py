#!/usr/bin/env python3
"""Instruction mix of the hot loop of a kernel from the compiler's assembly (`hipcc -save-temps`): the innermost backward-branch loop
that contains a marker instruction.  python tools/investigations/isa_loop_count.py <file.s> <kernel-symbol-substring> <marker-instruction> [rows per trip]"""
import collections
import re
import sys

path, sym, marker = sys.argv[1:4]
rows = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(sym) + r"\S*:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
loops = []
for i, l in enumerate(body):
    m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
# the loop with the most markers; among those the shortest (innermost)
cands = sorted((-sum(marker in l for l in body[a:b]), b - a, a, b) for a, b in loops)
_, _, a, b = cands[0]
ins = [l.split()[0] for l in body[a:b + 1] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
c = collections.Counter(ins)
valu = sum(v for k, v in c.items() if k.startswith("v_"))
print(f"{sym}: innermost loop with {marker}: {len(ins)} instructions, {valu} VALU, per row {len(ins) / rows:.1f} / {valu / rows:.1f}")
groups = collections.Counter()
for k, v in c.items():
    g = ("f64" if "f64" in k else "sqrt" if "sqrt" in k else "dpp" if "dpp" in k else "salu" if k.startswith("s_") else
         "mem" if k.startswith(("global_", "buffer_", "ds_", "flat_", "scratch_")) else "valu")
    groups[g] += v
print(dict(groups))
for k, v in c.most_common(45):
    print(f"  {k:30s}{v:5d}  {v / rows:7.2f}")

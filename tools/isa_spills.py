#!/usr/bin/env python3
"""Where a kernel's register spills sit: every scratch access of one kernel in a gfx950 assembly dump (hipcc --save-temps)
with the spans of the loops around it (instructions from the loop's head to its back edge; small = inner loop).
    tools/isa_spills.py file.s <mangled prefix>"""
import sys

s = open(sys.argv[1]).read().split("\n")
start = [i for i, l in enumerate(s) if l.startswith(sys.argv[2]) and ":" in l][0]
end = [i for i, l in enumerate(s) if i > start and l.strip().startswith("s_endpgm")][0]
body = s[start:end]
pos = {}
for i, l in enumerate(body):
    t = l.split(";")[0].strip()
    if t.endswith(":"):
        pos[t[:-1]] = i
loops = []
for i, l in enumerate(body):
    t = l.split(";")[0].strip()
    if t.startswith("s_cbranch") or t.startswith("s_branch"):
        tgt = t.split()[-1]
        if tgt in pos and pos[tgt] < i:
            loops.append((pos[tgt], i))
n = 0
for i, l in enumerate(body):
    if "scratch_" in l:
        n += 1
        inner = sorted(b - a for a, b in loops if a <= i <= b)
        print(i, l.strip()[:58], "loops", inner[:4])
print("scratch accesses", n, "of", len(body), "lines")

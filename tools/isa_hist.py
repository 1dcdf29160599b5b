#!/usr/bin/env python3
"""Instruction histogram of one kernel in a gfx950 assembly dump (hipcc --save-temps): tools/isa_hist.py file.s <mangled prefix>."""
import collections
import sys

s = open(sys.argv[1]).read().split("\n")
pref = sys.argv[2]
start = [i for i, l in enumerate(s) if l.startswith(pref) and ":" in l.split(";")[0]][0]
end = [i for i, l in enumerate(s) if i > start and l.strip().startswith("s_endpgm")][0]
c = collections.Counter()
for l in s[start:end]:
    l = l.split(";")[0].strip()
    if not l or l.startswith(".") or l.endswith(":"):
        continue
    c[l.split()[0]] += 1
print("instructions", sum(c.values()))
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print(f"  {k:32s} {v}")
print("scratch", sum(v for k, v in c.items() if "scratch" in k), "flat", sum(v for k, v in c.items() if k.startswith("flat")),
      "global", sum(v for k, v in c.items() if k.startswith("global")), "ds", sum(v for k, v in c.items() if k.startswith("ds_")),
      "s_load", sum(v for k, v in c.items() if k.startswith("s_load")), "valu", sum(v for k, v in c.items() if k.startswith("v_")),
      "salu", sum(v for k, v in c.items() if k.startswith("s_")))

#!/usr/bin/env python3
"""Static instruction table of one kernel by the source function (and, inside k_paths, the phase of its wave loop) its
instructions were compiled from.

    hipcc ... -gline-tables-only -save-temps=obj ...        (line tables only: the code is the release code)
    python tools/isa_by_source.py <file.s> <mangled kernel prefix> [--top-opcodes N]

Every instruction carries the `.loc` of the innermost inlined function it came from. Lines are mapped to functions by
the `PT_DEV` / `__global__` definitions in csrc/*.h; the body of k_paths is cut at its phase comments (A. refill,
B. shade, C. trace). Output: per function / phase the VALU, SALU, LDS, vector-memory and scalar-memory instruction counts,
and the most frequent VALU opcodes of the classes the SQ counters lump together as "other" (compares, selects, moves,
min/max, conversions, bit operations)."""
import argparse
import collections
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "path_tracing_amd" / "csrc"

COUNTED = ("v_fma", "v_fmac", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_rcp", "v_sqrt", "v_rsq", "v_add_u32", "v_sub_u32", "v_mul_lo", "v_mul_hi",
           "v_mad_u", "v_lshl", "v_lshr", "v_and_b32", "v_or_b32", "v_xor", "v_add_co", "v_addc", "v_subrev", "v_pk_")


def functions_of(path):
    """[(first line, name)] of the function definitions in a source file, k_paths split at its phase comments."""
    out = []
    lines = path.read_text().split("\n")
    for i, l in enumerate(lines, 1):
        m = re.match(r"\s*(?:template\s*<[^>]*>\s*)?(?:PT_DEV|__global__|__host__ __device__|__device__|static inline)\b[^(]*?\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", l)
        if m and not l.strip().startswith("//") and "#define" not in l and m.group(1) not in ("__launch_bounds__", "__attribute__"):
            out.append((i, m.group(1)))
        elif re.match(r"\s*__global__ __launch_bounds__", l):
            k = re.search(r"void\s+([A-Za-z_][A-Za-z0-9_]*)\s*\(", l)
            if k:
                out.append((i, k.group(1)))
        if "// ---- A. refill" in l:
            out.append((i, "k_paths: A refill"))
        if "// ---- B. shade" in l:
            out.append((i, "k_paths: B shade"))
        if "// ---- C. trace" in l:
            out.append((i, "k_paths: C trace"))
    return sorted(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("prefix")
    ap.add_argument("--top-opcodes", type=int, default=12)
    a = ap.parse_args()
    s = open(a.asm).read().split("\n")
    files = {}
    for l in s:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    start = [i for i, l in enumerate(s) if l.startswith(a.prefix) and ":" in l.split(";")[0]][0]
    end = [i for i, l in enumerate(s) if i > start and l.strip().startswith("s_endpgm")][0]
    fn_tables = {p.name: functions_of(p) for p in CSRC.glob("*.h")}

    def where(fname, line):
        tab = fn_tables.get(fname)
        if not tab:
            return fname
        name = fname
        for first, n in tab:
            if first <= line:
                name = n
            else:
                break
        return name

    cur = ("?", 0)
    per = collections.defaultdict(collections.Counter)
    other_ops = collections.Counter()
    for l in s[start:end]:
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            continue
        t = l.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        w = where(*cur)
        cls = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") and not op.startswith(("s_load", "s_buffer", "s_waitcnt", "s_nop")) else
               "lds" if op.startswith("ds_") else "smem" if op.startswith(("s_load", "s_buffer")) else
               "scratch" if op.startswith("scratch") else "vmem" if op.startswith(("global_", "flat_", "buffer_")) else "wait/nop")
        per[w][cls] += 1
        if cls == "valu" and not op.startswith(COUNTED):
            other_ops[op] += 1
    tot = collections.Counter()
    print(f"{'function / phase':34s} {'valu':>6s} {'salu':>6s} {'lds':>5s} {'vmem':>5s} {'smem':>5s} {'scratch':>7s}")
    for w, c in sorted(per.items(), key=lambda kv: -kv[1]["valu"]):
        print(f"{w[:34]:34s} {c['valu']:6d} {c['salu']:6d} {c['lds']:5d} {c['vmem']:5d} {c['smem']:5d} {c['scratch']:7d}")
        tot.update(c)
    print(f"{'total':34s} {tot['valu']:6d} {tot['salu']:6d} {tot['lds']:5d} {tot['vmem']:5d} {tot['smem']:5d} {tot['scratch']:7d}")
    print("\nVALU opcodes outside the counted classes (FMA / MUL / ADD / transcendental / INT32), static count:")
    for op, n in other_ops.most_common(a.top_opcodes):
        print(f"   {op:28s} {n}")


if __name__ == "__main__":
    main()

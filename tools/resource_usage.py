#!/usr/bin/env python3
"""Registers, scratch and occupancy of every kernel of the library, as the compiler reports them.

    python tools/resource_usage.py [out.txt]

Compiles path_tracing_amd/csrc/pt_amd.hip with -Rpass-analysis=kernel-resource-usage (nothing is installed) and
prints one line per kernel; the profile's launch records (VGPR_Count etc.) can be compared with these.
"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
KEYS = [("sgpr", "TotalSGPRs"), ("vgpr", "VGPRs"), ("agpr", "AGPRs"), ("scratch_B_per_lane", r"ScratchSize \[bytes/lane\]"),
        ("waves_per_simd", r"Occupancy \[waves/SIMD\]"), ("sgpr_spill", "SGPRs Spill"), ("vgpr_spill", "VGPRs Spill"),
        ("static_lds_B", r"LDS Size \[bytes/block\]")]


def main():
    if len(sys.argv) > 1 and not sys.argv[1].endswith(".txt"):
        raise SystemExit("the argument is the file the table is written to (something.txt)")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
                        "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/pt_resource_usage.so",
                        str(ROOT / "path_tracing_amd" / "csrc" / "pt_amd.hip")], capture_output=True, text=True)
    txt = r.stdout + r.stderr
    demangle = subprocess.run(["c++filt"], input=txt, capture_output=True, text=True).stdout
    lines = []
    for b in re.split(r"remark: Function Name: ", demangle)[1:]:
        name = b.split(" [-Rpass")[0].strip()
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        vals = []
        for label, key in KEYS:
            m = re.search(key + r": (\d+)", b)
            vals.append(f"{label} {m.group(1) if m else '?'}")
        lines.append(f"{name:44s} " + "  ".join(vals))
    out = "\n".join(lines) + "\n"
    sys.stdout.write(out)
    if len(sys.argv) > 1:
        Path(sys.argv[1]).write_text(out)


if __name__ == "__main__":
    main()

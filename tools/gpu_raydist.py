#!/usr/bin/env python3
"""Distribution of the per-pixel ray counts (how uneven the pixels of a frame are)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from path_tracing_amd import abi, scenes

def main():
    for name, sc in (("boxes", scenes.boxes(1920, 1080, 64, 10)), ("cornell", scenes.cornell_box(512, 512, 16, 10))):
        ctx = abi.Context()
        ctx.upload_scene(sc)
        fr = sc.frame()
        ctx.render(fr)
        rc = ctx.read_aux(abi.PT_AUX_RAY_COUNT).astype(np.int64).ravel()
        own = rc - (sc.samples - 1)           # rays k_paths + k_primary traced for the pixel
        listed = own[own > 1]
        print(name, "pixels", rc.size, "listed", listed.size, "sum", int(own.sum()))
        print("  percentiles 50/90/99/99.9/max:", [int(np.percentile(listed, p)) for p in (50, 90, 99, 99.9)], int(listed.max()))
        h, e = np.histogram(listed, bins=16)
        print("  hist", h.tolist(), [int(x) for x in e])
        print("  mean", float(listed.mean()), "lanes-needed-for-balance", float(listed.sum() / listed.max()))
        ctx.close()
main()

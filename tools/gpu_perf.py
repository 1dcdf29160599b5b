#!/usr/bin/env python3
"""Kernel timings of the four workload shapes for one or more builds of the library.

    python tools/gpu_perf.py [--libs a.so b.so ...] [--reps 5] [--scenes boxes cornell glb inst]

Each library is timed in its own process (PT_AMD_LIBRARY). Prints median/min trace_ms and a checksum of the image
bits so that a change of results is visible next to the timing.
"""
import argparse
import os
import subprocess
import sys
import zlib
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

SCENES = {
    "boxes": ("boxes", dict(width=1920, height=1080, samples=64)),
    "cornell": ("cornell", dict(width=512, height=512, samples=16)),
    "cornell_hd": ("cornell", dict(width=1920, height=1080, samples=64)),
    "glb": ("scene_glb_textured", dict(width=1920, height=1080, samples=16)),
    "inst": ("instanced_boxes", dict(width=1920, height=1080, samples=4)),
}


def run_one(names, reps):
    from path_tracing_amd import abi, scenes
    out = []
    for n in names:
        cfg, kw = SCENES[n]
        scene = scenes.CONFIGS[cfg](**kw)
        frame = scene.frame()
        with abi.Context() as ctx:
            ctx.upload_scene(scene)
            img = ctx.render(frame)
            ms = []
            for _ in range(reps):
                ctx.render(frame)
                ms.append(ctx.stats()["trace_ms"])
            st = ctx.stats()
        out.append(f"{n}: med {np.median(ms):7.3f} min {min(ms):7.3f} ms  {st['rays'] / np.median(ms) / 1e3:9.1f} Mrays/s  crc {zlib.crc32(img.tobytes()):08x}")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", nargs="*", default=[])
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--scenes", nargs="*", default=list(SCENES))
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child or not a.libs:
        for line in run_one(a.scenes, a.reps):
            print("   ", line, flush=True)
        return
    for lib in a.libs:
        print(lib, flush=True)
        env = dict(os.environ, PT_AMD_LIBRARY=str(Path(lib).resolve()))
        subprocess.run([sys.executable, __file__, "--child", "--reps", str(a.reps), "--scenes", *a.scenes], env=env, check=False)


if __name__ == "__main__":
    main()

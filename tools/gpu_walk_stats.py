#!/usr/bin/env python3
"""Per-ray work of the closest-hit walk on one scene: box tests, object tests, triangle tests per ray (detailed statistics
of one frame) and the frame's kernel time without them.

    python tools/gpu_walk_stats.py [inst|glb|boxes ...] [--spp N]

Environment knobs of the library (PT_AMD_TLAS_BUILDER=lbvh ...) apply as usual; run once per setting to compare."""
import argparse
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from path_tracing_amd import abi, scenes  # noqa: E402
from tools.gpu_perf import SCENES  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("scenes", nargs="*", default=["inst"])
ap.add_argument("--spp", type=int, default=16)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
for n in a.scenes:
    cfg, kw = SCENES[n]
    kw = dict(kw, samples=a.spp)
    scene = scenes.CONFIGS[cfg](**kw)
    frame = scene.frame()
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        ctx.render(frame)
        ms = []
        for _ in range(a.reps):
            ctx.render(frame)
            ms.append(ctx.stats()["trace_ms"])
        build_ms = ctx.stats()["build_ms"]
        ctx.set_stats_detail(True)
        ctx.render(frame)
        s = ctx.stats()
        acc = ctx.check_accel()
    rays = max(s["rays"], 1)
    print(f"{n} {a.spp} spp: {np.median(ms):8.3f} ms (min {min(ms):.3f})  {rays / np.median(ms) / 1e3:8.1f} Mrays/s | per ray: box tests {s['box_tests'] / rays:6.2f}  "
          f"object tests {s['xforms'] / rays:5.2f}  triangle tests {s['tri_tests'] / rays:6.2f} | lanes tracing per round {s['lane_rounds'] / max(s['wave_rounds'], 1):5.1f} | "
          f"TLAS levels {acc['tlas_depth']}  last TLAS build {build_ms:.2f} ms  errors {acc['errors']}", flush=True)

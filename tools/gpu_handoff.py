#!/usr/bin/env python3
"""Cost of handing the image to the host: wall time of pt_render (kernels + device-to-host copy) next to the kernels' time."""
import sys
import time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from path_tracing_amd import abi, scenes

sc = scenes.boxes(1920, 1080, 64, 10)
with abi.Context() as ctx:
    ctx.upload_scene(sc)
    cases = ((0, False, "f32, pageable"), (0, True, "f32, page-locked"), (abi.PT_FRAME_OUTPUT_F16, False, "f16, pageable"),
             (abi.PT_FRAME_OUTPUT_F16, True, "f16, page-locked"))
    for flags, locked, name in cases:
        fr = sc.frame(flags=flags)
        dt = np.float16 if flags & abi.PT_FRAME_OUTPUT_F16 else np.float32
        staging = ctx.host_array((fr.height, fr.width, 4), dt) if locked else np.empty((fr.height, fr.width, 4), dtype=dt)
        for _ in range(3):
            ctx.render(fr, out=staging)
        walls, kern = [], []
        for _ in range(10):
            t0 = time.perf_counter()
            img = ctx.render(fr, out=staging)
            walls.append((time.perf_counter() - t0) * 1e3)
            kern.append(ctx.stats()["trace_ms"])
        ho = np.median(walls) - np.median(kern)
        print(f"{name}: pt_render wall {np.median(walls):.3f} ms, kernels {np.median(kern):.3f} ms, hand-off {ho:.3f} ms, "
              f"{img.nbytes / 1e6:.1f} MB -> {img.nbytes / 1e9 / max(ho * 1e-3, 1e-9):.1f} GB/s")

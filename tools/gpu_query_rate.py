#!/usr/bin/env python3
"""Closest-hit queries alone (k_query) on the 83,334-object scene: random rays inside the cloud, several batches.
Run under `rocprofv3 --kernel-trace --stats` for the kernel's own duration; the wall time per call printed here
includes both copies over PCIe."""
import sys
import time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from path_tracing_amd import abi, scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
sc = scenes.instanced_boxes(83334, 64, 64, 1, 10)
rng = np.random.default_rng(7)
rays = np.zeros(n, dtype=abi.RAY_DT)
rays["pos"] = rng.uniform(-40, 40, (n, 3)).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32)
rays["dir"] = d / np.linalg.norm(d, axis=1, keepdims=True)
with abi.Context() as ctx:
    ctx.upload_scene(sc)
    for k in range(4):
        t0 = time.perf_counter()
        hits = ctx.query(rays)
        dt = time.perf_counter() - t0
        print(f"batch {k}: {n} rays, call {dt * 1e3:.1f} ms, hit fraction {(hits['object'] != 0xFFFFFFFF).mean():.3f}, mean t of hits {hits['t'][hits['object'] != 0xFFFFFFFF].mean():.2f}", flush=True)

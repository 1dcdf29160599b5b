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
if len(sys.argv) > 2 and sys.argv[2] == "sorted":
    # the same rays ordered by direction octant, then by the cell of their origin (cells of `cell` units, Morton order)
    cell = float(sys.argv[3]) if len(sys.argv) > 3 else 5.0
    q = np.clip(((rays["pos"] + 40.0) / cell).astype(np.uint32), 0, 1023)
    def spread(v):
        v = v.astype(np.uint64)
        v = (v | (v << 16)) & 0x030000FF0000FF
        v = (v | (v << 8)) & 0x0300F00F00F00F
        v = (v | (v << 4)) & 0x030C30C30C30C3
        v = (v | (v << 2)) & 0x09249249249249
        return v
    morton = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
    octant = ((rays["dir"][:, 0] < 0).astype(np.uint64) | ((rays["dir"][:, 1] < 0).astype(np.uint64) << 1) | ((rays["dir"][:, 2] < 0).astype(np.uint64) << 2))
    rays = rays[np.argsort((octant << np.uint64(40)) | morton, kind="stable")]
    print(f"rays sorted by octant and {cell}-unit origin cell")
with abi.Context() as ctx:
    ctx.upload_scene(sc)
    for k in range(4):
        t0 = time.perf_counter()
        hits = ctx.query(rays)
        dt = time.perf_counter() - t0
        print(f"batch {k}: {n} rays, call {dt * 1e3:.1f} ms, hit fraction {(hits['object'] != 0xFFFFFFFF).mean():.3f}, mean t of hits {hits['t'][hits['object'] != 0xFFFFFFFF].mean():.2f}", flush=True)

#!/usr/bin/env python3
"""TLAS build vs refit time for the 83,334-object scene (device time of pt_upload_objects' kernels)."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from path_tracing_amd import abi, scenes

sc = scenes.instanced_boxes(83334, 1920, 1080, 1, 10)
with abi.Context() as ctx:
    ctx.upload_scene(sc)
    fr = sc.frame()
    ctx.render(fr)
    print(f"first upload (fresh build): device {ctx.stats()['build_ms']:.3f} ms")
    objs = sc.objects.copy()
    for k in range(4):
        objs["local_to_world"][k * 1000, 13] += 0.25   # one object moves: an identical array would be a no-op
        t0 = time.perf_counter()
        ctx.upload_objects(objs)
        wall = (time.perf_counter() - t0) * 1e3
        ctx.render(fr)
        st = ctx.stats()
        print(f"upload {k}: refit={bool(st['accel_used'] & 4)} device {st['build_ms']:.3f} ms, call {wall:.2f} ms, frame {st['trace_ms']:.3f} ms")

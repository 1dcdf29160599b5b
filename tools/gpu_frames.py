#!/usr/bin/env python3
"""Render a few frames of one workload (a plain target for profilers): python tools/gpu_frames.py [scene] [frames]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from path_tracing_amd import abi, scenes
from tools.gpu_perf import SCENES
name = sys.argv[1] if len(sys.argv) > 1 else "boxes"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg, kw = SCENES[name]
scene = scenes.CONFIGS[cfg](**kw)
with abi.Context() as ctx:
    ctx.upload_scene(scene)
    for _ in range(n):
        ctx.render(scene.frame())
    print(name, ctx.stats()["trace_ms"])

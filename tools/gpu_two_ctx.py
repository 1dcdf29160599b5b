#!/usr/bin/env python3
"""Frames in flight: one context on one stream against two contexts on two streams taking alternate frames."""
import sys
import time
from pathlib import Path
import torch
torch.cuda.init()
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from path_tracing_amd import abi, scenes

dev = torch.device("cuda", 0)
sc = scenes.boxes(1920, 1080, 64, 10)
fr = sc.frame()
for n_ctx in (1, 2, 3):
    ctxs = [abi.Context() for _ in range(n_ctx)]
    for c in ctxs:
        c.upload_scene(sc)
    streams = [torch.cuda.Stream(dev) for _ in range(n_ctx)]
    outs = [torch.zeros((fr.height, fr.width, 4), dtype=torch.float32, device=dev) for _ in range(n_ctx)]
    def step(i):
        k = i % n_ctx
        ctxs[k].render_device(fr, outs[k].data_ptr(), streams[k].cuda_stream)
    for i in range(4 * n_ctx):
        step(i)
    torch.cuda.synchronize(dev)
    steps = 60
    t0 = time.perf_counter()
    for i in range(steps):
        step(i)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    rays = ctxs[0].stats()["rays"]
    same = all(torch.equal(outs[0], o) for o in outs)
    print(f"{n_ctx} context(s): {dt * 1e3:.3f} ms per frame, {rays / dt / 1e6:.0f} Mrays/s, images equal {same}")
    for c in ctxs:
        c.close()

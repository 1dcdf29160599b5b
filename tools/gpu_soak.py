#!/usr/bin/env python3
"""A soak run of one context: object uploads, frames on slots, synchronous frames and ray queries in a seeded random
order, every finished image compared with the one a fresh context renders for the same (object array, time).

    python tools/gpu_soak.py [scene: inst | boxes | glb] [iterations] [seed]

Prints a line every 200 iterations; exits 1 at the first image that differs, 3 when a call does not return for 30 s."""
import faulthandler
import os
import sys
import threading
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from path_tracing_amd import abi, scenes  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "inst"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
g = np.random.Generator(np.random.PCG64(seed))
scene = {"inst": lambda: scenes.instanced_boxes(n_objects=2000, width=320, height=200, samples=8, extent=9.0),
         "boxes": lambda: scenes.boxes(320, 200, 16, 10),
         "glb": lambda: scenes.scene_glb_textured(320, 200, 8, 10)}[name]()
TIMES = [0.5, 1.5, 2.5, 3.5]
N_ARRAYS = 5
K = 4


def moved(k):
    o = scene.objects.copy()
    if k:
        o["local_to_world"][:, 12] += np.float32(0.04 * k)
        for i in range(o.size):
            l2w = o["local_to_world"][i].reshape(4, 4).T.astype(np.float64)
            o["world_to_local"][i] = np.linalg.inv(l2w).T.reshape(16).astype(np.float32)
    return o


arrays = [moved(k) for k in range(N_ARRAYS)]
want = {}
for a in range(N_ARRAYS):
    with abi.Context() as c:
        c.upload_scene(scene)
        c.upload_objects(arrays[a])
        for t in TIMES:
            want[(a, t)] = c.render(scene.frame(time=t))
print(f"{name}: {len(want)} expected images", flush=True)

state = {"t": time.time(), "what": "start"}


def watchdog():
    while True:
        time.sleep(1.0)
        if time.time() - state["t"] > 30.0:
            print(f"STUCK: {state}", flush=True)
            faulthandler.dump_traceback()
            os._exit(3)


threading.Thread(target=watchdog, daemon=True).start()
rays = np.zeros(512, dtype=abi.RAY_DT)
rays["pos"] = g.uniform(-3, 3, (512, 3)).astype(np.float32)
d = g.normal(size=(512, 3))
rays["dir"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
counts = {"upload": 0, "submit": 0, "sync": 0, "query": 0, "checked": 0}
with abi.Context() as ctx:
    ctx.upload_scene(scene)
    cur = 0
    bufs = [ctx.host_array(want[(0, TIMES[0])].shape, np.float32) for _ in range(K)]
    pending = [None] * K     # (array, time) of the frame on the slot
    nxt = 0

    def take(s):
        state.update(t=time.time(), what=f"wait {s}")
        ctx.wait(s)
        if not np.array_equal(bufs[s].view(np.uint32), want[pending[s]].view(np.uint32)):
            print(f"MISMATCH: slot {s}, frame {pending[s]}, after {counts}", flush=True)
            os._exit(1)
        counts["checked"] += 1
        pending[s] = None

    for it in range(iters):
        r = g.uniform()
        if r < 0.3:
            cur = int(g.integers(0, N_ARRAYS))
            state.update(t=time.time(), what=f"upload {cur}")
            ctx.upload_objects(arrays[cur])
            counts["upload"] += 1
        elif r < 0.85:
            s = nxt
            nxt = (nxt + 1) % K
            if pending[s] is not None:
                take(s)
            t = TIMES[int(g.integers(0, len(TIMES)))]
            state.update(t=time.time(), what=f"submit {s}")
            ctx.submit(scene.frame(time=t), s, bufs[s])
            pending[s] = (cur, t)
            counts["submit"] += 1
        elif r < 0.95:
            t = TIMES[int(g.integers(0, len(TIMES)))]
            state.update(t=time.time(), what="render")
            img = ctx.render(scene.frame(time=t))
            if not np.array_equal(img.view(np.uint32), want[(cur, t)].view(np.uint32)):
                print(f"MISMATCH: synchronous frame {(cur, t)}, after {counts}", flush=True)
                os._exit(1)
            counts["sync"] += 1
            counts["checked"] += 1
        else:
            state.update(t=time.time(), what="query")
            ctx.query(rays)
            counts["query"] += 1
        if it % 200 == 199:
            print(f"{it + 1} iterations: {counts}", flush=True)
    for s in range(K):
        if pending[s] is not None:
            take(s)
    state["t"] = time.time() + 1e9
    err = ctx.check_accel()["errors"]
print(f"ok: {counts}, accel errors {err}", flush=True)
sys.exit(0 if err == 0 else 1)

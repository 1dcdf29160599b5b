#!/usr/bin/env python3
"""Object-list scenes of different character, one frame at a time (frame times from the library's own HIP events):
a 1080p scene of twelve overlapping meshes whose every pixel hits geometry (far more work than the launch has waves),
and a close-up of the scene.glb mesh (a few times the work of the benchmark view, still bound by its longest chains).
Prints six successive frames each; the first has no schedule hints. Switches are read from the environment
(PT_AMD_SPLIT_CHUNKS, PT_AMD_SPLIT_LOAD, PT_AMD_LIST_REGS, PT_AMD_GROUP_MARGIN ...), PT_AMD_DEBUG=1 adds what
k_chunk_order saw for every frame."""
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from path_tracing_amd import abi, scenes


def main():
    busy = scenes.overlapping_meshes(1920, 1080, 64, 8)
    close = scenes.scene_glb_textured(1920, 1080, 64, 10)
    eye = np.asarray(close.camera.eye, dtype=np.float64)
    tgt = np.asarray(close.camera.target, dtype=np.float64)
    close.camera = scenes.Camera(tuple(tgt + (eye - tgt) * 0.35), tuple(tgt), close.camera.fov_y)
    for name, scene in (("twelve overlapping meshes, 1080p, 64 spp", busy), ("scene.glb close-up, 1080p, 64 spp", close)):
        with abi.Context() as ctx:
            ctx.upload_scene(scene)
            ms = []
            for i in range(6):
                ctx.render(scene.frame(time=0.5 + i))
                ms.append(ctx.stats()["trace_ms"])
            st = ctx.stats()
        print(os.environ.get("LABEL", ""), name, "rays", st["rays"], "ms per frame", [round(t, 2) for t in ms])


if __name__ == "__main__":
    main()

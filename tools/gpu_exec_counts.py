#!/usr/bin/env python3
"""How often a k_paths wave executes each major block per round (needs a library built with -DPT_EXEC_COUNTS,
PT_AMD_LIBRARY pointing at it)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from path_tracing_amd import abi, scenes
from tools.gpu_perf import SCENES
NAMES = ["shade loop pass", "step 1 (any lane)", "add_emitted (hit)", "next_sample in step 1", "add_emitted (restart)", "scatter (brdf)",
         "next_sample after scatter", "trace executed", "world box test (wave)", "object pass", "triangle loop iteration", "exact gate", "finalize",
         "TLAS: lanes in node visits | list: cycles in the box loop", "TLAS: lanes in leaf calls | list: cycles in candidate passes",
         "global tri loop: lanes inside | list: cycles in hit records"]
# (TLAS scenes reuse five rows: "world box test" = node-visit iterations of the resumable walk, "object pass" = its leaf calls,
#  "triangle loop iteration" = iterations of the global-memory triangle loop, "trace executed" / "exact gate" = node-visit
#  iterations that ran with at most 16 / at most 32 lanes)
for n in (sys.argv[1:] or ["boxes"]):
    cfg, kw = SCENES[n]
    scene = scenes.CONFIGS[cfg](**kw)
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        ctx.set_stats_detail(True)
        ctx.render(scene.frame()); ctx.render(scene.frame())
        s = ctx.stats()
    rounds = max(s["wave_rounds"], 1)
    print(n, "rounds", rounds, "lanes/round", s["lane_rounds"] / rounds)
    for k, name in enumerate(NAMES):
        print(f"   {name:28s} {s['wave_life_hist'][16 + k] / rounds:8.3f} per round")

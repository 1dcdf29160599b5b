#!/usr/bin/env python3
"""Phase breakdown of the frame kernel (STATS variant): cycles in refill / closest hit / shading, rounds, lane occupancy."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from path_tracing_amd import abi, scenes
from tools.gpu_perf import SCENES
for n in (sys.argv[1:] or list(SCENES)):
    cfg, kw = SCENES[n]
    scene = scenes.CONFIGS[cfg](**kw)
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        ctx.set_stats_detail(True)
        ctx.render(scene.frame()); ctx.render(scene.frame())
        s = ctx.stats()
    print("   life histogram (0.5M-cycle buckets):", s["wave_life_hist"])
    tot = s["cycles_refill"] + s["cycles_trace"] + s["cycles_shade"]
    print(f"{n}: ms {s['trace_ms']:.3f} rays {s['rays']} rounds {s['wave_rounds']} lanes/round {s['lane_rounds'] / max(s['wave_rounds'], 1):.1f} "
          f"refill {100 * s['cycles_refill'] / tot:.1f}% trace {100 * s['cycles_trace'] / tot:.1f}% shade {100 * s['cycles_shade'] / tot:.1f}% "
          f"primary_ms {s['primary_ms']:.3f} paths_ms {s['paths_ms']:.3f} waves {s['waves']} life mean {s['wave_life_sum'] / max(s['waves'], 1) / 1e6:.2f}M max {s['wave_life_max'] / 1e6:.2f}M cyc "
          f"cycles/round {tot / max(s['wave_rounds'], 1):.0f} xforms/ray {s['xforms'] / s['rays']:.2f} box/ray {s['box_tests'] / s['rays']:.2f} tri/ray {s['tri_tests'] / s['rays']:.2f} shades/ray {s['shades'] / s['rays']:.2f} "
          f"object passes/round {s['object_passes'] / max(s['wave_rounds'], 1):.2f} lanes/pass {(s['xforms'] - s['primary_xforms']) / max(s['object_passes'], 1):.1f}")

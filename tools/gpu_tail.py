#!/usr/bin/env python3
"""The waves that end a frame of the scene.glb workload (1080p, 256 spp, one frame at a time): three plain frames, then
one with detailed statistics; from its per-wave records the last waves to finish with their rounds, cycles per round,
active lanes per round and how many of their trace steps ran with four lanes to a ray."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from path_tracing_amd import abi, scenes


def main():
    sc = scenes.scene_glb_textured(1920, 1080, 256, 10)
    with abi.Context() as ctx:
        ctx.upload_scene(sc)
        for i in range(3):
            ctx.render(sc.frame(time=0.5 + i))
            print("plain frame ms", round(ctx.stats()["trace_ms"], 2))
        ctx.set_stats_detail(True)
        ctx.render(sc.frame(time=3.5))
        s = ctx.stats()
        log = ctx.wave_log()
    tot = s["cycles_refill"] + s["cycles_trace"] + s["cycles_shade"]
    print(f"instrumented frame ms {s['trace_ms']:.2f}  wave rounds {s['wave_rounds']}  lanes/round {s['lane_rounds'] / max(s['wave_rounds'], 1):.1f}  "
          f"refill {100 * s['cycles_refill'] / tot:.1f}%  trace {100 * s['cycles_trace'] / tot:.1f}%  shade {100 * s['cycles_shade'] / tot:.1f}%  "
          f"cycles/round {tot / max(s['wave_rounds'], 1):.0f}")
    lg = log[log[:, 3] > 0]
    t0 = int(lg[:, 0].min())
    end = (lg[:, 1].astype(np.int64) - t0) / 100.0
    print("waves", lg.shape[0], " of them with four-lane trace steps", int((lg[:, 12] > 0).sum()))
    print("last 12 waves: end us | rounds | cycles/round | lanes/round | items | longest item: cycles/round, rounds | four-lane steps")
    for i in np.argsort(-end)[:12]:
        r = lg[i]
        print(f"  {end[i]:9.1f} {r[3]:6d} {r[2] * 16 / r[3]:8.0f} {r[4] / r[3]:5.1f} {r[6] >> 8:4d}   {r[8] * 16 / max(r[9], 1):8.0f} {r[9]:6d}  {r[12]:6d}")
    for name, m in (("waves with four-lane steps", lg[:, 12] > 0), ("other waves", lg[:, 12] == 0)):
        if m.any():
            q = lg[m]
            print(f"{name}: cycles/round mean {float((q[:, 2] * 16 / q[:, 3]).mean()):.0f}  rounds mean {float(q[:, 3].mean()):.0f} max {int(q[:, 3].max())}")


if __name__ == "__main__":
    main()

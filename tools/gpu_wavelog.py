#!/usr/bin/env python3
"""Per-wave lifetimes of the path kernel (detailed statistics build): where the tail of the launch comes from."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from path_tracing_amd import abi, scenes

def main():
    sc = scenes.boxes(1920, 1080, 64, 10)
    ctx = abi.Context()
    ctx.upload_scene(sc)
    fr = sc.frame()
    ctx.set_stats_detail(True)
    ctx.render(fr); ctx.render(fr)
    st = ctx.stats()
    log = ctx.wave_log()
    print("records", log.shape, "primary_ms", st["primary_ms"], "paths_ms", st["paths_ms"])
    n = log.shape[0]
    live = log[log[:, 3] > 0]
    t_first = int(live[:, 0].min())
    passes = [log] if n <= 4096 else [log[: n // 2], log[n // 2:]]
    for k, lg in enumerate(passes):
        lg = lg[lg[:, 3] > 0]
        start = (lg[:, 0].astype(np.int64) - t_first) / 100.0   # microseconds since the first wave of the frame
        end = (lg[:, 1].astype(np.int64) - t_first) / 100.0
        life = lg[:, 2].astype(np.int64) * 16
        rounds = lg[:, 3].astype(np.int64); lanes = lg[:, 4].astype(np.int64)
        chunks = lg[:, 6] >> 8; xcc = lg[:, 6] & 15
        pct = lambda a: [round(float(np.percentile(a, q)), 1) for q in (0, 10, 50, 90, 99, 100)]
        print(f"pass {k}: waves {lg.shape[0]}  start us {pct(start)}  end us {pct(end)}")
        print(f"   rounds {pct(rounds)} cycles/round {pct(life / rounds)} lanes/round {pct(lanes / rounds)} chunks {pct(chunks)}")
        late = end > np.percentile(end, 95)
        for name, m in (("late 5%", late), ("rest", ~late)):
            print(f"   {name}: rounds {rounds[m].mean():.0f} cyc/round {(life[m] / rounds[m]).mean():.0f} lanes/round {(lanes[m] / rounds[m]).mean():.1f}"
                  f" chunks {chunks[m].mean():.1f} first {lg[m, 5].mean():.0f} start {start[m].mean():.1f}")
        for x in np.unique(xcc):
            m = xcc == x
            print(f"   xcc {x}: waves {m.sum()} end mean {end[m].mean():.1f} max {end[m].max():.1f} rounds {rounds[m].mean():.0f} cyc/round {(life[m] / rounds[m]).mean():.0f}")
    ctx.close()


if len(sys.argv) == 1:
    main()


def hw_report():
    """Which hardware units the slow waves ran on."""
    sc = scenes.boxes(1920, 1080, 64, 10)
    ctx = abi.Context()
    ctx.upload_scene(sc)
    fr = sc.frame()
    ctx.set_stats_detail(True)
    ctx.render(fr); ctx.render(fr)
    lg = ctx.wave_log()
    lg = lg[lg[:, 3] > 0]
    t_first = int(lg[:, 0].min())
    end = (lg[:, 1].astype(np.int64) - t_first) / 100.0
    life = lg[:, 2].astype(np.int64) * 16
    rounds = lg[:, 3].astype(np.int64)
    cpr = life / rounds
    xcc = lg[:, 6] & 15
    hw = lg[:, 7]
    cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3
    cu_key = xcc * 1000 + se * 100 + sh * 16 + cu
    keys = np.unique(cu_key)
    print("distinct CUs", len(keys))
    per_cu = np.array([[k, cpr[cu_key == k].mean(), end[cu_key == k].max(), (cu_key == k).sum(), rounds[cu_key == k].sum()] for k in keys])
    o = np.argsort(per_cu[:, 2])
    print("per-CU last end (us): percentiles", [round(float(np.percentile(per_cu[:, 2], q)), 1) for q in (0, 10, 50, 90, 99, 100)])
    print("per-CU mean cycles/round: percentiles", [round(float(np.percentile(per_cu[:, 1], q))) for q in (0, 10, 50, 90, 99, 100)])
    print("per-CU total rounds: percentiles", [round(float(np.percentile(per_cu[:, 4], q))) for q in (0, 10, 50, 90, 99, 100)])
    print("slowest CUs: key, cyc/round, last end, waves, rounds")
    for i in o[-8:]:
        print("   ", per_cu[i].round(1).tolist())
    wc = lg[:, 8].astype(np.int64) * 16; wr = lg[:, 9].astype(np.int64)
    o2 = np.argsort(wc)[-12:]
    print("longest items: cycles (M), us at 2.4 GHz, rounds, cycles/round, queue index, wave end")
    for i in o2:
        print(f"    {wc[i]/1e6:.2f}M {wc[i]/2400:.0f}us rounds {wr[i]} cyc/round {wc[i]/max(1,wr[i]):.0f} item {lg[i,10]} end {end[i]:.0f}")
    late = np.argsort(end)[-8:]
    for i in late:
        print(f"   late wave: end {end[i]:.0f} rounds {rounds[i]} cyc/round {cpr[i]:.0f} chunks {lg[i,6]>>8} xcc {xcc[i]} se {se[i]} sh {sh[i]} cu {cu[i]} simd {simd[i]}")
    ctx.close()


if len(sys.argv) > 1 and sys.argv[1] == "hw":
    hw_report()

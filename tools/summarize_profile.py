#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile.sh into profiles/<tag>_summary.json (+ kernel stats CSV copy).

    python tools/summarize_profile.py <tag> [--kernel k_paths] [--workload boxes]

Per-launch averages over the dispatches of the chosen kernel:
  duration from --kernel-trace; SQ counters; HBM-side traffic from FETCH_SIZE / WRITE_SIZE (reported by rocprofv3 in KiB).
  What FETCH_SIZE counts depends on the access pattern, calibrated on this part with tools/fetch_calib.sh
  (profiles/r03_fetch_calib.txt): it reports half the bytes of a wide coalesced stream (16 B per lane; 2.000 measured) and
  the bytes themselves of scattered 64-byte record gathers (0.98 measured). The factor applied is --fetch-factor: 2 for
  the workloads whose misses are streams (path list, parked pixel state, images), 1 for the 83,334-object scene, whose
  misses are node and object-record gathers. Both raw and corrected figures are recorded.
Also updates profiles/traffic.json, which bench.py reads for roofline.traffic.
"""
import argparse
import csv
import glob
import json
import re
import shutil
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def is_stats_variant(name):
    """k_primary<MODE, true, ...> / k_paths<MODE, true, ...>: the instrumented builds used for the work counters."""
    return re.search(r"<\d+, true[,>]", name.split("(")[0]) is not None


def read_counters(d, kernel, last):
    """Average per launch over the last `last` dispatches of the kernel (the bench's timed region: the frames before it
    are the first ones of a slot, which the library schedules without sample sweeps until it has seen the caller's pattern)."""
    files = glob.glob(str(d / "**" / "*counter_collection.csv"), recursive=True)
    per = defaultdict(lambda: defaultdict(float))   # counter -> dispatch -> value
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if kernel in row["Kernel_Name"] and not is_stats_variant(row["Kernel_Name"]):
                    per[row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
    out = {}
    for c, v in per.items():
        ids = sorted(v)[-last:]
        out[c] = (sum(v[i] for i in ids) / len(ids), len(ids))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--kernel", default="k_paths")
    ap.add_argument("--workload", default="boxes")
    ap.add_argument("--fif-steps", type=int, default=20, help="frames of the timed region in the frames-in-flight trace")
    ap.add_argument("--alone-steps", type=int, default=5, help="frames of the timed region in the one-frame-at-a-time passes")
    ap.add_argument("--fetch-factor", type=float, default=None, help="bytes per byte FETCH_SIZE reports (default: 1 for instanced_boxes, else 2)")
    args = ap.parse_args()
    src = ROOT / "gpurun_out" / f"prof_{args.tag}"
    dst = ROOT / "profiles"
    dst.mkdir(exist_ok=True)
    out = {"tag": args.tag, "kernel": args.kernel, "workload": args.workload}

    stats = glob.glob(str(src / "trace" / "**" / "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], dst / f"{args.tag}_kernel_stats.csv")
        with open(stats[0]) as fh:
            rows = list(csv.DictReader(fh))
        out["kernel_stats"] = [{"name": r["Name"][:90], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                "pct": float(r["Percentage"])} for r in rows[:8]]
        for r in rows:
            if args.kernel in r["Name"] and not is_stats_variant(r["Name"]):
                out["avg_kernel_ms_all_calls"] = float(r["AverageNs"]) / 1e6
                out["calls"] = int(r["Calls"])
                break
    trace = glob.glob(str(src / "trace" / "**" / "*kernel_trace.csv"), recursive=True)
    if trace:
        with open(trace[0]) as fh:
            mine = [r for r in csv.DictReader(fh) if args.kernel in r["Kernel_Name"] and not is_stats_variant(r["Kernel_Name"])]
        if mine:
            r = mine[0]
            out["launch"] = {k: r[k] for k in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size",
                                               "Workgroup_Size_X", "Grid_Size_X")}
            # the kernel's duration: mean over the timed region, i.e. the last `alone_steps` launches of the trace
            mine.sort(key=lambda r: int(r["Start_Timestamp"]))
            last = mine[-args.alone_steps:]
            out["avg_kernel_ms"] = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last) / len(last) / 1e6
            out["avg_kernel_ms_launches"] = len(last)
    # Frames in flight as the bench runs them: the last `fif_steps` frames of that trace are its timed region. Per-step wall
    # time from the device timestamps, and how much the frames overlap (kernel time summed over wall time; share of the
    # wall time in which kernels of two or more frames run at once). The rows used are copied to profiles/ as well.
    fif = glob.glob(str(src / "trace_fif" / "**" / "*kernel_trace.csv"), recursive=True)
    if fif:
        with open(fif[0]) as fh:
            rows = [r for r in csv.DictReader(fh) if not is_stats_variant(r["Kernel_Name"])]
        frame_rows = [r for r in rows if "k_primary" in r["Kernel_Name"] or "k_paths" in r["Kernel_Name"] or "k_chunk" in r["Kernel_Name"]]
        paths = sorted((r for r in frame_rows if "k_paths" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
        n = min(args.fif_steps, len(paths))
        if n:
            t_lo = int(paths[-n]["Start_Timestamp"])
            prim = sorted((r for r in frame_rows if "k_primary" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
            # the k_primary of the first counted frame starts before its k_paths
            firsts = [int(r["Start_Timestamp"]) for r in prim if int(r["Start_Timestamp"]) <= t_lo]
            t0 = max(firsts) if firsts else t_lo
            region = sorted((r for r in frame_rows if int(r["Start_Timestamp"]) >= t0), key=lambda r: int(r["Start_Timestamp"]))
            t1 = max(int(r["End_Timestamp"]) for r in region)
            ev = []
            for r in region:
                ev.append((int(r["Start_Timestamp"]), 1))
                ev.append((int(r["End_Timestamp"]), -1))
            ev.sort()
            busy = multi = 0
            depth, last_t = 0, t0
            for t, d in ev:
                if depth >= 1:
                    busy += t - last_t
                if depth >= 2:
                    multi += t - last_t
                depth += d
                last_t = t
            ksum = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in region)
            wall = t1 - t0
            out["frames_in_flight_trace"] = {
                "frames": n, "wall_ns": wall, "ms_per_step": wall / n / 1e6, "kernel_time_over_wall": ksum / wall,
                "share_of_wall_with_two_or_more_kernels": multi / wall, "share_of_wall_with_no_kernel": 1.0 - busy / wall,
                "mean_k_paths_ms": sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in paths[-n:]) / n / 1e6,
                "rows": f"profiles/{args.tag}_trace_fif.csv"}
            with open(dst / f"{args.tag}_trace_fif.csv", "w") as fh:
                fh.write("kernel,queue,start_ns,end_ns\n")
                for r in region:
                    fh.write(f"{r['Kernel_Name'].split('(')[0][:40].replace(',', ';')},{r.get('Queue_Id', '')},{int(r['Start_Timestamp']) - t0},{int(r['End_Timestamp']) - t0}\n")
        logs = glob.glob(str(src / "trace_fif.log"))
        if logs:
            for line in open(logs[0]):
                if line.startswith("{") and "only_timed" in line:
                    out["frames_in_flight_trace_bench_line"] = json.loads(line)
    counters = {}
    for sub in ("pmc_sq", "pmc_sq2", "pmc_sq3", "pmc_fetch", "pmc_write"):
        for c, (avg, n) in read_counters(src / sub, args.kernel, args.alone_steps).items():
            counters[c] = avg
    out["counters_per_launch"] = counters
    if "SQ_WAVE_CYCLES" in counters and counters.get("SQ_WAVE_CYCLES"):
        wc = counters["SQ_WAVE_CYCLES"]
        out["derived"] = {
            "valu_active_share_of_wave_cycles": counters.get("SQ_ACTIVE_INST_VALU", 0) / wc,
            "any_inst_active_share": counters.get("SQ_ACTIVE_INST_ANY", 0) / wc,
            "wait_inst_any_share": counters.get("SQ_WAIT_INST_ANY", 0) / wc,
            "valu_insts_per_wave": counters.get("SQ_INSTS_VALU", 0) / max(counters.get("SQ_WAVES", 1), 1),
            "salu_insts_per_wave": counters.get("SQ_INSTS_SALU", 0) / max(counters.get("SQ_WAVES", 1), 1),
        }
        if "SQ_THREAD_CYCLES_VALU" in counters and counters.get("SQ_ACTIVE_INST_VALU"):
            # lanes active per VALU instruction (64 = no divergence)
            out["derived"]["valu_lane_utilisation"] = counters["SQ_THREAD_CYCLES_VALU"] / (counters["SQ_ACTIVE_INST_VALU"] * 64.0) / 4.0 * 4.0
    if "FETCH_SIZE" in counters or "WRITE_SIZE" in counters:
        fetch_kib = counters.get("FETCH_SIZE", 0.0)
        write_kib = counters.get("WRITE_SIZE", 0.0)
        factor = args.fetch_factor if args.fetch_factor is not None else (1.0 if args.workload == "instanced_boxes" else 2.0)
        hbm = {"fetch_bytes_raw": fetch_kib * 1024, "fetch_bytes_corrected_x2": fetch_kib * 2048, "write_bytes": write_kib * 1024,
               "fetch_factor_applied": factor, "fetch_factor_source": "profiles/r03_fetch_calib.txt"}
        hbm["hbm_bytes_per_launch"] = hbm["fetch_bytes_raw"] * factor + hbm["write_bytes"]
        if "avg_kernel_ms" in out:
            hbm["GBs"] = hbm["hbm_bytes_per_launch"] / (out["avg_kernel_ms"] * 1e-3) / 1e9
        out["hbm"] = hbm
        tf = dst / "traffic.json"
        cur = json.loads(tf.read_text()) if tf.exists() else {}
        cur[args.workload] = {"hbm_bytes_per_launch": hbm["hbm_bytes_per_launch"], "tag": args.tag}
        if "avg_kernel_ms" in out:   # the kernel's average duration in the one-frame-at-a-time trace (bench.py: roofline.kernel_ms_rocprof)
            cur[args.workload]["k_paths_ms_rocprof"] = round(out["avg_kernel_ms"], 4)
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAVES",
                  "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_INT32",
                  "SQ_INST_CYCLES_VALU", "SQ_WAIT_INST_LDS", "SQ_BUSY_CYCLES"):
            if c in counters:
                cur[args.workload][c] = counters[c]
        tf.write_text(json.dumps(cur, indent=1) + "\n")
    (dst / f"{args.tag}_summary.json").write_text(json.dumps(out, indent=1) + "\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

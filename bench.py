#!/usr/bin/env python3
"""Benchmark of the hot path: one step = one frame of the path tracer over one synthetic scene.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload boxes] [--no-cpu-baseline]

N = 1 renders the headline configuration: the examples/boxes.rs scene at 1920x1080, 64 spp, 10 bounces,
scene and image resident in HBM. For N > 1 (launched by torch.distributed.run, one rank per GPU) every
rank renders the same frame as an independent virtual frame (its own `time`, hence its own per-pixel
seeds, exactly how the reference de-correlates successive frames) and the radiance buffers are summed
with one RCCL all-reduce per step: N x 64 spp per step, per-GPU work fixed (weak scaling).
`--multi-gpu tiles` instead splits the 8x8 tiles of ONE frame over the ranks (all-reduce = gather, image
bit-identical to the one-GPU image, strong scaling).

Frames are kept in flight (--frames-in-flight, default 3; 5 and 4 for the two workloads whose frames are chains of few
pixels, cornell and scene_glb_textured): that many contexts take the frames in turn on their own streams, so the start of a frame overlaps the tail of the previous one; results are those of one context.

Every step is a new frame the way the reference makes one (src/raytrace.wgsl:210, src/extract.rs:297-346): `time`
advances, so every pixel draws new random numbers, and the object array is handed over again (unchanged, as the example
scenes are static). The timed region of K steps is repeated (--repeats, fewer for workloads of seconds per repeat);
`value` and `ms_per_step` are the median repeat, the fastest one is reported beside it. Three more timings of the same K
steps are reported: one frame at a time on one context (what a caller of the synchronous pt_render sees, without the
copy to the host), with a camera that swings about the view target by up to 0.002 rad per frame (the schedule hints then
come from a slightly different view), and with the hints switched off (what the first frame after a cut gets).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

# Frames in flight run on separate HIP streams; they only overlap on the GPU if each stream gets a hardware queue of
# its own. HIP's default of 4 queues per process is shared with the contexts' upload streams, torch's streams and
# RCCL's; with 8 a kernel trace still showed the fourth of four frame streams on the third one's queue, with 16 up to
# six streams ran side by side (read by the runtime when it initialises, i.e. before torch is imported).
if int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0) < 16:
    os.environ["GPU_MAX_HW_QUEUES"] = "16"

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from path_tracing_amd import abi, scenes  # noqa: E402

# Algorithmic work per unit (flop, bytes): ray-object transform, BVH node box test, ray-triangle test,
# hit-record construction, scatter evaluation. Counted from the reference's expressions
# (src/query.wgsl:73-74, :127-147, :95-109, :113-120; src/raytrace.wgsl:157-203).
WORK = {"xforms": (33 + 25, 96 + 24), "box_tests": (25, 32), "tri_tests": (53, 36), "hits": (73, 60), "shades": (200, 64)}
PEAK_FP32_VALU_TFLOPS = 157.3   # MI355X vector FP32 (256 CU x 128 lanes x 2 flop x 2.4 GHz)
PEAK_HBM_GBS = 8000.0
# What one wave64 VALU instruction costs a SIMD, measured with tools/micro/valu_mix.hip on this part at 2 to 8 waves per
# SIMD (profiles/r02_valu_mix.txt): v_fma_f32 / v_fmac_f32 3.6-3.9 cycles; v_mul_f32, v_add_f32, v_mov_b32, v_add_u32, v_xor_b32
# 2.3-2.4; v_min/max/max3/med3, compares, v_cndmask, conversions, v_mul_lo_u32, the v_div_* helpers, v_readlane 4.1-4.3;
# v_rcp/v_sqrt 8.2; v_pk_fma_f32 4.4 (two fused multiply-adds: the 157.3 TFLOP/s peak is the rate of back-to-back packed
# FMAs on full waves; a stream of single FMAs tops out at 83). One wave alone on a SIMD pays 4.8-5.2 for the 2.4- and
# 4-cycle instructions alike. The kernel's issue time is priced per class from the SQ_INSTS_VALU_* counters of the PMC
# passes; INT32 is a mix of 2.4- and 4.3-cycle instructions, "other" (min/max, compares, selects, conversions, moves,
# bit operations) mostly 4.2 with the moves at 2.4.
VALU_COST = {"SQ_INSTS_VALU_FMA_F32": 3.8, "SQ_INSTS_VALU_MUL_F32": 2.35, "SQ_INSTS_VALU_ADD_F32": 2.4, "SQ_INSTS_VALU_TRANS_F32": 8.2,
             "SQ_INSTS_VALU_INT32": 3.3}
VALU_COST_OTHER = 3.9
N_SIMD = 1024
SHADER_GHZ = 2.4

WORKLOADS = {
    "boxes": ("examples/boxes.rs scene, 1920x1080, 64 spp, 10 bounces", lambda: scenes.boxes(1920, 1080, 64, 10)),
    "cornell": ("examples/cornellbox.rs scene, 512x512, 16 spp, 10 bounces", lambda: scenes.cornell_box(512, 512, 16, 10)),
    "scene_glb_textured": ("assets/scene.glb + example.png texture, 1920x1080, 256 spp", lambda: scenes.scene_glb_textured(1920, 1080, 256, 10)),
    "instanced_boxes": ("83,334 instanced cubes (1,000,008 triangles), 1920x1080, 256 spp", lambda: scenes.instanced_boxes(83334, 1920, 1080, 256, 10)),
    "cornell_4k": ("Cornell box 3840x2160, 4096 spp", lambda: scenes.cornell_4k(4096)),
}


# Frames kept in flight by default (ms per frame at 3 / 4 / 5 / 6 with 16 hardware queues: boxes 1.52 / 1.60 / 1.63 / 1.59,
# cornell 0.466 / 0.433 / 0.413 / 0.412, scene_glb_textured 11.0 / 10.7 / 10.9 / 11.1)
FRAMES_IN_FLIGHT = {"cornell": 5, "scene_glb_textured": 4}


def virtual_frame_time(rank: int) -> float:
    """time + 1 is the seed multiplier (src/raytrace.wgsl:210). Rank 0 uses time = 0 (seed = pixel coordinates);
    further ranks use k + 0.5 so that (x + 0.5) * (time + 1) never sits on an integer boundary."""
    return 0.0 if rank == 0 else float(rank) + 0.5


def cpu_baseline(scene, spp: int, min_seconds: float = 10.0):
    """The CPU restatement (oracle/) timed on this host: same scene and resolution, frames of `spp` samples
    (successive virtual frames, like successive frames of the reference) until `min_seconds` of work are done."""
    from tests.oracle_lib import load_oracle, oracle_render
    lib = load_oracle()
    threads = int(lib.pto_num_threads())
    threads = max(1, min(threads, 16, os.cpu_count() or 1))
    rays, frames = 0, 0
    t0 = time.perf_counter()
    while True:
        frame = scene.frame(samples=spp, time=virtual_frame_time(frames))
        r = oracle_render(scene, frame, threads=threads)
        rays += int(r["ray_count"].astype(np.uint64).sum())
        frames += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds or frames >= 64:
            break
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{scene.name} {frame.width}x{frame.height}, {frames} frames of {spp} spp, all bounces "
                      f"({rays} rays in {dt:.2f} s; the restatement re-traces the primary ray for every sample as the reference does)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="boxes", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=None, help="override samples per pixel (invalidates the headline config)")
    ap.add_argument("--cpu-spp", type=int, default=64)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, default=None, help="contexts that take the frames in turn (1: strictly one frame at a time); "
                    "default: 3, for the workloads whose frames are chains of few pixels what measured best (cornell 5, scene_glb_textured 4)")
    ap.add_argument("--repeats", type=int, default=10, help="times the K-step timed region is run (median and minimum are reported); "
                    "cut down so that the repeats of one timing take about 8 s at most")
    ap.add_argument("--only-timed", action="store_true", help="stop after the timed region and print a short line (for profiler passes: "
                    "the last steps x repeats frames of the trace are then the timed region)")
    ap.add_argument("--fixed-time", action="store_true", help="render the identical frame every step (time = this rank's constant) instead of advancing time")
    ap.add_argument("--multi-gpu", default="virtual_frames", choices=["virtual_frames", "tiles"],
                    help="N > 1: every GPU renders its own virtual frame (N x spp, weak scaling) or a share of the tiles of one frame (strong scaling)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo: rehearsal of the N>1 path with every rank on GPU 0")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.backend == "gloo":
        local_rank = 0   # rehearsal on a one-GPU box: all ranks share device 0, the reduction goes through the host
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend="gloo")

    desc, make = WORKLOADS[args.workload]
    scene = make()
    if args.spp:
        scene.samples = args.spp
    tiles = args.multi_gpu == "tiles" and world > 1
    if tiles:
        # exact split: 8x8 tiles dealt round-robin to the ranks, all-reduce(sum) of the disjoint images = gather;
        # the result is bit-identical to the one-GPU image (total work fixed: strong scaling)
        frame = scene.frame(time=0.0, part_index=rank, part_count=world)
    else:
        frame = scene.frame(time=virtual_frame_time(rank))
    W, H = frame.width, frame.height

    # Frames in flight: F contexts (the scene arrays are small), each with its own stream and image buffer, take
    # the frames in turn, so the tail of one frame's path kernel (waves running out of work) is filled by the
    # start of the next frame. For N > 1 the all-reduce of a frame runs on one more stream while later frames are
    # traced. Every frame and every reduction is complete before the closing synchronize of the timed region.
    F = max(1, args.frames_in_flight if args.frames_in_flight is not None else FRAMES_IN_FLIGHT.get(args.workload, 3))
    ctxs = [abi.Context(device=local_rank) for _ in range(F)]
    for c in ctxs:
        c.upload_scene(scene)
    ctx = ctxs[0]
    bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device=device) for _ in range(F)]
    out = bufs[0]
    streams = [torch.cuda.Stream(device) for _ in range(F)]
    stream = streams[0]
    comm_stream = torch.cuda.Stream(device) if world > 1 else None
    rendered = [torch.cuda.Event() for _ in bufs]
    reduced = [torch.cuda.Event() for _ in bufs]
    step_no = [0]

    frames = [scene.frame(time=virtual_frame_time(rank), part_index=rank, part_count=world) if tiles else scene.frame(time=virtual_frame_time(rank))
              for _ in range(F)]

    def frame_time(i):
        """`time` of this rank's i-th step: every rank and step its own virtual frame (tile split: all ranks render the same one)."""
        if args.fixed_time:
            return 0.0 if tiles else virtual_frame_time(rank)
        return virtual_frame_time(i if tiles else i * world + rank)

    # a camera that flies (reference examples/common.rs:25-95): it swings about the view target, +-0.05 rad around the
    # scene's own view (so that the work per frame stays what the other timings measure), at most 0.002 rad per frame
    orbit = [False]
    eye0 = np.asarray(scene.camera.eye, dtype=np.float64)
    tgt0 = np.asarray(scene.camera.target, dtype=np.float64)
    wfv0 = [float(frames[0].world_from_view[j]) for j in range(16)]

    orbit_views = []   # made outside the timed regions: one period of 160 frames, then the resting view

    def make_orbit_views():
        d = eye0 - tgt0
        for k in range(160):
            a = 0.05 * np.sin(2.0 * np.pi * k / 160.0)   # at most 0.002 rad from one frame to the next
            eye = tgt0 + np.array([np.cos(a) * d[0] + np.sin(a) * d[2], d[1], -np.sin(a) * d[0] + np.cos(a) * d[2]])
            wfv, _ = scenes.Camera(tuple(eye), tuple(tgt0), scene.camera.fov_y).matrices(W, H)
            orbit_views.append((abi.C.c_float * 16)(*[float(x) for x in wfv]))
        orbit_views.append((abi.C.c_float * 16)(*wfv0))

    def set_view(fr, i):
        if orbit[0]:
            fr.world_from_view = orbit_views[i % 160]
        elif orbit_views:
            fr.world_from_view = orbit_views[160]

    def step(cs=None, one_at_a_time=False):
        cs = cs or ctxs
        n = len(cs)
        i = step_no[0]
        k = i % n
        step_no[0] += 1
        buf = bufs[k]
        fr = frames[k]
        fr.time = frame_time(i)
        set_view(fr, i)
        cs[k].upload_objects(scene.objects)      # extract_visible hands the objects over every frame; unchanged here
        if world == 1:
            cs[k].render_device(fr, buf.data_ptr(), streams[k].cuda_stream)
            if one_at_a_time:
                streams[k].synchronize()
            return
        streams[k].wait_event(reduced[k])        # this buffer's previous reduction (F steps ago) has finished
        cs[k].render_device(fr, buf.data_ptr(), streams[k].cuda_stream)
        rendered[k].record(streams[k])
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(rendered[k])
            if args.backend == "nccl":
                dist.all_reduce(buf, op=dist.ReduceOp.SUM)
            else:
                host = buf.cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
                buf.copy_(host)
            if not tiles:
                buf.mul_(1.0 / world)
            reduced[k].record(comm_stream)
        if one_at_a_time:
            comm_stream.synchronize()

    # one untimed frame per context: allocations, and the ray counts the next frame's schedule is derived from
    for k in range(F):
        ctxs[k].render_device(frame, bufs[k].data_ptr(), streams[k].cuda_stream)
    torch.cuda.synchronize(device)
    if world > 1:
        # one untimed reduction of a full-size image: the collective library sets up its channels and buffers at the first
        # call, which must not fall into the timed region when the caller asks for no warm-up steps
        with torch.cuda.stream(comm_stream):
            if args.backend == "nccl":
                dist.all_reduce(bufs[0], op=dist.ReduceOp.SUM)
            else:
                host = bufs[0].cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
        comm_stream.synchronize()
        dist.barrier()

    # detailed work counters from one untimed pass (slower kernel variant)
    ctx.set_stats_detail(True)
    ctx.render_device(frame, out.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize(device)
    detail = ctx.stats()
    ctx.set_stats_detail(False)

    def timed(cs=None, one_at_a_time=False, repeats=None):
        """K steps between barrier + synchronize on both sides, `repeats` times; seconds per repeat (max over ranks)."""
        for _ in range(args.warmup):
            step(cs, one_at_a_time)
        out_s = []
        reps = repeats or args.repeats
        r = 0
        while r < reps:
            torch.cuda.synchronize(device)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step(cs, one_at_a_time)
            torch.cuda.synchronize(device)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(device)
            dt_r = time.perf_counter() - t0
            if world > 1:
                tm = torch.tensor([dt_r], dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
                dt_r = float(tm[0].item())
            out_s.append(dt_r)
            r += 1
            if r == 1 and not repeats:   # repeats of seconds each: fewer of them (every rank sees the same max-reduced time)
                reps = max(1, min(reps, int(8.0 / max(dt_r, 1e-9))))
        return out_s

    times = timed()
    dt = float(np.median(times))
    dt_min = float(np.min(times))
    mean_img = float(bufs[(step_no[0] - 1) % F][..., :3].mean().item())   # the last (reduced) image
    # rays of a step: the mean over the last frame of every context (the count moves by a fraction of a percent with `time`)
    last = [c.stats() for c in ctxs]
    if args.only_timed:
        if rank == 0:
            print(json.dumps({"only_timed": True, "workload": args.workload, "frames_in_flight": F, "steps": args.steps, "repeats": len(times),
                              "ms_per_step": round(dt / args.steps * 1e3, 4), "ms_per_step_min": round(dt_min / args.steps * 1e3, 4),
                              "rays_per_step": int(round(float(np.mean([x["rays"] for x in last]))))}), flush=True)
        for c in ctxs:
            c.close()
        if world > 1:
            dist.destroy_process_group()
        return
    n_rep_other = max(1, min(len(times), 5))
    # the same K steps, one frame at a time on one context: each frame complete before the next is enqueued
    times_one = timed([ctxs[0]], one_at_a_time=True, repeats=n_rep_other)
    # ... with a camera that moves a little every frame (the schedule hints then come from a slightly different view)
    make_orbit_views()
    orbit[0] = True
    times_mv = timed(repeats=n_rep_other)
    orbit[0] = False
    # ... and with the schedule hints off (what a cut to another view gets: the previous frame's ray counts say nothing)
    os.environ["PT_AMD_NO_COST_HINTS"] = "1"
    ctxs_nh = [abi.Context(device=local_rank) for _ in range(F)]
    del os.environ["PT_AMD_NO_COST_HINTS"]
    for c in ctxs_nh:
        c.upload_scene(scene)
    times_nh = timed(ctxs_nh, repeats=n_rep_other)
    for c in ctxs_nh:
        c.close()

    # kernel durations (HIP events recorded by the library on the launch stream), one step at a time
    kernel_ms, paths_ms, primary_ms = [], [], []
    for _ in range(args.steps):
        ctx.render_device(frame, out.data_ptr(), stream.cuda_stream)
        torch.cuda.synchronize(device)
        st = ctx.stats()
        kernel_ms.append(st["trace_ms"])
        paths_ms.append(st["paths_ms"])
        primary_ms.append(st["primary_ms"])
    rays_local = int(round(float(np.mean([x["rays"] for x in last]))))
    st = {"rays_reference": int(round(float(np.mean([x["rays_reference"] for x in last])))), "paths": last[0]["paths"]}
    t_max = dt
    rays_total = rays_local
    if world > 1:
        buf = torch.tensor([dt, float(rays_local), float(st["rays_reference"]), float(st["paths"])], dtype=torch.float64,
                           device=device if args.backend == "nccl" else "cpu")
        tmax = buf.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        t_max = float(tmax[0].item())
        rays_total = int(buf[1].item())
        rays_ref_total = int(buf[2].item())
        paths_total = int(buf[3].item())
    else:
        rays_ref_total = int(st["rays_reference"])
        paths_total = int(st["paths"])

    if rank == 0:
        ms_per_step = t_max / args.steps * 1e3
        value = rays_total / (t_max / args.steps) / 1e6
        k_ms = float(np.mean(paths_ms))   # the dominant kernel: k_paths
        # work of k_paths alone: the frame's counters minus what k_primary did
        work = {k: int(detail[k]) - int(detail.get("primary_" + k, 0)) for k in WORK}
        flops = sum(work[k] * WORK[k][0] for k in WORK)
        abytes = sum(work[k] * WORK[k][1] for k in WORK) + W * H * 16
        # what must cross HBM at least once per frame: the scene arrays in, the image and the four aux planes out
        scene_bytes = sum(int(a.nbytes) for a in (scene.objects, scene.meshes, scene.indices, scene.vertices, scene.materials,
                                                  scene.textures, scene.texture_data))
        compulsory = scene_bytes + W * H * 32
        achieved_tflops = flops / (k_ms * 1e-3) / 1e12
        traffic = None
        issue = None
        tf = ROOT / "profiles" / "traffic.json"
        if tf.exists():
            try:
                prof = json.loads(tf.read_text()).get(args.workload, {})
                traffic = prof.get("hbm_bytes_per_launch")
                if prof.get("SQ_INSTS_VALU"):
                    # instruction-issue view of the same kernel: VALU wave-instructions per launch by class (rocprofv3 PMC passes
                    # of tools/profile.sh) x the measured issue cost of the class, against the kernel's duration on all SIMDs
                    n_inst = float(prof["SQ_INSTS_VALU"])
                    classed = {k: float(prof[k]) for k in VALU_COST if prof.get(k)}
                    cycles = sum(classed[k] * VALU_COST[k] for k in classed) + (n_inst - sum(classed.values())) * VALU_COST_OTHER
                    issue_ms = cycles / N_SIMD / (SHADER_GHZ * 1e9) * 1e3
                    issue = {"valu_wave_insts_per_launch": int(n_inst), "by_class": {k.replace("SQ_INSTS_VALU_", ""): int(v) for k, v in classed.items()},
                             "cycles_per_inst": round(cycles / n_inst, 3),
                             "issue_ms_at_2.4GHz": round(issue_ms, 4), "frac_of_kernel_ms": round(issue_ms / k_ms, 4),
                             "frac_of_ms_per_step": round(issue_ms / ms_per_step, 4),
                             "lanes_active_per_inst": round(float(prof["SQ_THREAD_CYCLES_VALU"]) / (float(prof["SQ_ACTIVE_INST_VALU"]) * 64.0), 4)
                             if prof.get("SQ_THREAD_CYCLES_VALU") and prof.get("SQ_ACTIVE_INST_VALU") else None,
                             "profile": prof.get("tag")}
            except Exception:
                traffic = None
        line = {
            "metric": "Mrays/s (all bounces) at 1080p" if H == 1080 else "Mrays/s (all bounces)",
            "value": round(value, 1),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "repeats": len(times),
            "ms_per_step_min": round(dt_min / args.steps * 1e3, 4),
            "ms_per_step_all_repeats": [round(t / args.steps * 1e3, 4) for t in times],
            "ms_per_step_one_in_flight": round(float(np.median(times_one)) / args.steps * 1e3, 4),
            "ms_per_step_no_hints": round(float(np.median(times_nh)) / args.steps * 1e3, 4),
            "ms_per_step_moving_camera": round(float(np.median(times_mv)) / args.steps * 1e3, 4),
            "step_definition": "a new frame per step: time advances (new seeds for every pixel, src/raytrace.wgsl:210) and the unchanged object array is "
                               "handed over again (src/extract.rs:297-346)" if not args.fixed_time else "the identical frame every step",
            "higher_is_better": True,
            "scaling": "strong" if tiles else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": desc, "scene": scene.name, "width": W, "height": H, "spp_per_gpu": scene.samples,
                       "bounces": scene.bounces, "objects": int(scene.objects.size), "triangles": scene.n_triangles,
                       "multi_gpu": ("8x8 tiles round-robin over the GPUs + RCCL all-reduce(sum) as gather" if tiles else
                                     "one virtual frame per GPU + RCCL all-reduce(sum) of the f32 radiance buffer") if world > 1 else "single GPU",
                       "frames_in_flight": F},
            "rays_per_step": rays_total,
            "rays_definition": "closest-hit queries executed; the primary ray of a pixel is identical for all of its samples "
                               "(src/raytrace.wgsl:223) and is traced once, the reference would trace rays_reference_per_step",
            "rays_reference_per_step": rays_ref_total,
            "mrays_reference_equiv_per_s": round(rays_ref_total / (t_max / args.steps) / 1e6, 1),
            "samples_per_s": round(paths_total / (t_max / args.steps), 1),
            "image_mean": round(mean_img, 6),
            "kernels_ms": {"k_primary": round(float(np.mean(primary_ms)), 4), "k_paths": round(float(np.mean(paths_ms)), 4),
                           "frame": round(float(np.mean(kernel_ms)), 4)},
            "roofline": {"bound": "valu", "kernel": "k_paths", "achieved": round(achieved_tflops, 3), "peak": PEAK_FP32_VALU_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved_tflops / PEAK_FP32_VALU_TFLOPS, 5),
                         "traffic": traffic,
                         "kernel_ms": round(k_ms, 4), "algorithmic_flop_per_launch": int(flops),
                         "on_chip_bytes_per_launch": int(abytes),
                         "hbm": {"compulsory_bytes_per_launch": int(compulsory), "achieved_GBs": round(compulsory / (k_ms * 1e-3) / 1e9, 2),
                                 "peak_GBs": PEAK_HBM_GBS, "frac": round(compulsory / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 6)},
                         "valu_issue": issue,
                         "counts": work},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scene, min(args.cpu_spp, scene.samples))
        print(json.dumps(line), flush=True)

    for c in ctxs:
        c.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

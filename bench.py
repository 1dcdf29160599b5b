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

Frames are kept in flight (--frames-in-flight, default 3; 8 for the two workloads whose frames are chains of few
pixels, cornell and scene_glb_textured, 4 for instanced_boxes) on ONE context: pt_render_submit_device on that many slots in turn, pt_render_wait
on a slot before it is reused, so the start of a frame overlaps the tail of the previous one over one copy of the scene.
(N > 1 ranks, whose frames are followed by a collective on a stream of torch's, use one context per frame in flight with
pt_render_device on torch streams instead; `ms_per_step_contexts` times that arrangement on one GPU as well, in a child
process of its own.)

Every step is a new frame the way the reference makes one (src/raytrace.wgsl:210, src/extract.rs:297-346): `time`
advances, so every pixel draws new random numbers, and the object array is handed over again (unchanged, as the example
scenes are static). The timed region of K steps is repeated (--repeats, fewer for workloads of seconds per repeat);
`value` and `ms_per_step` are the median repeat, the fastest one is reported beside it. More timings of the same K
steps are reported: one frame at a time (one slot), with a camera that swings about the view target by up to 0.002 rad
per frame (the schedule hints then come from a slightly different view), with the hints switched off (what the first
frame after a cut gets), and what a plugin that replaces RayTraceNode::run gets through the boundary, image delivered to
page-locked host memory as binary16 each frame: synchronously (`ms_per_step_plugin_sync`: pt_render) and with frames in
flight (`ms_per_step_plugin_async`: pt_render_submit / pt_render_wait); and the headline arrangement in a process that
leaves HIP's hardware-queue count at its default (`ms_per_step_default_hw_queues`), also in one that never loads torch
(`ms_per_step_default_hw_queues_plain_hip`: --plain-hip).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

# Frames in flight run on separate HIP streams; they only overlap on the GPU if each stream gets a hardware queue of
# its own. HIP's default of 4 queues per process is shared with the contexts' upload streams, torch's streams and
# RCCL's; with 8 a kernel trace still showed the fourth of four frame streams on the third one's queue, with 16 up to
# six streams ran side by side (read by the runtime when it initialises, i.e. before torch is imported).
# --default-hw-queues leaves the variable alone (the child process behind `ms_per_step_default_hw_queues`).
# This process asks for 24: at times it holds two contexts with up to eight slots each (the headline's and the one
# without cost hints), and the arrangement with one context per frame in flight (--contexts, N > 1 ranks) has an upload
# stream and a torch stream per frame. A host application with one context needs its slots + 2.
_HW_QUEUES = 24
if "--default-hw-queues" not in sys.argv and int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0) < _HW_QUEUES:
    os.environ["GPU_MAX_HW_QUEUES"] = str(_HW_QUEUES)

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


# Frames kept in flight by default (ms per frame at 3 / 4 / 5 / 6 / 8 with 16 hardware queues: boxes 1.52 / 1.51 / 1.55 / 1.55 / 1.54,
# cornell 0.47 / 0.43 / 0.40 / 0.39 / 0.37, scene_glb_textured 11.0 / 10.5 / 9.7 / 9.4 / 8.8 -- from five on the library serves
# object-list scenes for throughput: no BLAS in LDS, no chunks in parts)
FRAMES_IN_FLIGHT = {"cornell": 8, "scene_glb_textured": 8, "instanced_boxes": 4}   # (instanced_boxes at 3 / 4 / 5: 354 / 342 / 339 ms)


def virtual_frame_time(rank: int) -> float:
    """time + 1 is the seed multiplier (src/raytrace.wgsl:210). Rank 0 uses time = 0 (seed = pixel coordinates);
    further ranks use k + 0.5 so that (x + 0.5) * (time + 1) never sits on an integer boundary."""
    return 0.0 if rank == 0 else float(rank) + 0.5


def cpu_baseline(scene, spp: int, min_seconds: float = 10.0):
    """The CPU restatement (oracle/) timed on this host: same scene and resolution, frames of `spp` samples
    (successive virtual frames, like successive frames of the reference) until `min_seconds` of work are done.
    The restatement tests every triangle of every object for every ray, as the reference does: scenes with more than
    10,000 instanced triangles are timed on a 64 x 64 crop around the image centre with at most 4 samples per pixel."""
    from tests.oracle_lib import load_oracle, oracle_render
    lib = load_oracle()
    threads = int(lib.pto_num_threads())
    threads = max(1, min(threads, 16, os.cpu_count() or 1))
    n_tris = int(sum(int(scene.meshes["tri_count"][int(m)]) for m in scene.objects["mesh"])) if scene.objects.size else 0
    rect = None
    if n_tris > 10000:
        cx, cy = scene.width // 2, scene.height // 2
        rect = (max(cx - 32, 0), max(cy - 32, 0), min(cx + 32, scene.width), min(cy + 32, scene.height))
        spp = min(spp, 4)
    rays, frames = 0, 0
    t0 = time.perf_counter()
    while True:
        frame = scene.frame(samples=spp, time=virtual_frame_time(frames))
        r = oracle_render(scene, frame, threads=threads, rect=rect) if rect else oracle_render(scene, frame, threads=threads)
        rays += int(r["ray_count"].astype(np.uint64).sum())
        frames += 1
        dt = time.perf_counter() - t0
        print(f"[bench] cpu baseline: frame {frames}, {dt:.1f} s", file=sys.stderr, flush=True)
        if dt >= min_seconds or frames >= 64:
            break
    where = f"{frame.width}x{frame.height}" if not rect else f"pixels [{rect[0]},{rect[2]}) x [{rect[1]},{rect[3]}) of {frame.width}x{frame.height}"
    return {"value": rays / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{scene.name} {where}, {frames} frames of {spp} spp, all bounces "
                      f"({rays} rays in {dt:.2f} s; the restatement re-traces the primary ray for every sample as the reference does)"}


def plain_hip_timed(args):
    """--plain-hip: the --only-timed measurement of the headline arrangement (F slots of one context, a new frame every
    step, images left in device memory) in a process that never loads torch, i.e. on the system's HIP runtime with no
    streams but the library's own -- what a host application written against the C ABI gets. Device images come from
    hipMalloc; the timed region is bracketed by hipDeviceSynchronize."""
    import ctypes as C
    desc, make = WORKLOADS[args.workload]
    scene = make()
    if args.spp:
        scene.samples = args.spp
    F = max(1, args.frames_in_flight if args.frames_in_flight is not None else FRAMES_IN_FLIGHT.get(args.workload, 3))
    if F > abi.PT_MAX_FRAME_SLOTS:
        raise SystemExit(f"--frames-in-flight: at most {abi.PT_MAX_FRAME_SLOTS}")
    abi.load_library()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    frames = [scene.frame(time=virtual_frame_time(0)) for _ in range(F)]
    nbytes = frames[0].width * frames[0].height * 16
    bufs = [C.c_void_p() for _ in range(F)]
    for b in bufs:
        if hip.hipMalloc(C.byref(b), nbytes) != 0:
            raise SystemExit("hipMalloc failed")
    times = []
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        busy = [False] * F
        step_no = [0]

        def step():
            i = step_no[0]
            s = i % F
            step_no[0] += 1
            if busy[s]:
                ctx.wait(s)
            frames[s].time = 0.0 if args.fixed_time else virtual_frame_time(i)
            ctx.upload_objects(scene.objects)
            ctx.submit_device(frames[s], s, bufs[s].value)
            busy[s] = True

        def drain():
            for s in range(F):
                if busy[s]:
                    ctx.wait(s)
                    busy[s] = False

        for _ in range(args.warmup):
            step()
        drain()
        reps = args.repeats
        r = 0
        while r < reps:
            hip.hipDeviceSynchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            drain()
            hip.hipDeviceSynchronize()
            times.append(time.perf_counter() - t0)
            r += 1
            print(f"[bench] plain HIP: repeat {r} of {reps}: {times[-1] / args.steps * 1e3:.4f} ms per step", file=sys.stderr, flush=True)
            if r == 1:
                reps = max(1, min(reps, int(8.0 / max(times[0], 1e-9))))
        rays = ctx.stats()["rays"]
    for b in bufs:
        hip.hipFree(b)
    print(json.dumps({"only_timed": True, "workload": args.workload, "frames_in_flight": F, "steps": args.steps, "repeats": len(times),
                      "arrangement": "slots of one context", "host": "plain HIP (torch never loaded)",
                      "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "default"),
                      "ms_per_step": round(float(np.median(times)) / args.steps * 1e3, 4), "ms_per_step_min": round(float(np.min(times)) / args.steps * 1e3, 4),
                      "rays_per_step": int(rays)}), flush=True)


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
    ap.add_argument("--default-hw-queues", action="store_true", help="do not raise GPU_MAX_HW_QUEUES (HIP's default is 4 hardware queues per process)")
    ap.add_argument("--contexts", action="store_true", help="one GPU: keep the frames in flight on one context each (pt_render_device on torch "
                    "streams) instead of on the slots of one context")
    ap.add_argument("--plain-hip", action="store_true", help="with --only-timed on one GPU: the same timing in a process that never loads torch "
                    "(device images from hipMalloc); what a host application on the system's HIP runtime gets")
    args = ap.parse_args()
    if args.plain_hip:
        if not args.only_timed or args.gpus != 1 or args.contexts or int(os.environ.get("WORLD_SIZE", "1")) != 1:
            raise SystemExit("--plain-hip goes with --only-timed on one GPU (slots of one context)")
        return plain_hip_timed(args)

    # The headline arrangement in a process that leaves HIP's hardware-queue count alone, measured first: a child process
    # of its own, run to its end before this process creates its GPU state (a process that merely holds contexts and
    # streams of its own on the device slows another one's overlapping frames down by a factor of up to two).
    child_extra = {}
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.gpus == 1 and not args.only_timed and not args.default_hw_queues and not args.contexts:
        def child(key, what, flags, env):
            cmd = [sys.executable, str(ROOT / "bench.py"), "--workload", args.workload, "--steps", str(args.steps), "--warmup", str(args.warmup),
                   "--repeats", str(max(1, min(args.repeats, 5))), "--only-timed"] + flags
            if args.frames_in_flight is not None:
                cmd += ["--frames-in-flight", str(args.frames_in_flight)]
            if args.spp:
                cmd += ["--spp", str(args.spp)]
            if args.fixed_time:
                cmd += ["--fixed-time"]
            print(f"[bench] child process {what} ...", file=sys.stderr, flush=True)
            try:
                r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True, timeout=900)   # (its progress lines go to this process's stderr)
                child_extra[key] = json.loads(r.stdout.strip().splitlines()[-1])["ms_per_step"]
            except Exception as e:   # (reported, not fatal: the headline does not depend on it)
                child_extra[key] = None
                child_extra[key + "_error"] = repr(e)[:200]
        child("ms_per_step_default_hw_queues", "with the default hardware-queue count", ["--default-hw-queues"],
              {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"})
        # ... and the same without torch in the process: torch brings its own HIP runtime and streams, which take hardware
        # queues too; a host application written against the C ABI has the four default queues to itself
        child("ms_per_step_default_hw_queues_plain_hip", "with the default hardware-queue count, torch never loaded", ["--default-hw-queues", "--plain-hip"],
              {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"})
        # The arrangement of earlier records, one context (and one copy of the scene) per frame in flight, each driven with
        # pt_render_device on a torch stream: in a process of its own as well, since its streams and the slots' streams of
        # the headline arrangement together would outnumber the hardware queues.
        child("ms_per_step_contexts", "with one context per frame in flight", ["--contexts"], dict(os.environ))

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

    # Frames in flight. One GPU: ONE context, F slots (pt_render_submit_device / pt_render_wait), one copy of the scene;
    # the tail of one frame's path kernel (waves running out of work) is filled by the start of the next frame.
    # N > 1 (or --contexts): F contexts, each with its own torch stream and image buffer, take the frames in turn, and the
    # all-reduce of a frame runs on one more stream while later frames are traced. Every frame and every reduction is
    # complete before the closing synchronize of the timed region.
    F = max(1, args.frames_in_flight if args.frames_in_flight is not None else FRAMES_IN_FLIGHT.get(args.workload, 3))
    if F > abi.PT_MAX_FRAME_SLOTS:
        raise SystemExit(f"--frames-in-flight: at most {abi.PT_MAX_FRAME_SLOTS}")
    use_slots = world == 1 and not args.contexts
    # (one context per frame in flight: a context sees only its own frame, so it is told that the pipeline is deep --
    # object-list scenes are then served for throughput, as the slots of one context find out for themselves)
    deep = not use_slots and F >= 5 and "PT_AMD_DEEP_PIPELINE" not in os.environ
    if deep:
        os.environ["PT_AMD_DEEP_PIPELINE"] = "0"
    ctxs = [abi.Context(device=local_rank) for _ in range(1 if use_slots else F)]
    if deep:
        del os.environ["PT_AMD_DEEP_PIPELINE"]
    for c in ctxs:
        c.upload_scene(scene)
    ctx = ctxs[0]
    bufs = [torch.zeros((H, W, 4), dtype=torch.float32, device=device) for _ in range(F)]
    out = bufs[0]
    # (torch streams only where frames are driven through them: a stream that is never used still takes one of the
    # process's hardware queues, and with HIP's default of four the slots' own streams then have to share)
    streams = [torch.cuda.Stream(device) for _ in range(F)] if not use_slots else []
    stream = streams[0] if streams else torch.cuda.current_stream(device)
    comm_stream = torch.cuda.Stream(device) if world > 1 else None
    rendered = [torch.cuda.Event() for _ in bufs]
    reduced = [torch.cuda.Event() for _ in bufs]
    comm_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in bufs]
    comm_ms = []
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

    class Slots:
        """K frames in flight on the slots of one context; `host`: K page-locked arrays the images are copied into
        (the plugin's path), else they stay in `bufs` on the device."""

        def __init__(self, c, k, host=None, f16=False):
            self.c, self.k, self.host = c, k, host
            self.busy = [False] * k
            self.flags = abi.PT_FRAME_OUTPUT_F16 if f16 else 0

        def step(self):
            i = step_no[0]
            s = i % self.k
            step_no[0] += 1
            if self.busy[s]:
                self.c.wait(s)
            fr = frames[s]
            fr.time = frame_time(i)
            fr.flags = self.flags
            set_view(fr, i)
            self.c.upload_objects(scene.objects)      # extract_visible hands the objects over every frame; unchanged here
            if self.host is not None:
                self.c.submit(fr, s, self.host[s])
            else:
                self.c.submit_device(fr, s, bufs[s].data_ptr())
            self.busy[s] = True

        def drain(self):
            for s in range(self.k):
                if self.busy[s]:
                    self.c.wait(s)
                    self.busy[s] = False
            for fr in frames:
                fr.flags = 0

    class Contexts:
        """One context per frame in flight, pt_render_device on a torch stream each; for N > 1 followed by the collective."""

        def __init__(self, cs):
            self.cs = cs

        def step(self):
            cs = self.cs
            n = len(cs)
            i = step_no[0]
            k = i % n
            step_no[0] += 1
            buf = bufs[k]
            fr = frames[k]
            fr.time = frame_time(i)
            set_view(fr, i)
            cs[k].upload_objects(scene.objects)
            if world == 1:
                cs[k].render_device(fr, buf.data_ptr(), streams[k].cuda_stream)
                return
            streams[k].wait_event(reduced[k])        # this buffer's previous reduction (F steps ago) has finished
            cs[k].render_device(fr, buf.data_ptr(), streams[k].cuda_stream)
            rendered[k].record(streams[k])
            with torch.cuda.stream(comm_stream):
                comm_stream.wait_event(rendered[k])
                comm_ev[k][0].record(comm_stream)
                if args.backend == "nccl":
                    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
                else:
                    host = buf.cpu()
                    dist.all_reduce(host, op=dist.ReduceOp.SUM)
                    buf.copy_(host)
                if not tiles:
                    buf.mul_(1.0 / world)
                comm_ev[k][1].record(comm_stream)
                reduced[k].record(comm_stream)

        def drain(self):
            torch.cuda.synchronize(device)
            if world > 1:   # device time of the last reductions (events on the collective's stream), for comm_ms_per_step
                for a, b in comm_ev[:len(self.cs)]:
                    try:
                        comm_ms.append(a.elapsed_time(b))
                    except Exception:
                        pass

    class PluginSync:
        """What a drop-in replacement of RayTraceNode::run (reference src/shader.rs:121-213) does: one synchronous pt_render per frame, binary16
        image into page-locked host memory."""

        def __init__(self, c, host):
            self.c, self.host = c, host

        def step(self):
            i = step_no[0]
            step_no[0] += 1
            fr = frames[0]
            fr.time = frame_time(i)
            fr.flags = abi.PT_FRAME_OUTPUT_F16
            set_view(fr, i)
            self.c.upload_objects(scene.objects)
            self.c.render(fr, out=self.host)

        def drain(self):
            frames[0].flags = 0

    runner = Slots(ctx, F) if use_slots else Contexts(ctxs)

    # one untimed frame per slot / context: allocations, and the ray counts the next frame's schedule is derived from
    for k in range(F):
        if use_slots:
            ctx.submit_device(frame, k, bufs[k].data_ptr())
        else:
            ctxs[k].render_device(frame, bufs[k].data_ptr(), streams[k].cuda_stream)
    if use_slots:
        for k in range(F):
            ctx.wait(k)
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

    def timed(run=None, repeats=None):
        """K steps between barrier + synchronize on both sides, `repeats` times; seconds per repeat (max over ranks)."""
        run = run or runner
        for _ in range(args.warmup):
            run.step()
        run.drain()
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
                run.step()
            run.drain()
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
            if rank == 0:   # (a line of progress per repeat: long workloads would otherwise be silent for minutes)
                print(f"[bench] {type(run).__name__}: repeat {r} of {reps}: {dt_r / args.steps * 1e3:.4f} ms per step", file=sys.stderr, flush=True)
            if r == 1 and not repeats:   # repeats of seconds each: fewer of them (every rank sees the same max-reduced time)
                reps = max(1, min(reps, int(8.0 / max(dt_r, 1e-9))))
        return out_s

    times = timed()
    dt = float(np.median(times))
    dt_min = float(np.min(times))
    mean_img = float(bufs[(step_no[0] - 1) % F][..., :3].mean().item())   # the last (reduced) image
    # rays of a step: those of the last frame completed (the count moves by a fraction of a percent with `time`)
    last = [c.stats() for c in ctxs]
    if args.only_timed:
        if rank == 0:
            print(json.dumps({"only_timed": True, "workload": args.workload, "frames_in_flight": F, "steps": args.steps, "repeats": len(times),
                              "arrangement": "slots of one context" if use_slots else "one context per frame in flight",
                              "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "default"),
                              "ms_per_step": round(dt / args.steps * 1e3, 4), "ms_per_step_min": round(dt_min / args.steps * 1e3, 4),
                              "rays_per_step": int(round(float(np.mean([x["rays"] for x in last]))))}), flush=True)
        for c in ctxs:
            c.close()
        if world > 1:
            dist.destroy_process_group()
        return
    n_rep_other = max(1, min(len(times), 5))
    # the same K steps, one frame at a time: each frame complete before the next is enqueued
    if use_slots:
        times_one = timed(Slots(ctx, 1), repeats=n_rep_other)
    else:
        class OneAtATime(Contexts):
            def step(self):
                super().step()
                (comm_stream or streams[0]).synchronize()
        times_one = timed(OneAtATime([ctxs[0]]), repeats=n_rep_other)
    # ... with a camera that moves a little every frame (the schedule hints then come from a slightly different view)
    make_orbit_views()
    orbit[0] = True
    times_mv = timed(repeats=n_rep_other)
    orbit[0] = False
    # ... and with the schedule hints off (what a cut to another view gets: the previous frame's ray counts say nothing)
    os.environ["PT_AMD_NO_COST_HINTS"] = "1"
    ctxs_nh = [abi.Context(device=local_rank) for _ in range(len(ctxs))]
    del os.environ["PT_AMD_NO_COST_HINTS"]
    for c in ctxs_nh:
        c.upload_scene(scene)
    times_nh = timed(Slots(ctxs_nh[0], F) if use_slots else Contexts(ctxs_nh), repeats=n_rep_other)
    for c in ctxs_nh:
        c.close()
    extra = {}
    if use_slots:
        # what the plugin gets through the boundary: binary16 image in page-locked host memory every frame
        host16 = [ctx.host_array((H, W, 4), np.float16) for _ in range(F)]
        extra["ms_per_step_plugin_sync"] = round(float(np.median(timed(PluginSync(ctx, host16[0]), repeats=n_rep_other))) / args.steps * 1e3, 4)
        extra["ms_per_step_plugin_async"] = round(float(np.median(timed(Slots(ctx, F, host=host16, f16=True), repeats=n_rep_other))) / args.steps * 1e3, 4)
        extra.update(child_extra)

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
        rocprof_ms = None
        tf = ROOT / "profiles" / "traffic.json"
        if tf.exists():
            try:
                prof = json.loads(tf.read_text()).get(args.workload, {})
                traffic = prof.get("hbm_bytes_per_launch")
                rocprof_ms = prof.get("k_paths_ms_rocprof")
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
            **extra,
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
                       "frames_in_flight": F,
                       "arrangement": "one context, frames on its slots (pt_render_submit_device / pt_render_wait)" if use_slots
                                      else "one context per frame in flight (pt_render_device on torch streams)",
                       "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "default")},
            "rays_per_step": rays_total,
            "rays_definition": "closest-hit queries executed; the primary ray of a pixel is identical for all of its samples "
                               "(src/raytrace.wgsl:223) and is traced once, the reference would trace rays_reference_per_step",
            "rays_reference_per_step": rays_ref_total,
            "mrays_reference_equiv_per_s": round(rays_ref_total / (t_max / args.steps) / 1e6, 1),
            "samples_per_s": round(paths_total / (t_max / args.steps), 1),
            "image_mean": round(mean_img, 6),
            "kernels_ms": {"k_primary": round(float(np.mean(primary_ms)), 4), "k_paths": round(float(np.mean(paths_ms)), 4),
                           "frame": round(float(np.mean(kernel_ms)), 4)},
            # frac = frac_one_in_flight: the kernel alone on the chip (kernel_ms, HIP events of the library, one frame at a time);
            # frac_per_step: the same flop over ms_per_step, i.e. with the frames in flight whose kernels overlap (a step is
            # shorter than one kernel then); kernel_ms_rocprof: the kernel's average in the committed rocprofv3 trace
            "roofline": {"bound": "valu", "kernel": "k_paths", "achieved": round(achieved_tflops, 3), "peak": PEAK_FP32_VALU_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved_tflops / PEAK_FP32_VALU_TFLOPS, 5),
                         "frac_one_in_flight": round(achieved_tflops / PEAK_FP32_VALU_TFLOPS, 5),
                         "frac_per_step": round(flops / (ms_per_step * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS, 5),
                         "kernel_ms_rocprof": rocprof_ms, "frac_rocprof": round(flops / (rocprof_ms * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS, 5) if rocprof_ms else None,
                         "traffic": traffic,
                         "kernel_ms": round(k_ms, 4), "algorithmic_flop_per_launch": int(flops),
                         "on_chip_bytes_per_launch": int(abytes),
                         "hbm": {"compulsory_bytes_per_launch": int(compulsory), "achieved_GBs": round(compulsory / (k_ms * 1e-3) / 1e9, 2),
                                 "peak_GBs": PEAK_HBM_GBS, "frac": round(compulsory / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 6)},
                         "valu_issue": issue,
                         "counts": work},
        }
        if world > 1:
            # device time of one frame's collective (all-reduce + scale), HIP events on its stream; it runs beside the tracing
            # of later frames, so it is not a share of ms_per_step
            line["comm_ms_per_step"] = round(float(np.median(comm_ms)), 4) if comm_ms else None
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(scene, min(args.cpu_spp, scene.samples))
        print(json.dumps(line), flush=True)

    for c in ctxs:
        c.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

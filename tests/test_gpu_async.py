"""Frames in flight on ONE context (pt_render_submit / pt_render_submit_device / pt_render_wait), through the C ABI.

The asynchronous pair stands where `RayTraceNode::run` stands in the reference (src/shader.rs:121-213: the pass is
recorded, the renderer submits it later). Whatever the number of frames in flight, every image, aux plane and ray
count must be the one the synchronous call gives, i.e. the CPU restatement's, bit for bit; object uploads between
submits (extract_visible rewrites the objects every frame, src/extract.rs:297-346) must not disturb frames already
submitted; accumulated frames add up in submit order.
"""
import dataclasses

import numpy as np
import pytest

from path_tracing_amd import abi, scenes
from tests.oracle_lib import oracle_render

pytestmark = pytest.mark.gpu

TIMES = [0.0, 1.5, 2.5, 3.5, 0.5, 4.5, 5.5]


def _scene(name):
    return {"cornell": lambda: scenes.cornell_box(96, 64, 8, 10),                      # staged in LDS, sample sweeps
            "boxes": lambda: scenes.boxes(128, 72, 16, 10),
            "glb": lambda: scenes.scene_glb_textured(96, 56, 8, 10),                    # object list + BLAS
            "instanced": lambda: scenes.instanced_boxes(n_objects=600, width=96, height=64, samples=4, extent=6.0)}[name]()   # TLAS


@pytest.mark.parametrize("name", ["cornell", "boxes", "glb", "instanced"])
@pytest.mark.parametrize("in_flight", [1, 3, 8])
def test_frames_in_flight_match_the_oracle(name, in_flight):
    scene = _scene(name)
    refs = [oracle_render(scene, scene.frame(time=t)) for t in TIMES]
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        bufs = [ctx.host_array((scene.height, scene.width, 4), np.float32) for _ in range(in_flight)]
        done = 0
        for i, t in enumerate(TIMES + [None] * in_flight):
            slot = i % in_flight
            if i >= in_flight:      # the frame submitted `in_flight` steps ago
                j = i - in_flight
                ctx.wait(slot)
                ref = refs[j]
                assert np.array_equal(bufs[slot].view(np.uint32), ref["image"].view(np.uint32)), f"frame {j}"
                assert np.array_equal(ctx.read_aux(abi.PT_AUX_PRIMARY_FLAT_ID), ref["primary_flat_id"]), f"frame {j}"
                assert np.array_equal(ctx.read_aux(abi.PT_AUX_PRIMARY_OBJECT), ref["primary_object"]), f"frame {j}"
                assert np.array_equal(ctx.read_aux(abi.PT_AUX_RAY_COUNT), ref["ray_count"]), f"frame {j}"
                assert ctx.stats()["rays_reference"] == int(ref["ray_count"].astype(np.uint64).sum())
                done += 1
            if t is not None:
                bufs[slot][...] = -1.0
                ctx.upload_objects(scene.objects)      # the unchanged array, every frame, as the reference hands it over
                ctx.submit(scene.frame(time=t), slot, bufs[slot])
        assert done == len(TIMES)
        # the synchronous entry point on the same context afterwards
        assert np.array_equal(ctx.render(scene.frame(time=TIMES[1])).view(np.uint32), refs[1]["image"].view(np.uint32))


def test_f16_and_device_outputs_of_submitted_frames():
    import ctypes as C
    scene = scenes.cornell_box(64, 48, 4, 10)
    ref = oracle_render(scene, scene.frame())["image"]
    abi.load_library()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    d32, d16 = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(d32), ref.nbytes) == 0 and hip.hipMalloc(C.byref(d16), ref.nbytes // 2) == 0
    try:
        with abi.Context() as ctx:
            ctx.upload_scene(scene)
            half = ctx.host_array(ref.shape, np.float16)
            ctx.submit(scene.frame(flags=abi.PT_FRAME_OUTPUT_F16), 0, half)
            ctx.submit_device(scene.frame(), 1, d32.value)
            ctx.submit_device(scene.frame(flags=abi.PT_FRAME_OUTPUT_F16), 2, d16.value)
            for s in range(3):
                ctx.wait(s)
            got32 = np.empty(ref.shape, np.float32)
            got16 = np.empty(ref.shape, np.float16)
            assert hip.hipMemcpy(got32.ctypes.data_as(C.c_void_p), d32, got32.nbytes, 2) == 0
            assert hip.hipMemcpy(got16.ctypes.data_as(C.c_void_p), d16, got16.nbytes, 2) == 0
            half = half.copy()     # (the page-locked array goes away with the context)
        want16 = ref.astype(np.float16)
        assert np.array_equal(half.view(np.uint16), want16.view(np.uint16))
        assert np.array_equal(got16.view(np.uint16), want16.view(np.uint16))
        assert np.array_equal(got32.view(np.uint32), ref.view(np.uint32))
    finally:
        hip.hipFree(d32)
        hip.hipFree(d16)


def _moved(scene, k):
    """The scene's objects with every instance shifted a little (step k): same count, so the TLAS is refitted."""
    o = scene.objects.copy()
    d = np.float32(0.05 * k)
    o["local_to_world"][:, 12] += d          # translation x of the column-major matrix
    # world_to_local = inverse: for p' = M p + (d,0,0) the inverse subtracts the shift in local space
    for i in range(o.size):
        l2w = o["local_to_world"][i].reshape(4, 4).T.astype(np.float64)
        o["world_to_local"][i] = np.linalg.inv(l2w).T.reshape(16).astype(np.float32)
    return o


@pytest.mark.parametrize("name,n", [("instanced", 600), ("boxes", None)])
def test_object_uploads_between_submits_leave_frames_in_flight_alone(name, n):
    """Every frame gets its own object array, uploaded right before its submit while the earlier frames are still running;
    each image must be the one a fresh context renders for that array (which the restatement confirms on one of them)."""
    scene = (scenes.instanced_boxes(n_objects=n, width=320, height=200, samples=8, extent=6.0) if name == "instanced"
             else scenes.boxes(320, 200, 32, 10))
    steps = 6
    arrays = [_moved(scene, k) for k in range(steps)]
    want = []
    for k in range(steps):
        with abi.Context() as c:
            c.upload_scene(scene)
            c.upload_objects(arrays[k])
            want.append(c.render(scene.frame(time=float(k) + 0.5)))
    assert not np.array_equal(want[0], want[1])
    moved = dataclasses.replace(scene, objects=arrays[2])
    x0, y0, x1, y1 = 140, 90, 180, 110
    ref = oracle_render(moved, moved.frame(time=2.5), rect=(x0, y0, x1, y1))
    assert np.array_equal(want[2][y0:y1, x0:x1].view(np.uint32), ref["image"].view(np.uint32))
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        bufs = [ctx.host_array(want[0].shape, np.float32) for _ in range(3)]
        for rep in range(2):     # the second pass runs with schedule hints and refitted trees
            for k in range(steps + 3):
                slot = k % 3
                if k >= 3:
                    ctx.wait(slot)
                    assert np.array_equal(bufs[slot].view(np.uint32), want[k - 3].view(np.uint32)), f"pass {rep}, frame {k - 3}"
                if k < steps:
                    ctx.upload_objects(arrays[k])
                    ctx.submit(scene.frame(time=float(k) + 0.5), slot, bufs[slot])
            assert ctx.check_accel()["errors"] == 0


def test_accumulation_follows_submit_order():
    scene = scenes.cornell_box(64, 64, 4, 10)
    times = [0.0, 1.5, 2.5, 3.5, 4.5]
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        want = [ctx.render(scene.frame(time=t, flags=abi.PT_FRAME_ACCUMULATE | (abi.PT_FRAME_RESET if i == 0 else 0))) for i, t in enumerate(times)]
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        bufs = [ctx.host_array(want[0].shape, np.float32) for _ in times]
        for i, t in enumerate(times):
            ctx.submit(scene.frame(time=t, flags=abi.PT_FRAME_ACCUMULATE | (abi.PT_FRAME_RESET if i == 0 else 0)), i, bufs[i])
        for i in reversed(range(len(times))):
            ctx.wait(i)
        for i in range(len(times)):
            assert np.array_equal(bufs[i].view(np.uint32), want[i].view(np.uint32)), f"frame {i}"


def test_submit_and_wait_error_codes():
    scene = scenes.cornell_box(32, 32, 1, 2)
    with abi.Context() as ctx:
        buf = ctx.host_array((32, 32, 4), np.float32)
        with pytest.raises(abi.PtError) as e:
            ctx.submit(scene.frame(), 0, buf)               # no scene yet
        assert e.value.code == abi.PT_ERR_NOT_READY
        ctx.upload_scene(scene)
        with pytest.raises(abi.PtError) as e:
            ctx.wait(0)                                      # (the failed submit left nothing behind)
        assert e.value.code == abi.PT_ERR_NOT_READY
        with pytest.raises(abi.PtError) as e:
            ctx.submit(scene.frame(), abi.PT_MAX_FRAME_SLOTS, buf)
        assert e.value.code == abi.PT_ERR_INVALID_ARGUMENT
        with pytest.raises(abi.PtError) as e:
            ctx.wait(abi.PT_MAX_FRAME_SLOTS)
        assert e.value.code == abi.PT_ERR_INVALID_ARGUMENT
        ctx.submit(scene.frame(), 0, buf)
        with pytest.raises(abi.PtError) as e:
            ctx.submit(scene.frame(), 0, buf)               # the slot is taken
        assert e.value.code == abi.PT_ERR_INVALID_ARGUMENT and "pt_render_wait" in e.value.message
        ctx.wait(0)
        with pytest.raises(abi.PtError) as e:
            ctx.wait(0)
        assert e.value.code == abi.PT_ERR_NOT_READY
        ref = ctx.render(scene.frame())
        assert np.array_equal(buf.view(np.uint32), ref.view(np.uint32))
        # a context destroyed with frames in flight waits for them
        ctx.submit(scene.frame(samples=64, bounces=10), 1, buf)
        ctx.submit(scene.frame(samples=64, bounces=10), 2, buf)


def test_frames_in_flight_at_1080p_equal_one_at_a_time():
    """The headline configuration's size: three slots taking frames in turn, hints and split chunks active from each
    slot's second frame on; every image must be the one the synchronous call gives."""
    scene = scenes.scene_glb_textured(1920, 1080, 16, 10)
    times = [0.5, 1.5, 2.5]
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        want = [ctx.render(scene.frame(time=t)) for t in times]
        bufs = [ctx.host_array(want[0].shape, np.float32) for _ in range(3)]
        for rep in range(3):
            for s, t in enumerate(times):
                ctx.submit(scene.frame(time=t), s, bufs[s])
            for s in range(3):
                ctx.wait(s)
                assert np.array_equal(bufs[s].view(np.uint32), want[s].view(np.uint32)), f"pass {rep}, slot {s}"
                assert ctx.stats()["rays"] > 0


def test_frames_that_share_the_device_run_without_sample_sweeps():
    """A frame that has the device to itself runs its samples in sweeps and its most expensive chunks in parts; a frame
    that is scheduled beside other frames runs every chunk in one item, so that no wave of it ever polls for another wave's
    sweep while the hardware shares the compute units between queues. pt_debug_slot_state shows which way a frame was
    scheduled; the images are the same either way."""
    import time
    scene = scenes.scene_glb_textured(1920, 1080, 16, 10)
    times = [0.5, 1.5, 2.5]
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        want = [ctx.render(scene.frame(time=t)) for t in times]
        bufs = [ctx.host_array(want[0].shape, np.float32) for _ in range(3)]
        time.sleep(0.1)   # (the device remembers for 20 ms that frames shared it)
        for _ in range(4):  # (a caller's first two frames on a slot run without sweeps: it might be about to keep several in flight)
            ctx.submit(scene.frame(time=times[0]), 0, bufs[0])
            ctx.wait(0)
            assert np.array_equal(bufs[0].view(np.uint32), want[0].view(np.uint32))
        st = ctx.slot_state(0)
        records = [st[4 + 8 * k: 12 + 8 * k] for k in range(2)]
        assert all(int(r[6]) == 0 for r in records), records            # nobody else was running
        assert any(int(r[7]) > 0 for r in records), records             # chunks in parts: the frame was scheduled with sweeps
        for rep in range(3):
            for s, t in enumerate(times):
                ctx.submit(scene.frame(time=t), s, bufs[s])
            for s in range(3):
                ctx.wait(s)
                assert np.array_equal(bufs[s].view(np.uint32), want[s].view(np.uint32)), f"pass {rep}, slot {s}"
        shared = 0
        for s in range(3):
            st = ctx.slot_state(s)
            for k in range(2):
                r = st[4 + 8 * k: 12 + 8 * k]
                if int(r[6]) & 0x100:
                    shared += 1
                    assert int(r[7]) == 0, (s, k, r)
        assert shared >= 3, shared


def test_object_list_scenes_are_served_for_throughput_from_five_frames_in_flight():
    """With few frames in flight the mesh of scene.glb has its BLAS in LDS (shorter chains); from five on the path
    kernel runs without it (more workgroups per compute unit). pt_debug_slot_state tells which; the images are the same."""
    scene = scenes.scene_glb_textured(1920, 1080, 16, 10)
    times = [0.5 + k for k in range(8)]
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        want = [ctx.render(scene.frame(time=t)) for t in times]
        bufs = [ctx.host_array(want[0].shape, np.float32) for _ in range(8)]
        for k_slots, deep in ((3, False), (8, True)):
            for rep in range(3):     # (the first pass of a slot has no cost hints, and the BLAS goes into LDS only with them)
                for s in range(k_slots):
                    ctx.submit(scene.frame(time=times[s]), s, bufs[s])
                for s in range(k_slots):
                    ctx.wait(s)
                    assert np.array_equal(bufs[s].view(np.uint32), want[s].view(np.uint32)), f"{k_slots} slots, pass {rep}, slot {s}"
            how = [int(ctx.slot_state(s)[3]) for s in range(k_slots)]
            assert all(h & 0x80000000 for h in how), how
            last = how[k_slots - 1]                     # enqueued with every other slot in flight
            assert ((last >> 8) & 15) == k_slots - 1, hex(last)
            assert bool(last & 0x20) == deep, hex(last)
            assert (last & 15) == (0 if deep else 3), hex(last)     # object list without / with a BLAS in LDS


def test_gpu_side_ordering_of_a_slot_against_a_consumer_stream():
    """pt_render_stream_wait / pt_render_slot_wait_for: a consumer on a stream of its own (here: a slow chain of device copies
    that ends by saving the slot's image) reads every frame behind the frame and before the slot's next frame overwrites it,
    without the host waiting for either. The node of the reference sits in a render graph whose later nodes read its image on
    the GPU (src/shader.rs:83-88)."""
    import ctypes as C
    scene = scenes.boxes(160, 90, 8, 10)
    n_frames, k_slots = 7, 2
    times = [0.5 + i for i in range(n_frames)]
    abi.load_library()
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipFree.argtypes = [C.c_void_p]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    hip.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    D2D, D2H = 3, 2
    nbytes = scene.height * scene.width * 16
    junk_bytes = 256 << 20
    ptrs = [C.c_void_p() for _ in range(k_slots + 3)]
    sizes = [nbytes] * k_slots + [nbytes * n_frames, junk_bytes, junk_bytes]
    for p, sz in zip(ptrs, sizes):
        assert hip.hipMalloc(C.byref(p), sz) == 0
    bufs, archive, junk_a, junk_b = ptrs[:k_slots], ptrs[k_slots], ptrs[k_slots + 1], ptrs[k_slots + 2]
    consumer = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(consumer)) == 0
    try:
        with abi.Context() as ctx:
            ctx.upload_scene(scene)
            want = [ctx.render(scene.frame(time=t)) for t in times]
            assert not np.array_equal(want[0], want[k_slots])
            with pytest.raises(abi.PtError) as e:
                ctx.stream_wait(0, consumer.value)          # nothing submitted to the slot
            assert e.value.code == abi.PT_ERR_NOT_READY
            for i, t in enumerate(times):
                s = i % k_slots
                if i >= k_slots:
                    ctx.wait(s)                                  # (frame i - K is long done; its consumer may not be)
                ctx.slot_wait_for(s, consumer.value)           # frame i overwrites what the consumer of frame i - K reads
                ctx.submit_device(scene.frame(time=t), s, bufs[s].value)
                ctx.stream_wait(s, consumer.value)
                for _ in range(4):                              # the consumer is slow: about a millisecond before it reads
                    assert hip.hipMemcpyAsync(junk_b, junk_a, junk_bytes, D2D, consumer) == 0
                assert hip.hipMemcpyAsync(C.c_void_p(archive.value + i * nbytes), bufs[s], nbytes, D2D, consumer) == 0
            for s in range(k_slots):
                ctx.wait(s)
            assert hip.hipStreamSynchronize(consumer) == 0
            got = np.empty((n_frames, scene.height, scene.width, 4), np.float32)
            assert hip.hipMemcpy(got.ctypes.data_as(C.c_void_p), archive, got.nbytes, D2H) == 0
            for i in range(n_frames):
                assert np.array_equal(got[i].view(np.uint32), want[i].view(np.uint32)), f"frame {i}"
    finally:
        hip.hipStreamDestroy(consumer)
        for p in ptrs:
            hip.hipFree(p)


@pytest.mark.parametrize("name,iterations,seed", [("inst", 500, 11), ("boxes", 500, 12)])
def test_random_order_of_uploads_frames_and_queries(name, iterations, seed):
    """tools/gpu_soak.py: object uploads (five versions of the array), frames on four slots, synchronous frames and ray
    queries in a seeded random order on ONE context; every finished image must be the one a fresh context renders for the
    same object array and time (src/extract.rs:297-346 rewrites the objects every frame, src/shader.rs:121-213 draws)."""
    import subprocess
    import sys
    from pathlib import Path
    tool = Path(__file__).resolve().parent.parent / "tools" / "gpu_soak.py"
    r = subprocess.run([sys.executable, str(tool), name, str(iterations), str(seed)], capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    assert "ok:" in r.stdout

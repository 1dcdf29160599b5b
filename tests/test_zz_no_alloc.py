"""pt_render_device_noalloc: the frame of pt_render_device rendered without any device-memory allocation, its work
memory being a fixed per-GPU array inside the library. Called through the C ABI and compared bit for bit with
pt_render_device on the same context: two calls in a row, the second with a smaller frame (the array keeps the first
call's contents), for a scene staged in LDS (sample sweeps, parked pixel state) and for one walked through the TLAS."""
import ctypes as C

import numpy as np
import pytest

from path_tracing_amd import abi, scenes

pytestmark = pytest.mark.gpu


class Hip:
    """The few HIP runtime calls the test itself needs for its own image buffers (the entry point under test makes none)."""

    def __init__(self):
        abi.load_library()   # brings in the HIP runtime the library is linked against
        self.lib = C.CDLL("libamdhip64.so")
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipFree.argtypes = [C.c_void_p]
        self.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.lib.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.lib.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        self.lib.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.lib.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.lib.hipStreamDestroy.argtypes = [C.c_void_p]

    def check(self, rc, what):
        assert rc == 0, f"{what} failed with HIP error {rc}"

    def malloc(self, n):
        p = C.c_void_p()
        self.check(self.lib.hipMalloc(C.byref(p), n), "hipMalloc")
        return p

    def free(self, p):
        self.check(self.lib.hipFree(p), "hipFree")

    def fill(self, p, byte, n):
        self.check(self.lib.hipMemset(p, byte, n), "hipMemset")

    def to_host(self, p, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        self.check(self.lib.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, out.nbytes, 2), "hipMemcpy")   # 2 = device to host
        return out

    def free_bytes(self):
        f, t = C.c_size_t(), C.c_size_t()
        self.check(self.lib.hipMemGetInfo(C.byref(f), C.byref(t)), "hipMemGetInfo")
        return f.value


def _frames(scene):
    big = scene.frame()
    small = scene.frame(width=scene.width // 2 + 3, height=scene.height // 2 + 5)   # ragged: partial tiles
    return [big, small]


@pytest.mark.parametrize("name", ["boxes", "instanced", "glb"])
def test_noalloc_render_matches_render_device(name):
    scene = {"boxes": lambda: scenes.boxes(320, 200, 32, 10),                       # staged in LDS, three sample sweeps
             "instanced": lambda: scenes.instanced_boxes(n_objects=600, width=192, height=128, samples=4, extent=6.0),   # TLAS walk
             "glb": lambda: scenes.scene_glb_textured(160, 96, 16, 10)}[name]()     # object list + BLAS walk, sweeps
    hip = Hip()
    frames = _frames(scene)
    n_max = max(f.width * f.height for f in frames)
    d_ref, d_out = hip.malloc(16 * n_max), hip.malloc(16 * n_max)
    try:
        with abi.Context() as ctx:
            ctx.upload_scene(scene)
            want = []
            for f in frames:
                ctx.render_device(f, d_ref.value)
                want.append(hip.to_host(d_ref, (f.height, f.width, 4), np.float32))
                assert np.isfinite(want[-1]).all() and want[-1][..., :3].max() > 0
            # one call to let the runtime finish whatever it sets up on a kernel's first launch, then the two calls that
            # are checked: big frame, then the smaller one into work memory that still holds the big frame's state
            ctx.render_device_noalloc(frames[0], d_out.value)
            free_before = hip.free_bytes()
            for k, f in enumerate(frames):
                hip.fill(d_out, 0xFF, 16 * n_max)
                ctx.render_device_noalloc(f, d_out.value)
                got = hip.to_host(d_out, (f.height, f.width, 4), np.float32)
                assert np.array_equal(got.view(np.uint32), want[k].view(np.uint32)), f"{name}: call {k} differs from pt_render_device"
            assert hip.free_bytes() == free_before, "device memory was allocated or freed during pt_render_device_noalloc"
            # the allocating entry point still works on the same context afterwards, and so does the statistics path
            ctx.render_device(frames[1], d_ref.value)
            again = hip.to_host(d_ref, (frames[1].height, frames[1].width, 4), np.float32)
            assert np.array_equal(again.view(np.uint32), want[1].view(np.uint32))
            assert ctx.stats()["rays"] > 0
    finally:
        hip.free(d_ref)
        hip.free(d_out)


def test_noalloc_limits_are_reported():
    scene = scenes.cornell_box(512, 512, 1, 2)
    hip = Hip()
    d_out = hip.malloc(16 * 520 * 520)
    try:
        with abi.Context() as ctx:
            ctx.upload_scene(scene)
            ctx.render_device_noalloc(scene.frame(), d_out.value)    # 512 x 512 = PT_NOALLOC_MAX_PIXELS: the largest frame
            img = hip.to_host(d_out, (512, 512, 4), np.float32)
            assert np.array_equal(img.view(np.uint32), ctx.render(scene.frame()).view(np.uint32))
            for f in (scene.frame(width=520, height=512), scene.frame(width=513, height=512)):   # 513 rounds up to 520
                with pytest.raises(abi.PtError) as e:
                    ctx.render_device_noalloc(f, d_out.value)
                assert e.value.code == abi.PT_ERR_UNSUPPORTED and "PT_NOALLOC_MAX_PIXELS" in e.value.message
            with pytest.raises(abi.PtError) as e:
                ctx.render_device_noalloc(scene.frame(flags=abi.PT_FRAME_ACCUMULATE), d_out.value)
            assert e.value.code == abi.PT_ERR_UNSUPPORTED
        with abi.Context() as ctx:   # no scene: the ordinary error, not a fault
            with pytest.raises(abi.PtError) as e:
                ctx.render_device_noalloc(scene.frame(), d_out.value)
            assert e.value.code == abi.PT_ERR_NOT_READY
    finally:
        hip.free(d_out)


def test_scene_changes_and_destroy_right_after_an_asynchronous_noalloc_frame():
    """On a caller's stream the call returns while the frame is still running. An upload that follows at once must not
    change what that frame reads, and destroying the context must wait for it."""
    a = scenes.instanced_boxes(n_objects=600, width=256, height=160, samples=16, extent=6.0)
    b = scenes.instanced_boxes(n_objects=600, width=256, height=160, samples=16, extent=6.0, seed=77)
    hip = Hip()
    stream = C.c_void_p()
    hip.check(hip.lib.hipStreamCreate(C.byref(stream)), "hipStreamCreate")
    f = a.frame()
    n = f.width * f.height
    d_ref, d_out = hip.malloc(16 * n), hip.malloc(16 * n)
    try:
        with abi.Context() as ctx:
            ctx.upload_scene(a)
            ctx.render_device(f, d_ref.value)
            want_a = hip.to_host(d_ref, (f.height, f.width, 4), np.float32)
            ctx.upload_objects(b.objects)
            ctx.render_device(f, d_ref.value)
            want_b = hip.to_host(d_ref, (f.height, f.width, 4), np.float32)
            assert not np.array_equal(want_a, want_b)
            ctx.upload_objects(a.objects)
            for k in range(3):
                hip.fill(d_out, 0xFF, 16 * n)
                ctx.render_device_noalloc(f, d_out.value, stream.value)     # returns at once
                ctx.upload_objects(b.objects if k % 2 == 0 else a.objects)  # ... and the scene changes under it
                hip.check(hip.lib.hipStreamSynchronize(stream), "hipStreamSynchronize")
                got = hip.to_host(d_out, (f.height, f.width, 4), np.float32)
                want = want_a if k % 2 == 0 else want_b
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"call {k}"
            ctx.upload_meshes(a.meshes, a.indices, a.vertices)              # waits for everything, rebuilds in place
            ctx.render_device_noalloc(f, d_out.value, stream.value)
        # (the context is gone: pt_destroy waited for the frame)
        got = hip.to_host(d_out, (f.height, f.width, 4), np.float32)
        assert np.array_equal(got.view(np.uint32), want_b.view(np.uint32))
    finally:
        hip.check(hip.lib.hipStreamSynchronize(stream), "hipStreamSynchronize")
        hip.lib.hipStreamDestroy(stream)
        hip.free(d_ref)
        hip.free(d_out)

"""ctypes / numpy mirror of include/pt_amd.h and loader for libpt_amd.so.

The numpy dtypes are the byte layouts of the reference's storage buffers
(reference src/data.rs:15-73): scene arrays built with them can be handed to the
C ABI without any repacking, which is what the Rust host does with its
`StorageBuffer<Vec<...>>` contents.

There is no CPU fallback: if the HIP library is missing or fails to load,
`load_library()` raises.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

PT_ABI_VERSION = 1
PT_U32_MAX = 0xFFFFFFFF

PT_OK = 0
PT_ERR_INVALID_ARGUMENT = -1
PT_ERR_INVALID_SCENE = -2
PT_ERR_NOT_READY = -3
PT_ERR_DEVICE = -4
PT_ERR_OUT_OF_MEMORY = -5
PT_ERR_UNSUPPORTED = -6
PT_ERR_NO_DEVICE = -7

PT_FRAME_ACCUMULATE = 0x1
PT_FRAME_RESET = 0x2
PT_FRAME_OUTPUT_F16 = 0x4

PT_ACCEL_AUTO, PT_ACCEL_BRUTE, PT_ACCEL_BVH = 0, 1, 2
PT_TIE_REFERENCE, PT_TIE_LOWEST = 0, 1

PT_NOALLOC_MAX_PIXELS = 1 << 18
PT_MAX_FRAME_SLOTS = 8

PT_AUX_PRIMARY_OBJECT, PT_AUX_PRIMARY_FLAT_ID, PT_AUX_PRIMARY_T, PT_AUX_RAY_COUNT = 0, 1, 2, 3

# ---- wire structs as numpy dtypes (align=False; explicit padding) ------------
OBJECT_DT = np.dtype([("local_to_world", "<f4", (16,)), ("world_to_local", "<f4", (16,)),
                      ("mat", "<u4"), ("mesh", "<u4"), ("_pad", "<u4", (2,))])
MESH_DT = np.dtype([("aabb_min", "<f4", (3,)), ("_pad0", "<u4"), ("aabb_max", "<f4", (3,)),
                    ("ihead", "<u4"), ("vhead", "<u4"), ("tri_count", "<u4"), ("_pad1", "<u4", (2,))])
VERTEX_DT = np.dtype([("position", "<f4", (3,)), ("_pad0", "<u4"), ("normal", "<f4", (3,)), ("_pad1", "<u4"),
                      ("uv", "<f4", (2,)), ("_pad2", "<u4", (2,))])
MATERIAL_DT = np.dtype([("albedo", "<f4", (3,)), ("albedo_texture", "<u4"),
                        ("emissive", "<f4", (3,)), ("emissive_texture", "<u4"),
                        ("roughness", "<f4"), ("metallic", "<f4"), ("metallic_roughness_texture", "<u4"),
                        ("reflectance", "<f4"), ("normal_map_texture", "<u4"), ("_pad", "<u4", (3,))])
TEXTURE_DT = np.dtype([("width", "<u4"), ("height", "<u4"), ("offset", "<u4"), ("format", "<u4")])
RAY_DT = np.dtype([("pos", "<f4", (3,)), ("dir", "<f4", (3,))])
HIT_DT = np.dtype([("object", "<u4"), ("triangle", "<u4"), ("flat_id", "<u4"), ("t", "<f4"),
                   ("p", "<f4", (3,)), ("n", "<f4", (3,)), ("uv", "<f4", (2,))])

assert OBJECT_DT.itemsize == 144 and MESH_DT.itemsize == 48 and VERTEX_DT.itemsize == 48
assert MATERIAL_DT.itemsize == 64 and TEXTURE_DT.itemsize == 16
assert RAY_DT.itemsize == 24 and HIT_DT.itemsize == 48


class PtSettings(C.Structure):
    _fields_ = [("bounces", C.c_uint32), ("samples", C.c_uint32), ("_pad", C.c_uint32 * 2),
                ("sky_color", C.c_float * 4)]


class PtFrame(C.Structure):
    _fields_ = [("world_from_view", C.c_float * 16), ("view_from_clip", C.c_float * 16),
                ("width", C.c_uint32), ("height", C.c_uint32), ("time", C.c_float), ("flags", C.c_uint32),
                ("settings", PtSettings),
                ("part_index", C.c_uint32), ("part_count", C.c_uint32), ("_reserved", C.c_uint32 * 2)]


class PtConfig(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("device", C.c_int32), ("accel", C.c_uint32),
                ("tie_rule", C.c_uint32), ("flags", C.c_uint32), ("_reserved", C.c_uint32 * 3)]


PT_MULTI_MAX_DEVICES = 16
PT_MULTI_TILES, PT_MULTI_VIRTUAL_FRAMES = 0, 1
PT_MULTI_FLAG_HOST_SUM = 0x1
PT_MULTI_FLAG_FORCE_COLLECTIVE = 0x2


class PtMultiConfig(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("n_devices", C.c_uint32), ("devices", C.c_int32 * PT_MULTI_MAX_DEVICES),
                ("mode", C.c_uint32), ("accel", C.c_uint32), ("tie_rule", C.c_uint32), ("flags", C.c_uint32),
                ("_reserved", C.c_uint32 * 4)]


class PtStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("rays_reference", C.c_uint64), ("paths", C.c_uint64),
                ("xforms", C.c_uint64), ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64),
                ("shades", C.c_uint64), ("hits", C.c_uint64), ("trace_ms", C.c_double), ("primary_ms", C.c_double), ("paths_ms", C.c_double),
                ("build_ms", C.c_double),
                ("trace_launches", C.c_uint32), ("accel_used", C.c_uint32),
                ("cycles_refill", C.c_uint64), ("cycles_trace", C.c_uint64), ("cycles_shade", C.c_uint64),
                ("wave_rounds", C.c_uint64), ("lane_rounds", C.c_uint64),
                ("wave_life_sum", C.c_uint64), ("wave_life_max", C.c_uint64), ("waves", C.c_uint64),
                ("wave_life_hist", C.c_uint32 * 32),
                ("primary_xforms", C.c_uint64), ("primary_box_tests", C.c_uint64), ("primary_tri_tests", C.c_uint64),
                ("primary_hits", C.c_uint64), ("object_passes", C.c_uint64)]

    def as_dict(self):
        return {n: (list(getattr(self, n)) if n == "wave_life_hist" else getattr(self, n)) for n, _ in self._fields_}


assert C.sizeof(PtSettings) == 32 and C.sizeof(PtFrame) == 192 and C.sizeof(PtConfig) == 32

# every symbol include/pt_amd.h declares
EXPORTS = ["pt_create", "pt_destroy", "pt_last_error", "pt_upload_meshes", "pt_upload_materials",
           "pt_upload_objects", "pt_render", "pt_render_device", "pt_query", "pt_read_aux",
           "pt_set_stats_detail", "pt_get_stats", "pt_device_info", "pt_debug_check_accel", "pt_debug_wave_log", "pt_debug_slot_state", "pt_host_alloc", "pt_host_free",
           "pt_render_device_noalloc", "pt_render_submit", "pt_render_submit_device", "pt_render_wait",
           "pt_render_stream_wait", "pt_render_slot_wait_for",
           "pt_multi_create", "pt_multi_destroy", "pt_multi_last_error", "pt_multi_upload_meshes", "pt_multi_upload_materials",
           "pt_multi_upload_objects", "pt_multi_render", "pt_multi_context", "pt_multi_rccl_version"]

LIB_NAME = "libpt_amd.so"


def library_path() -> Path:
    env = os.environ.get("PT_AMD_LIBRARY")
    if env:
        return Path(env)
    return Path(__file__).resolve().parent / LIB_NAME


_lib = None


def load_library():
    """dlopen libpt_amd.so and declare the prototypes. Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not path.exists():
        raise RuntimeError(f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           f"(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(str(path))
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
    lib.pt_create.argtypes = [C.POINTER(PtConfig), C.POINTER(vp)]
    lib.pt_destroy.argtypes = [vp]
    lib.pt_last_error.argtypes = [vp]
    lib.pt_last_error.restype = C.c_char_p
    lib.pt_upload_meshes.argtypes = [vp, vp, u32, vp, u32, vp, u32]
    lib.pt_upload_materials.argtypes = [vp, vp, u32, vp, u32, vp, u32]
    lib.pt_upload_objects.argtypes = [vp, vp, u32]
    lib.pt_render.argtypes = [vp, C.POINTER(PtFrame), vp]
    lib.pt_render_device.argtypes = [vp, C.POINTER(PtFrame), vp, vp]
    lib.pt_render_device_noalloc.argtypes = [vp, C.POINTER(PtFrame), vp, vp]
    lib.pt_render_submit.argtypes = [vp, C.POINTER(PtFrame), u32, vp]
    lib.pt_render_submit_device.argtypes = [vp, C.POINTER(PtFrame), u32, vp]
    lib.pt_render_wait.argtypes = [vp, u32]
    lib.pt_render_stream_wait.argtypes = [vp, u32, vp]
    lib.pt_render_slot_wait_for.argtypes = [vp, u32, vp]
    lib.pt_query.argtypes = [vp, vp, u32, vp]
    lib.pt_read_aux.argtypes = [vp, u32, vp, C.c_size_t]
    lib.pt_set_stats_detail.argtypes = [vp, i32]
    lib.pt_get_stats.argtypes = [vp, C.POINTER(PtStats)]
    lib.pt_device_info.argtypes = [vp, C.c_char_p, C.c_size_t]
    lib.pt_debug_check_accel.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.pt_debug_wave_log.argtypes = [vp, vp, C.c_size_t, C.POINTER(C.c_uint32)]
    lib.pt_debug_slot_state.argtypes = [vp, u32, vp, u32]
    lib.pt_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.pt_host_free.argtypes = [vp, vp]
    lib.pt_multi_create.argtypes = [C.POINTER(PtMultiConfig), C.POINTER(vp)]
    lib.pt_multi_destroy.argtypes = [vp]
    lib.pt_multi_last_error.argtypes = [vp]
    lib.pt_multi_last_error.restype = C.c_char_p
    lib.pt_multi_upload_meshes.argtypes = [vp, vp, u32, vp, u32, vp, u32]
    lib.pt_multi_upload_materials.argtypes = [vp, vp, u32, vp, u32, vp, u32]
    lib.pt_multi_upload_objects.argtypes = [vp, vp, u32]
    lib.pt_multi_render.argtypes = [vp, C.POINTER(PtFrame), vp]
    lib.pt_multi_context.argtypes = [vp, u32, C.POINTER(vp)]
    lib.pt_multi_rccl_version.argtypes = [C.POINTER(C.c_int)]
    for name in EXPORTS:
        if name not in ("pt_last_error", "pt_multi_last_error"):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


class PtError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"pt_amd error {code}: {message}")
        self.code = code
        self.message = message


def _ptr(a: np.ndarray | None):
    if a is None or a.size == 0:
        return None
    return a.ctypes.data_as(C.c_void_p)


def make_frame(world_from_view, view_from_clip, width, height, *, time=0.0, bounces=10, samples=2,
               sky_color=(0.0, 0.0, 0.0), flags=0, part_index=0, part_count=1) -> PtFrame:
    """Fill a PtFrame. Matrices: 16 floats column-major (glam order)."""
    f = PtFrame()
    wfv = np.asarray(world_from_view, dtype=np.float32).reshape(16)
    vfc = np.asarray(view_from_clip, dtype=np.float32).reshape(16)
    for i in range(16):
        f.world_from_view[i] = float(wfv[i])
        f.view_from_clip[i] = float(vfc[i])
    f.width, f.height = int(width), int(height)
    f.time = float(np.float32(time))
    f.flags = int(flags)
    f.settings.bounces = int(bounces)
    f.settings.samples = int(samples)
    sky = list(sky_color) + [1.0] * (4 - len(sky_color))
    for i in range(4):
        f.settings.sky_color[i] = float(np.float32(sky[i]))
    f.part_index, f.part_count = int(part_index), int(part_count)
    return f


class MultiContext:
    """pt_multi: several GPUs behind one handle (the scene on every GPU, a frame partitioned over them, one RCCL sum)."""

    def __init__(self, devices, mode: int = PT_MULTI_TILES, accel: int = PT_ACCEL_AUTO, tie_rule: int = PT_TIE_REFERENCE, flags: int = 0):
        self._lib = load_library()
        cfg = PtMultiConfig(abi_version=PT_ABI_VERSION, n_devices=len(devices), mode=mode, accel=accel, tie_rule=tie_rule, flags=flags)
        for i, d in enumerate(devices):
            cfg.devices[i] = int(d)
        h = C.c_void_p()
        rc = self._lib.pt_multi_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise PtError(rc, (self._lib.pt_multi_last_error(None) or b"").decode())
        self._h = h
        self.n = len(devices)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pt_multi_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise PtError(rc, (self._lib.pt_multi_last_error(self._h) or b"").decode())

    def upload_scene(self, scene):
        meshes = np.ascontiguousarray(scene.meshes, dtype=MESH_DT)
        indices = np.ascontiguousarray(scene.indices, dtype=np.uint32)
        vertices = np.ascontiguousarray(scene.vertices, dtype=VERTEX_DT)
        self._check(self._lib.pt_multi_upload_meshes(self._h, _ptr(meshes), meshes.size, _ptr(indices), indices.size, _ptr(vertices), vertices.size))
        materials = np.ascontiguousarray(scene.materials, dtype=MATERIAL_DT)
        textures = np.ascontiguousarray(scene.textures, dtype=TEXTURE_DT)
        texture_data = np.ascontiguousarray(scene.texture_data, dtype=np.float32)
        self._check(self._lib.pt_multi_upload_materials(self._h, _ptr(materials), materials.size, _ptr(textures), textures.size,
                                                        _ptr(texture_data), texture_data.size))
        objects = np.ascontiguousarray(scene.objects, dtype=OBJECT_DT)
        self._check(self._lib.pt_multi_upload_objects(self._h, _ptr(objects), objects.size))

    def render(self, frame: PtFrame) -> np.ndarray:
        dt = np.float16 if frame.flags & PT_FRAME_OUTPUT_F16 else np.float32
        out = np.empty((frame.height, frame.width, 4), dtype=dt)
        self._check(self._lib.pt_multi_render(self._h, C.byref(frame), _ptr(out)))
        return out

    def rays(self, k: int) -> int:
        """Closest-hit queries GPU k executed for the last frame."""
        h = C.c_void_p()
        self._check(self._lib.pt_multi_context(self._h, k, C.byref(h)))
        st = PtStats()
        rc = self._lib.pt_get_stats(h, C.byref(st))
        if rc != 0:
            raise PtError(rc, (self._lib.pt_last_error(h) or b"").decode())
        return int(st.rays)


def rccl_version() -> int:
    lib = load_library()
    v = C.c_int(0)
    rc = lib.pt_multi_rccl_version(C.byref(v))
    if rc != 0:
        raise PtError(rc, (lib.pt_multi_last_error(None) or b"").decode())
    return v.value


class Context:
    """Thin RAII wrapper over pt_ctx. All methods raise PtError on a negative return."""

    def __init__(self, device: int = 0, accel: int = PT_ACCEL_AUTO, tie_rule: int = PT_TIE_REFERENCE):
        self._lib = load_library()
        cfg = PtConfig(abi_version=PT_ABI_VERSION, device=device, accel=accel, tie_rule=tie_rule, flags=0)
        h = C.c_void_p()
        rc = self._lib.pt_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise PtError(rc, (self._lib.pt_last_error(None) or b"").decode())
        self._h = h
        self.size = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise PtError(rc, (self._lib.pt_last_error(self._h) or b"").decode())

    def upload_meshes(self, meshes, indices, vertices):
        meshes = np.ascontiguousarray(meshes, dtype=MESH_DT)
        indices = np.ascontiguousarray(indices, dtype=np.uint32)
        vertices = np.ascontiguousarray(vertices, dtype=VERTEX_DT)
        self._check(self._lib.pt_upload_meshes(self._h, _ptr(meshes), meshes.size, _ptr(indices), indices.size,
                                               _ptr(vertices), vertices.size))

    def upload_materials(self, materials, textures, texture_data):
        materials = np.ascontiguousarray(materials, dtype=MATERIAL_DT)
        textures = np.ascontiguousarray(textures, dtype=TEXTURE_DT)
        texture_data = np.ascontiguousarray(texture_data, dtype=np.float32)
        self._check(self._lib.pt_upload_materials(self._h, _ptr(materials), materials.size, _ptr(textures),
                                                  textures.size, _ptr(texture_data), texture_data.size))

    def upload_objects(self, objects):
        objects = np.ascontiguousarray(objects, dtype=OBJECT_DT)
        self._check(self._lib.pt_upload_objects(self._h, _ptr(objects), objects.size))

    def upload_scene(self, scene):
        self.upload_meshes(scene.meshes, scene.indices, scene.vertices)
        self.upload_materials(scene.materials, scene.textures, scene.texture_data)
        self.upload_objects(scene.objects)

    def host_array(self, shape, dtype) -> np.ndarray:
        """A numpy array over page-locked memory owned by the context (valid until close()): the staging buffer to
        pass as `out=` to render() frame after frame."""
        dtype = np.dtype(dtype)
        n = int(np.prod(shape)) * dtype.itemsize
        p = C.c_void_p()
        self._check(self._lib.pt_host_alloc(self._h, max(n, 1), C.byref(p)))
        buf = (C.c_char * max(n, 1)).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def render(self, frame: PtFrame, out: np.ndarray | None = None) -> np.ndarray:
        """One frame into a host array (a new one, or `out` when the caller keeps a staging buffer across frames)."""
        dt = np.float16 if frame.flags & PT_FRAME_OUTPUT_F16 else np.float32
        if out is None:
            out = np.empty((frame.height, frame.width, 4), dtype=dt)
        if out.dtype != dt or out.shape != (frame.height, frame.width, 4) or not out.flags.c_contiguous:
            raise ValueError("render: `out` must be a C-contiguous (height, width, 4) array of the frame's output type")
        self._check(self._lib.pt_render(self._h, C.byref(frame), _ptr(out) if out.size else None))
        self.size = (frame.height, frame.width)
        return out

    def render_device(self, frame: PtFrame, device_ptr: int, stream: int | None = None):
        self._check(self._lib.pt_render_device(self._h, C.byref(frame), C.c_void_p(device_ptr),
                                               C.c_void_p(stream) if stream else None))
        self.size = (frame.height, frame.width)

    def render_device_noalloc(self, frame: PtFrame, device_ptr: int, stream: int | None = None):
        """pt_render_device without device-memory allocation (frames of at most PT_NOALLOC_MAX_PIXELS pixels)."""
        self._check(self._lib.pt_render_device_noalloc(self._h, C.byref(frame), C.c_void_p(device_ptr),
                                                       C.c_void_p(stream) if stream else None))

    def submit(self, frame: PtFrame, slot: int, out: np.ndarray):
        """pt_render_submit: enqueue the frame on `slot`; `out` (a C-contiguous (height, width, 4) array of the frame's output
        type, best from host_array()) holds the image once wait(slot) has returned. The caller keeps `out` alive until then."""
        dt = np.float16 if frame.flags & PT_FRAME_OUTPUT_F16 else np.float32
        if out.dtype != dt or out.shape != (frame.height, frame.width, 4) or not out.flags.c_contiguous:
            raise ValueError("submit: `out` must be a C-contiguous (height, width, 4) array of the frame's output type")
        self._check(self._lib.pt_render_submit(self._h, C.byref(frame), slot, _ptr(out)))
        self.size = (frame.height, frame.width)

    def submit_device(self, frame: PtFrame, slot: int, device_ptr: int):
        """pt_render_submit_device: the image stays in device memory at `device_ptr`, valid after wait(slot)."""
        self._check(self._lib.pt_render_submit_device(self._h, C.byref(frame), slot, C.c_void_p(device_ptr)))
        self.size = (frame.height, frame.width)

    def wait(self, slot: int):
        """pt_render_wait: blocks until the frame submitted to `slot` is complete; stats()/read_aux() then describe it."""
        self._check(self._lib.pt_render_wait(self._h, slot))

    def stream_wait(self, slot: int, stream: int = 0):
        """pt_render_stream_wait: work enqueued on the HIP stream `stream` from now on starts when the frame submitted to
        `slot` is complete; the host does not block (0: the context's own stream)."""
        self._check(self._lib.pt_render_stream_wait(self._h, slot, C.c_void_p(stream) if stream else None))

    def slot_wait_for(self, slot: int, stream: int = 0):
        """pt_render_slot_wait_for: what is submitted to `slot` from now on starts when the work enqueued on `stream` so far
        is complete."""
        self._check(self._lib.pt_render_slot_wait_for(self._h, slot, C.c_void_p(stream) if stream else None))

    def slot_state(self, slot: int) -> np.ndarray:
        """pt_debug_slot_state: 20 words on where the slot's frames stand; waits for nothing."""
        out = np.zeros(20, dtype=np.uint32)
        self._check(self._lib.pt_debug_slot_state(self._h, slot, _ptr(out), out.size))
        return out

    def query(self, rays: np.ndarray) -> np.ndarray:
        rays = np.ascontiguousarray(rays, dtype=RAY_DT)
        hits = np.zeros(rays.size, dtype=HIT_DT)
        self._check(self._lib.pt_query(self._h, _ptr(rays), rays.size, _ptr(hits)))
        return hits

    def read_aux(self, kind: int) -> np.ndarray:
        if self.size is None:
            raise PtError(PT_ERR_NOT_READY, "read_aux before render")
        dt = np.float32 if kind == PT_AUX_PRIMARY_T else np.uint32
        out = np.empty(self.size, dtype=dt)
        self._check(self._lib.pt_read_aux(self._h, kind, _ptr(out), out.nbytes))
        return out

    def set_stats_detail(self, on: bool):
        self._check(self._lib.pt_set_stats_detail(self._h, 1 if on else 0))

    def stats(self) -> dict:
        st = PtStats()
        self._check(self._lib.pt_get_stats(self._h, C.byref(st)))
        return st.as_dict()

    def wave_log(self) -> np.ndarray:
        """Per-wave records (n, 16) u32 of the last frame rendered with detailed statistics."""
        n = C.c_uint32(0)
        self._check(self._lib.pt_debug_wave_log(self._h, None, 0, C.byref(n)))
        out = np.zeros((n.value, 16), dtype=np.uint32)
        if n.value:
            self._check(self._lib.pt_debug_wave_log(self._h, _ptr(out), out.nbytes, C.byref(n)))
        return out

    def check_accel(self) -> dict:
        e, t, b = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        self._check(self._lib.pt_debug_check_accel(self._h, C.byref(e), C.byref(t), C.byref(b)))
        return dict(errors=e.value, tlas_depth=t.value, max_blas_depth=b.value)

    def device_info(self) -> str:
        buf = C.create_string_buffer(512)
        self._check(self._lib.pt_device_info(self._h, buf, 512))
        return buf.value.decode()

"""ctypes loader for oracle/libpt_oracle.so — the CPU checker.

Only tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from path_tracing_amd.abi import HIT_DT, RAY_DT, PtFrame

ROOT = Path(__file__).resolve().parent.parent
ORACLE_DIR = ROOT / "oracle"
ORACLE_SO = ORACLE_DIR / "libpt_oracle.so"
# the same restatement with normalize() as v * (1 / sqrt(dot(v, v))), correctly rounded (-DPTO_IEEE_NORMALIZE)
ORACLE_IEEE_SO = ORACLE_DIR / "libpt_oracle_ieee.so"


class PtoScene(C.Structure):
    _fields_ = [("objects", C.c_void_p), ("n_objects", C.c_uint32),
                ("meshes", C.c_void_p), ("n_meshes", C.c_uint32),
                ("indices", C.c_void_p), ("n_indices", C.c_uint32),
                ("vertices", C.c_void_p), ("n_vertices", C.c_uint32),
                ("materials", C.c_void_p), ("n_materials", C.c_uint32),
                ("textures", C.c_void_p), ("n_textures", C.c_uint32),
                ("texture_data", C.c_void_p), ("n_texture_data", C.c_uint32)]


class PtoAux(C.Structure):
    _fields_ = [("primary_object", C.c_void_p), ("primary_flat_id", C.c_void_p),
                ("primary_t", C.c_void_p), ("ray_count", C.c_void_p)]


# variant -> library: "recipe" is the evaluation the kernels implement; the others are further legal evaluations of the same
# WGSL (see the build flags in oracle/pt_oracle.c), kept only to measure distances
VARIANTS = {"recipe": ORACLE_SO, "ieee": ORACLE_IEEE_SO, "nofma": ORACLE_DIR / "libpt_oracle_nofma.so",
            "nofma_libm": ORACLE_DIR / "libpt_oracle_nofma_libm.so"}
_libs = {}


def build_oracle():
    subprocess.run(["make", "-C", str(ORACLE_DIR), "all"], check=True, capture_output=True)


def load_oracle(variant="recipe"):
    """variant "recipe": the evaluation the kernels implement; "ieee": normalize() with sqrt and division; "nofma": that and
    no fused multiply-add anywhere; "nofma_libm": that and the C library's sin/cos."""
    if variant not in _libs:
        so = VARIANTS[variant]
        if not so.exists():
            build_oracle()
        _libs[variant] = _declare(C.CDLL(str(so)))
    return _libs[variant]


def _declare(lib):
    lib.pto_render.argtypes = [C.POINTER(PtoScene), C.POINTER(PtFrame), C.c_uint32, C.c_uint32, C.c_uint32,
                               C.c_uint32, C.c_void_p, C.POINTER(PtoAux), C.c_int]
    lib.pto_render.restype = C.c_int
    lib.pto_query.argtypes = [C.POINTER(PtoScene), C.c_void_p, C.c_uint32, C.c_void_p, C.c_int]
    lib.pto_query.restype = C.c_int
    lib.pto_validate.argtypes = [C.POINTER(PtoScene)]
    lib.pto_validate.restype = C.c_int
    lib.pto_pcg3d.argtypes = [C.c_void_p]
    lib.pto_rand.argtypes = [C.c_void_p, C.c_void_p]
    lib.pto_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.pto_rsqrt.argtypes = [C.c_float]
    lib.pto_rsqrt.restype = C.c_float
    lib.pto_num_threads.restype = C.c_int
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a.size else None


class OracleScene:
    """Keeps the numpy arrays alive next to the PtoScene that points into them."""

    def __init__(self, scene):
        self.arrays = [np.ascontiguousarray(a) for a in (scene.objects, scene.meshes, scene.indices, scene.vertices,
                                                         scene.materials, scene.textures, scene.texture_data)]
        o, m, i, v, ma, t, td = self.arrays
        self.c = PtoScene(_p(o), o.size, _p(m), m.size, _p(i), i.size, _p(v), v.size, _p(ma), ma.size,
                          _p(t), t.size, _p(td), td.size)


def oracle_render(scene, frame: PtFrame, rect=None, threads=0, variant="recipe"):
    """Returns dict(image (h,w,4) f32, primary_object, primary_flat_id, primary_t, ray_count)."""
    lib = load_oracle(variant)
    os_ = OracleScene(scene)
    x0, y0, x1, y1 = rect if rect is not None else (0, 0, frame.width, frame.height)
    h, w = y1 - y0, x1 - x0
    img = np.zeros((h, w, 4), dtype=np.float32)
    po = np.zeros((h, w), dtype=np.uint32)
    pf = np.zeros((h, w), dtype=np.uint32)
    pt = np.zeros((h, w), dtype=np.float32)
    rc = np.zeros((h, w), dtype=np.uint32)
    aux = PtoAux(_p(po), _p(pf), _p(pt), _p(rc))
    r = lib.pto_render(C.byref(os_.c), C.byref(frame), x0, y0, x1, y1, _p(img), C.byref(aux), threads)
    if r != 0:
        raise RuntimeError(f"pto_render returned {r}")
    return dict(image=img, primary_object=po, primary_flat_id=pf, primary_t=pt, ray_count=rc)


def oracle_query(scene, rays: np.ndarray, threads=0) -> np.ndarray:
    lib = load_oracle()
    os_ = OracleScene(scene)
    rays = np.ascontiguousarray(rays, dtype=RAY_DT)
    hits = np.zeros(rays.size, dtype=HIT_DT)
    r = lib.pto_query(C.byref(os_.c), _p(rays), rays.size, _p(hits), threads)
    if r != 0:
        raise RuntimeError(f"pto_query returned {r}")
    return hits


def oracle_pcg3d(state):
    lib = load_oracle()
    s = np.array(state, dtype=np.uint32)
    lib.pto_pcg3d(_p(s))
    return s


def oracle_rand(state):
    lib = load_oracle()
    s = np.array(state, dtype=np.uint32)
    out = np.zeros(3, dtype=np.float32)
    lib.pto_rand(_p(s), _p(out))
    return s, out


def oracle_sincos(x):
    lib = load_oracle()
    s, c = C.c_float(), C.c_float()
    lib.pto_sincos(C.c_float(float(np.float32(x))), C.byref(s), C.byref(c))
    return np.float32(s.value), np.float32(c.value)

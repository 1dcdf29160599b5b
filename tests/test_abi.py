"""CPU tests of the boundary: the header is valid C and C++, the library loads and exports every
declared symbol, the ctypes/numpy mirrors have the header's layouts, and the scene builders produce
arrays with the reference's strides. No compute calls (there is no GPU here)."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

from path_tracing_amd import abi, scenes

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "pt_amd.h"


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()  # incremental: recompiles only when a source is newer than the library
    return abi.load_library()


def declared_functions():
    text = re.sub(r"/\*.*?\*/", "", HEADER.read_text(), flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z_]+)\s*\(", text)))


def test_header_declares_what_python_binds():
    assert declared_functions() == sorted(abi.EXPORTS)


def test_library_exports_every_declared_symbol(lib):
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} is declared in include/pt_amd.h but not exported"


@pytest.mark.parametrize("compiler,lang", [("gcc", "c"), ("g++", "c++")])
def test_header_compiles_standalone(compiler, lang, tmp_path):
    src = tmp_path / ("t." + ("c" if lang == "c" else "cpp"))
    src.write_text('#include "pt_amd.h"\n'
                   "int main(void){ PtFrame f; PtConfig c; PtStats s; (void)f; (void)c; (void)s;\n"
                   f"  return (sizeof(PtFrame)=={C.sizeof(abi.PtFrame)} && sizeof(PtConfig)=={C.sizeof(abi.PtConfig)} && "
                   f"sizeof(PtStats)=={C.sizeof(abi.PtStats)} && sizeof(PtObject)==144 && sizeof(PtMesh)==48 && sizeof(PtVertex)==48 && "
                   "sizeof(PtMaterial)==64 && sizeof(PtTexture)==16 && sizeof(PtSettings)==32 && sizeof(PtRay)==24 && sizeof(PtHit)==48) ? 0 : 1; }\n")
    exe = tmp_path / "t"
    subprocess.run([compiler, "-Wall", "-Werror", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    assert subprocess.run([str(exe)]).returncode == 0


def test_numpy_layouts_match_reference_offsets():
    # reference src/data.rs:23-73 under encase/std430
    assert abi.OBJECT_DT.fields["world_to_local"][1] == 64 and abi.OBJECT_DT.fields["mat"][1] == 128 and abi.OBJECT_DT.fields["mesh"][1] == 132
    assert abi.MESH_DT.fields["aabb_max"][1] == 16 and abi.MESH_DT.fields["ihead"][1] == 28 and abi.MESH_DT.fields["tri_count"][1] == 36
    assert abi.VERTEX_DT.fields["normal"][1] == 16 and abi.VERTEX_DT.fields["uv"][1] == 32
    m = abi.MATERIAL_DT.fields
    assert [m[k][1] for k in ("albedo_texture", "emissive", "emissive_texture", "roughness", "metallic",
                               "metallic_roughness_texture", "reflectance", "normal_map_texture")] == [12, 16, 28, 32, 36, 40, 44, 48]


def test_a_collective_library_that_cannot_be_opened_is_reported():
    """PT_AMD_RCCL_LIBRARY names the only file tried; a failure comes back as PT_ERR_UNSUPPORTED with the loader's message
    (one process per case: the library is opened once per process)."""
    import os, sys
    code = ("from path_tracing_amd import abi\n"
            "try:\n    abi.rccl_version()\n    print('opened')\n"
            "except abi.PtError as e:\n    print('error', e.code, str(e))\n")
    env = dict(os.environ, PT_AMD_RCCL_LIBRARY="/nonexistent/librccl-missing.so")
    r = subprocess.run([sys.executable, "-c", code], cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = r.stdout.strip()
    assert out.startswith("error"), out
    assert str(abi.PT_ERR_UNSUPPORTED) in out and "could not be opened" in out and "librccl-missing" in out, out


def test_pt_create_error_paths_without_device(lib):
    h = C.c_void_p()
    assert lib.pt_create(None, C.byref(h)) == abi.PT_ERR_INVALID_ARGUMENT
    cfg = abi.PtConfig(abi_version=999, device=0)
    assert lib.pt_create(C.byref(cfg), C.byref(h)) == abi.PT_ERR_INVALID_ARGUMENT
    assert b"abi_version" in lib.pt_last_error(None)
    assert lib.pt_destroy(None) == abi.PT_OK
    assert lib.pt_render(None, None, None) == abi.PT_ERR_INVALID_ARGUMENT


@pytest.mark.parametrize("name,n_obj,n_tri", [("cornell", 7, 14), ("boxes", 6, 62), ("scene_glb_textured", 3, 982), ("texture_cube", 1, 12)])
def test_scene_builders(name, n_obj, n_tri):
    s = scenes.CONFIGS[name]()
    assert s.objects.size == n_obj and s.n_triangles == n_tri
    assert s.objects.dtype == abi.OBJECT_DT and s.vertices.dtype == abi.VERTEX_DT
    # world_to_local is the inverse of local_to_world
    for o in s.objects:
        l2w = o["local_to_world"].reshape(4, 4).T.astype(np.float64)
        w2l = o["world_to_local"].reshape(4, 4).T.astype(np.float64)
        assert np.allclose(l2w @ w2l, np.eye(4), atol=1e-5)
    # every triangle faces along its vertex normals (counter-clockwise seen from outside)
    for me in s.meshes:
        idx = s.indices[me["ihead"]: me["ihead"] + 3 * me["tri_count"]].reshape(-1, 3) + me["vhead"]
        p = s.vertices["position"][idx]
        n = np.cross(p[:, 1] - p[:, 0], p[:, 2] - p[:, 0])
        vn = s.vertices["normal"][idx].sum(axis=1)
        assert np.all((n * vn).sum(axis=1) > 0)


def test_instanced_boxes_full_size_counts():
    s = scenes.instanced_boxes(n_objects=83334, width=64, height=64, samples=1)
    assert s.objects.size == 83334 and s.n_triangles == 1000008 and s.materials.size == 65


def test_example_png_flattening():
    texels, w, h = scenes.load_example_png_f32()
    assert (w, h) == (16, 16) and texels.size == 1024
    px = texels.reshape(16, 16, 4)
    assert np.all(px[0, 0] == np.float32([1, 0, 0, 1])) and np.all(px[0, 15] == np.float32([0, 1, 0, 1]))
    assert np.all(px[15, 0] == np.float32([0, 0, 1, 1]))
    assert set(np.unique(texels)) == {np.float32(0.0), np.float32(1.0)}

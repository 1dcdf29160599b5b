"""Texture ingestion (reference src/data.rs:99-160): the Python mirror against hand-worked values of each format
branch, against the committed example image, and against the C++ mirror bit for bit. CPU only."""
import struct
import subprocess
from pathlib import Path

import numpy as np
import pytest

from path_tracing_amd import scenes
from path_tracing_amd.textures import DIV_255, FORMATS, TextureData

ROOT = Path(__file__).resolve().parent.parent
SELFTEST = ROOT / "path_tracing_amd" / "host" / "texture_selftest"


def f32(x):
    return np.float32(x)


def half(bits):
    return struct.unpack("<e", struct.pack("<H", bits))[0]


def test_div_255_is_the_f32_constant():
    assert DIV_255.dtype == np.float32 and DIV_255 == f32(1.0) / f32(255.0)
    assert float(DIV_255) != 1.0 / 255.0          # rounded to f32, as `const DIV_255: f32` is


def test_rgba8_r8_rg8_scale_every_byte():
    td = TextureData()
    t = td.append_texture(bytes([0, 1, 127, 255, 255, 0, 0, 255]), 2, 1, "Rgba8UnormSrgb")
    assert (t["width"], t["height"], t["offset"], t["format"]) == (2, 1, 0, 4)
    want = np.array([0, 1, 127, 255, 255, 0, 0, 255], dtype=np.float32) * DIV_255
    assert np.array_equal(td.data.view(np.uint32), want.view(np.uint32))
    assert td.data[3] != 1.0 or f32(255.0) * DIV_255 == f32(1.0)     # 255 * (1/255) in f32, whatever that rounds to
    t2 = td.append_texture(bytes([10, 20, 30]), 3, 1, "R8Unorm")
    assert (t2["offset"], t2["format"]) == (8, 1)                      # offsets count f32 elements of the pool
    t3 = td.append_texture(bytes([10, 20, 30, 40]), 2, 1, "Rg8Unorm")
    assert (t3["offset"], t3["format"]) == (11, 2)
    assert td.data.size == 15 and td.texture_array().shape == (3,)


def test_rgba16float_is_also_divided_by_255():
    td = TextureData()
    one, half_, two = 0x3C00, 0x3800, 0x4000
    raw = struct.pack("<4H", one, half_, two, 0x0000)
    t = td.append_texture(raw, 1, 1, "Rgba16Float")
    assert t["format"] == 4
    want = np.array([1.0, 0.5, 2.0, 0.0], dtype=np.float32) * DIV_255
    assert np.array_equal(td.data.view(np.uint32), want.view(np.uint32))


def test_rgb9e5_follows_the_reference_bit_arithmetic():
    td = TextureData()
    # d = [0x80, 0, 0, 0]: e = 0; r = [0 | (0x80 >> 7), (0x80 << 1) as u8] = [1, 0] -> half 0x0001; g = b = 0
    # d = [0, 0, 0, 0x07]: e = (0x07 << 2) & 0x1f = 0x1c; r = g = b = [0x1c, 0] -> half 0x001c
    # d = [0x41, 0x81, 0xC3, 0xFF]: e = (0xfc & 0x1f) = 0x1c; r = [0x1c, 0x82]; g = [0x1c, 0x04]; b = [0x1c, 0x18]
    raw = bytes([0x80, 0, 0, 0, 0, 0, 0, 0x07, 0x41, 0x81, 0xC3, 0xFF])
    t = td.append_texture(raw, 3, 1, "Rgb9e5Ufloat")
    assert t["format"] == 3 and td.data.size == 9
    want = [half(0x0001), 0.0, 0.0, half(0x001C), half(0x001C), half(0x001C), half(0x821C), half(0x041C), half(0x181C)]
    assert np.array_equal(td.data.view(np.uint32), np.array(want, dtype=np.float32).view(np.uint32))


def test_unsupported_format_raises_like_the_reference_panics():
    with pytest.raises(ValueError, match="is not supported"):
        TextureData().append_texture(bytes(16), 2, 2, "Bgra8Unorm")


def test_example_png_matches_the_scene_builder():
    rgba = np.load(scenes.ASSET_DIR / "example_png.npz")["rgba8"]
    td = TextureData()
    t = td.append_texture(rgba.tobytes(), rgba.shape[1], rgba.shape[0], "Rgba8UnormSrgb")
    flat, w, h = scenes.load_example_png_f32()
    assert (t["width"], t["height"]) == (w, h) and np.array_equal(td.data.view(np.uint32), flat.view(np.uint32))
    scene = scenes.texture_cube(32, 32, 1, 2)
    assert np.array_equal(scene.texture_data.view(np.uint32), flat.view(np.uint32))


@pytest.mark.parametrize("fmt", FORMATS)
def test_cpp_mirror_agrees_bit_for_bit(fmt, tmp_path):
    if not SELFTEST.exists():
        pytest.skip("texture_selftest not built (run __graft_entry__.build())")
    g = np.random.Generator(np.random.PCG64(99))
    raw = g.integers(0, 256, 4 * 64, dtype=np.uint8)
    if fmt == "Rgba16Float":   # keep the halves finite: NaN payloads are not part of the comparison
        h = raw.view("<u2").copy()
        h[(h & 0x7C00) == 0x7C00] &= 0x83FF
        raw = h.view(np.uint8)
    (tmp_path / "in.bin").write_bytes(raw.tobytes())
    r = subprocess.run([str(SELFTEST), fmt, "8", "8", str(tmp_path / "in.bin"), str(tmp_path / "out.f32")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    td = TextureData()
    t = td.append_texture(raw.tobytes(), 8, 8, fmt)
    assert r.stdout.split() == ["8", "8", "3", str(int(t["format"]))]     # the C++ pool already held 3 floats
    got = np.fromfile(tmp_path / "out.f32", dtype=np.float32)
    assert np.array_equal(got.view(np.uint32), td.data.view(np.uint32))


def test_cpp_mirror_rejects_unknown_format(tmp_path):
    if not SELFTEST.exists():
        pytest.skip("texture_selftest not built")
    (tmp_path / "in.bin").write_bytes(bytes(16))
    r = subprocess.run([str(SELFTEST), "Bgra8Unorm", "2", "2", str(tmp_path / "in.bin"), str(tmp_path / "out.f32")], capture_output=True, text=True)
    assert r.returncode == 1 and "not supported" in r.stdout


def test_plugin_mirror_flattens_images_in_order():
    from path_tracing_amd import abi
    from path_tracing_amd.plugin import RayTraceMeta
    meta = RayTraceMeta()
    mats = np.zeros(1, abi.MATERIAL_DT)
    meta.set_images(mats, [(bytes([255, 0, 0, 255]), 1, 1, "Rgba8UnormSrgb"), (bytes([128]), 1, 1, "R8Unorm")])
    assert meta.textures["offset"].tolist() == [0, 4] and meta.textures["format"].tolist() == [4, 1]
    assert meta.texture_data.size == 5 and "materials" in meta._dirty

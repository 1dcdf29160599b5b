#!/usr/bin/env python3
"""Decode the reference's two asset files into flat arrays.

Run in the build container (the reference tree is not present on the GPU box):

    python tools/decode_assets.py /root/reference/assets path_tracing_amd/assets

Writes
  scene_glb.npz   per glTF mesh primitive: positions (n,3) f32, normals (n,3) f32,
                  uvs (n,2) f32, indices (m,) u32, plus node transforms and the
                  material factors of assets/scene.glb
  example_png.npz rgba8 (16,16,4) u8 of assets/example.png

Only the container formats are parsed here (GLB chunks + accessors, PNG through
PIL); no loader of the reference is involved.
"""
import json
import struct
import sys
from pathlib import Path

import numpy as np

COMPONENT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def read_glb(path: Path):
    data = path.read_bytes()
    magic, version, length = struct.unpack_from("<III", data, 0)
    assert magic == 0x46546C67 and version == 2 and length == len(data)
    off = 12
    chunks = []
    while off < length:
        clen, ctype = struct.unpack_from("<II", data, off)
        chunks.append((ctype, data[off + 8: off + 8 + clen]))
        off += 8 + clen
    gltf = json.loads(chunks[0][1].decode("utf-8"))
    blob = chunks[1][1]
    return gltf, blob


def accessor(gltf, blob, idx):
    acc = gltf["accessors"][idx]
    bv = gltf["bufferViews"][acc["bufferView"]]
    dt = np.dtype(COMPONENT[acc["componentType"]]).newbyteorder("<")
    n = NCOMP[acc["type"]]
    start = bv.get("byteOffset", 0) + acc.get("byteOffset", 0)
    stride = bv.get("byteStride", 0)
    count = acc["count"]
    if stride and stride != dt.itemsize * n:
        out = np.empty((count, n), dtype=dt)
        for i in range(count):
            out[i] = np.frombuffer(blob, dtype=dt, count=n, offset=start + i * stride)
    else:
        out = np.frombuffer(blob, dtype=dt, count=count * n, offset=start).reshape(count, n)
    return out


def main(src: Path, dst: Path):
    dst.mkdir(parents=True, exist_ok=True)
    gltf, blob = read_glb(src / "scene.glb")
    out = {}
    names = []
    for ni, node in enumerate(gltf["nodes"]):
        if "mesh" not in node:
            continue
        mesh = gltf["meshes"][node["mesh"]]
        assert len(mesh["primitives"]) == 1
        prim = mesh["primitives"][0]
        key = f"n{ni}"
        names.append(node.get("name", key))
        out[key + "_positions"] = accessor(gltf, blob, prim["attributes"]["POSITION"]).astype(np.float32)
        out[key + "_normals"] = accessor(gltf, blob, prim["attributes"]["NORMAL"]).astype(np.float32)
        out[key + "_uvs"] = accessor(gltf, blob, prim["attributes"]["TEXCOORD_0"]).astype(np.float32)
        out[key + "_indices"] = accessor(gltf, blob, prim["indices"]).astype(np.uint32).reshape(-1)
        out[key + "_translation"] = np.asarray(node.get("translation", [0, 0, 0]), dtype=np.float32)
        out[key + "_rotation"] = np.asarray(node.get("rotation", [0, 0, 0, 1]), dtype=np.float32)
        out[key + "_scale"] = np.asarray(node.get("scale", [1, 1, 1]), dtype=np.float32)
        mat = prim.get("material")
        if mat is not None:
            m = gltf["materials"][mat]
            pbr = m.get("pbrMetallicRoughness", {})
            strength = m.get("extensions", {}).get("KHR_materials_emissive_strength", {}).get("emissiveStrength", 1.0)
            out[key + "_base_color"] = np.asarray(pbr.get("baseColorFactor", [1, 1, 1, 1]), dtype=np.float32)
            out[key + "_metallic"] = np.float32(pbr.get("metallicFactor", 1.0))
            out[key + "_roughness"] = np.float32(pbr.get("roughnessFactor", 1.0))
            out[key + "_emissive"] = np.asarray(m.get("emissiveFactor", [0, 0, 0]), dtype=np.float32) * np.float32(strength)
    out["node_names"] = np.asarray(names)
    np.savez_compressed(dst / "scene_glb.npz", **out)
    for k, v in out.items():
        print(k, getattr(v, "shape", None), getattr(v, "dtype", None))

    from PIL import Image
    im = Image.open(src / "example.png").convert("RGBA")
    rgba = np.asarray(im, dtype=np.uint8)
    np.savez_compressed(dst / "example_png.npz", rgba8=rgba)
    print("example.png", rgba.shape, np.unique(rgba.reshape(-1, 4), axis=0))


if __name__ == "__main__":
    main(Path(sys.argv[1]), Path(sys.argv[2]))

"""Scene and camera builders for the workload configurations.

These stand in for the reference's example apps (examples/cornellbox.rs,
boxes.rs, texture.rs, scene.rs), whose Bevy scene set-up cannot run here. Every
mesh is written out as explicit vertex arrays; nothing depends on Bevy's mesh
generators. The output of each builder is a `Scene` holding exactly the arrays
the reference's extract systems would upload (src/extract.rs:102-147, :191-346).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

from .abi import (MATERIAL_DT, MESH_DT, OBJECT_DT, PT_U32_MAX, TEXTURE_DT, VERTEX_DT, PtFrame, make_frame)

ASSET_DIR = Path(__file__).resolve().parent / "assets"


# ---- small linear algebra (float64 internally, f32 at the boundary) ----------
def translation(x, y, z):
    m = np.eye(4)
    m[:3, 3] = (x, y, z)
    return m


def scaling(x, y, z):
    return np.diag([x, y, z, 1.0])


def rotation_x(a):
    c, s = np.cos(a), np.sin(a)
    m = np.eye(4)
    m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
    return m


def rotation_y(a):
    c, s = np.cos(a), np.sin(a)
    m = np.eye(4)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    return m


def rotation_z(a):
    c, s = np.cos(a), np.sin(a)
    m = np.eye(4)
    m[0, 0], m[0, 1], m[1, 0], m[1, 1] = c, -s, s, c
    return m


def quat_to_mat(q):
    x, y, z, w = [float(v) for v in q]
    m = np.eye(4)
    m[:3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                 [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                 [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]]
    return m


def colmajor(m) -> np.ndarray:
    """4x4 (row, col) matrix -> 16 floats in glam / WGSL column-major order."""
    return np.asarray(m, dtype=np.float64).T.reshape(16).astype(np.float32)


def look_at(eye, target, up=(0.0, 1.0, 0.0)):
    """Camera-to-world matrix of `Transform::from_translation(eye).looking_at(target, up)`:
    the camera looks down its -Z axis."""
    eye, target, up = (np.asarray(v, dtype=np.float64) for v in (eye, target, up))
    back = eye - target
    back /= np.linalg.norm(back)
    right = np.cross(up, back)
    right /= np.linalg.norm(right)
    up2 = np.cross(back, right)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, up2, back, eye
    return m


def view_from_clip(fov_y, aspect, near):
    """Inverse of an infinite reverse-Z right-handed perspective projection
    (clip = (f/aspect*x, f*y, near, -z)): view_from_clip*(dx,dy,1,1) = (dx*aspect/f, dy/f, -1, 1/near)."""
    f = 1.0 / np.tan(0.5 * fov_y)
    m = np.zeros((4, 4))
    m[0, 0] = aspect / f
    m[1, 1] = 1.0 / f
    m[2, 3] = -1.0
    m[3, 2] = 1.0 / near
    return m


@dataclass
class Camera:
    eye: tuple
    target: tuple
    fov_y: float = np.pi / 4.0
    near: float = 0.1

    def matrices(self, width, height):
        return colmajor(look_at(self.eye, self.target)), colmajor(view_from_clip(self.fov_y, width / height, self.near))


# ---- scene container --------------------------------------------------------
@dataclass
class Scene:
    name: str
    meshes: np.ndarray
    indices: np.ndarray
    vertices: np.ndarray
    materials: np.ndarray
    textures: np.ndarray
    texture_data: np.ndarray
    objects: np.ndarray
    camera: Camera
    sky_color: tuple = (0.0, 0.0, 0.0)
    width: int = 512
    height: int = 512
    samples: int = 2     # RT_SAMPLES default, reference examples/common.rs:18
    bounces: int = 10    # RT_BOUNCES default, reference examples/common.rs:19
    notes: dict = field(default_factory=dict)

    @property
    def n_triangles(self) -> int:
        return int(self.meshes["tri_count"][self.objects["mesh"]].sum()) if self.objects.size else 0

    def frame(self, width=None, height=None, samples=None, bounces=None, time=0.0, **kw) -> PtFrame:
        w = self.width if width is None else width
        h = self.height if height is None else height
        wfv, vfc = self.camera.matrices(w, h)
        return make_frame(wfv, vfc, w, h, time=time,
                          bounces=self.bounces if bounces is None else bounces,
                          samples=self.samples if samples is None else samples,
                          sky_color=self.sky_color, **kw)


class SceneBuilder:
    """Accumulates meshes / materials / objects the way the reference's extract
    systems concatenate them (src/extract.rs:113-129)."""

    def __init__(self):
        self.meshes, self.indices, self.vertices = [], [], []
        self.materials, self.textures, self.texture_data, self.objects = [], [], [], []
        self._n_idx = 0
        self._n_vtx = 0

    def add_mesh(self, positions, normals, uvs, indices) -> int:
        positions = np.asarray(positions, dtype=np.float32).reshape(-1, 3)
        normals = np.asarray(normals, dtype=np.float32).reshape(-1, 3)
        uvs = np.asarray(uvs, dtype=np.float32).reshape(-1, 2)
        indices = np.asarray(indices, dtype=np.uint32).reshape(-1)
        v = np.zeros(len(positions), dtype=VERTEX_DT)
        v["position"], v["normal"], v["uv"] = positions, normals, uvs
        m = np.zeros((), dtype=MESH_DT)
        m["aabb_min"] = positions.min(axis=0)
        m["aabb_max"] = positions.max(axis=0)
        m["ihead"], m["vhead"], m["tri_count"] = self._n_idx, self._n_vtx, len(indices) // 3
        self.meshes.append(m)
        self.indices.append(indices)
        self.vertices.append(v)
        self._n_idx += len(indices)
        self._n_vtx += len(v)
        return len(self.meshes) - 1

    def add_texture(self, pixels_f32, width, height, channels) -> int:
        t = np.zeros((), dtype=TEXTURE_DT)
        t["width"], t["height"], t["format"] = width, height, channels
        t["offset"] = sum(len(d) for d in self.texture_data)
        self.textures.append(t)
        self.texture_data.append(np.asarray(pixels_f32, dtype=np.float32).reshape(-1))
        return len(self.textures) - 1

    def add_material(self, albedo=(1, 1, 1), emissive=(0, 0, 0), roughness=0.5, metallic=0.0, reflectance=0.5,
                     albedo_texture=PT_U32_MAX, emissive_texture=PT_U32_MAX,
                     metallic_roughness_texture=PT_U32_MAX, normal_map_texture=PT_U32_MAX) -> int:
        """Defaults are Bevy's StandardMaterial defaults (perceptual_roughness 0.5, metallic 0, reflectance 0.5)."""
        m = np.zeros((), dtype=MATERIAL_DT)
        m["albedo"], m["emissive"] = albedo, emissive
        m["roughness"], m["metallic"], m["reflectance"] = roughness, metallic, reflectance
        m["albedo_texture"], m["emissive_texture"] = albedo_texture, emissive_texture
        m["metallic_roughness_texture"], m["normal_map_texture"] = metallic_roughness_texture, normal_map_texture
        self.materials.append(m)
        return len(self.materials) - 1

    def add_object(self, mesh: int, mat: int, local_to_world=None):
        l2w = np.eye(4) if local_to_world is None else np.asarray(local_to_world, dtype=np.float64)
        l2w32 = l2w.astype(np.float32).astype(np.float64)
        o = np.zeros((), dtype=OBJECT_DT)
        o["local_to_world"] = colmajor(l2w32)
        o["world_to_local"] = colmajor(np.linalg.inv(l2w32))
        o["mat"], o["mesh"] = mat, mesh
        self.objects.append(o)
        return len(self.objects) - 1

    def add_objects_bulk(self, mesh_ids, mat_ids, l2w_stack):
        """l2w_stack: (n,4,4) float64."""
        l2w32 = np.asarray(l2w_stack, dtype=np.float32).astype(np.float64)
        inv = np.linalg.inv(l2w32)
        o = np.zeros(len(l2w32), dtype=OBJECT_DT)
        o["local_to_world"] = np.transpose(l2w32, (0, 2, 1)).reshape(-1, 16).astype(np.float32)
        o["world_to_local"] = np.transpose(inv, (0, 2, 1)).reshape(-1, 16).astype(np.float32)
        o["mat"], o["mesh"] = mat_ids, mesh_ids
        self.objects.extend(list(o))

    def build(self, name, camera, **kw) -> Scene:
        def cat(lst, dt):
            if not lst:
                return np.zeros(0, dtype=dt)
            if isinstance(lst[0], np.ndarray) and lst[0].ndim >= 1:
                return np.concatenate(lst).astype(dt)
            return np.array(lst, dtype=dt)
        return Scene(name=name,
                     meshes=cat(self.meshes, MESH_DT), indices=cat(self.indices, np.uint32),
                     vertices=cat(self.vertices, VERTEX_DT), materials=cat(self.materials, MATERIAL_DT),
                     textures=cat(self.textures, TEXTURE_DT), texture_data=cat(self.texture_data, np.float32),
                     objects=cat(self.objects, OBJECT_DT), camera=camera, **kw)


# ---- primitive meshes (explicit arrays) --------------------------------------
_AXIS_FRAMES = {  # normal -> (s, t) with s x t = normal
    (0, 1, 0): ((0, 0, 1), (1, 0, 0)), (0, -1, 0): ((1, 0, 0), (0, 0, 1)),
    (0, 0, 1): ((1, 0, 0), (0, 1, 0)), (0, 0, -1): ((0, 1, 0), (1, 0, 0)),
    (1, 0, 0): ((0, 1, 0), (0, 0, 1)), (-1, 0, 0): ((0, 0, 1), (0, 1, 0)),
}


def plane_mesh(normal, half):
    """Two-triangle square of half-size `half` through the origin, facing `normal`
    (an axis direction): 4 vertices, counter-clockwise seen from the normal side."""
    n = tuple(int(v) for v in normal)
    s, t = (np.asarray(v, dtype=np.float64) for v in _AXIS_FRAMES[n])
    assert np.allclose(np.cross(s, t), n)
    corners = [(-1, -1), (1, -1), (1, 1), (-1, 1)]
    pos = [half * (a * s + b * t) for a, b in corners]
    uv = [((a + 1) / 2, (b + 1) / 2) for a, b in corners]
    return np.array(pos), np.tile(np.asarray(n, dtype=np.float64), (4, 1)), np.array(uv), np.array([0, 1, 2, 2, 3, 0])


def cube_mesh(size=1.0):
    """Axis-aligned cube centred at the origin: 24 vertices (4 per face, face normals, per-face uv 0..1),
    12 triangles, counter-clockwise seen from outside."""
    pos, nor, uv, idx = [], [], [], []
    h = size / 2
    for n, (s, t) in _AXIS_FRAMES.items():
        n, s, t = (np.asarray(v, dtype=np.float64) for v in (n, s, t))
        base = len(pos)
        for a, b in [(-1, -1), (1, -1), (1, 1), (-1, 1)]:
            pos.append(h * (n + a * s + b * t))
            nor.append(n)
            uv.append(((a + 1) / 2, (b + 1) / 2))
        idx += [base, base + 1, base + 2, base + 2, base + 3, base]
    return np.array(pos), np.array(nor), np.array(uv), np.array(idx)


def load_example_png_f32():
    """assets/example.png flattened the way TextureData::append_texture flattens an
    Rgba8UnormSrgb image: every byte * (1/255), no sRGB decode (reference src/data.rs:102-112)."""
    rgba = np.load(ASSET_DIR / "example_png.npz")["rgba8"]
    div255 = np.float32(1.0) / np.float32(255.0)
    return (rgba.astype(np.float32) * div255).reshape(-1), rgba.shape[1], rgba.shape[0]


# ---- the workload configurations ---------------------------------------------
def cornell_box(width=512, height=512, samples=16, bounces=10) -> Scene:
    """examples/cornellbox.rs:59-113: seven planes, each its own mesh; white/red/green walls with the
    default roughness 0.5; black emissive light; black sky; camera (3,3,3) -> origin."""
    b = SceneBuilder()
    white = b.add_material(albedo=(1, 1, 1))
    red = b.add_material(albedo=(1, 0, 0))
    green = b.add_material(albedo=(0, 1, 0))
    light = b.add_material(albedo=(0, 0, 0), emissive=(1, 1, 1))
    walls = [((0, 1, 0), 1.1, white, (0, -1, 0)), ((0, -1, 0), 1.1, white, (0, 1, 0)),
             ((0, 0, 1), 1.1, white, (0, 0, -1)), ((0, 0, -1), 1.1, white, (0, 0, 1)),
             ((1, 0, 0), 1.1, red, (-1, 0, 0)), ((-1, 0, 0), 1.1, green, (1, 0, 0)),
             ((0, -1, 0), 0.25, light, (0, 0.95, 0))]
    for normal, half, mat, pos in walls:
        m = b.add_mesh(*plane_mesh(normal, half))
        b.add_object(m, mat, translation(*pos))
    return b.build("cornell", Camera((3, 3, 3), (0, 0, 0)), sky_color=(0, 0, 0),
                   width=width, height=height, samples=samples, bounces=bounces)


def boxes(width=1920, height=1080, samples=64, bounces=10) -> Scene:
    """examples/boxes.rs:59-130: one unit cube mesh instanced five times + a ground plane."""
    b = SceneBuilder()
    cube = b.add_mesh(*cube_mesh(1.0))
    b.add_object(cube, b.add_material(albedo=(0, 0, 1), roughness=0.5), translation(0, 0, 1.5))
    b.add_object(cube, b.add_material(albedo=(1, 0, 0), roughness=1.0), translation(0, 0, -1.5))
    b.add_object(cube, b.add_material(albedo=(0, 0, 0), emissive=(2, 2, 2)),
                 translation(1.5, 0, 0) @ scaling(0.5, 0.5, 2.0))
    b.add_object(cube, b.add_material(albedo=(0, 0, 0), emissive=(2, 1.7, 0)),
                 translation(-1.5, 0, 0) @ scaling(0.5, 2.0, 0.5))
    # Quat::from_euler(EulerRot::XYZ, 45deg, 45deg, 0) = Rx(45) * Ry(45)
    rot = rotation_x(np.radians(45.0)) @ rotation_y(np.radians(45.0))
    b.add_object(cube, b.add_material(albedo=(0, 1, 0), roughness=0.0, metallic=0.1), rot @ scaling(0.5, 0.5, 0.5))
    plane = b.add_mesh(*plane_mesh((0, 1, 0), 5.0))
    b.add_object(plane, b.add_material(albedo=(0.4, 0.4, 0.4), roughness=0.1, metallic=1.0), translation(0, -2, 0))
    return b.build("boxes", Camera((3, 3, 3), (0, 0, 0)), sky_color=(0.1, 0.2, 0.4),
                   width=width, height=height, samples=samples, bounces=bounces)


def scene_glb_textured(width=1920, height=1080, samples=256, bounces=10) -> Scene:
    """assets/scene.glb (emissive plane + 968-triangle Suzanne, examples/scene.rs) with Suzanne's
    material given the example.png albedo texture, plus the textured unit cube of
    examples/texture.rs:59-67 placed beside it. Camera (0,0,7) looking -Z; grey sky."""
    g = np.load(ASSET_DIR / "scene_glb.npz")
    b = SceneBuilder()
    texels, tw, th = load_example_png_f32()
    tex = b.add_texture(texels, tw, th, 4)

    def node_matrix(key):
        return translation(*g[key + "_translation"]) @ quat_to_mat(g[key + "_rotation"]) @ scaling(*g[key + "_scale"])

    plane = b.add_mesh(g["n0_positions"], g["n0_normals"], g["n0_uvs"], g["n0_indices"])
    suz = b.add_mesh(g["n1_positions"], g["n1_normals"], g["n1_uvs"], g["n1_indices"])
    cube = b.add_mesh(*cube_mesh(1.0))
    plane_mat = b.add_material(albedo=tuple(g["n0_base_color"][:3]), emissive=tuple(g["n0_emissive"]),
                               roughness=float(g["n0_roughness"]), metallic=float(g["n0_metallic"]))
    # a glTF primitive without a material gets metallic 1 / roughness 1 / white
    suz_mat = b.add_material(albedo=(1, 1, 1), roughness=1.0, metallic=1.0, albedo_texture=tex)
    cube_mat = b.add_material(albedo=(1, 1, 1), albedo_texture=tex)
    b.add_object(plane, plane_mat, node_matrix("n0"))
    b.add_object(suz, suz_mat, node_matrix("n1"))
    b.add_object(cube, cube_mat, translation(2.4, 0.0, 0.0) @ rotation_y(np.radians(30.0)))
    return b.build("scene_glb_textured", Camera((0, 0, 7), (0, 0, 0)), sky_color=(0.5, 0.5, 0.5),
                   width=width, height=height, samples=samples, bounces=bounces)


def overlapping_meshes(width=256, height=144, samples=16, bounces=8) -> Scene:
    """Synthetic (not one of the reference's scenes): twelve objects (the most that still get the object list rather than a TLAS): bumpy sheets with a BLAS, thin strips,
    cubes without one, two of the sheets coincident so that equal distances occur; boxes overlap heavily."""
    rng = np.random.default_rng(5)
    b = SceneBuilder()
    n = 24
    x = np.linspace(-1.0, 1.0, n + 1)
    xx, zz = np.meshgrid(x, x, indexing="ij")
    yy = 0.2 * np.sin(5 * xx) * np.cos(4 * zz)
    pos = np.stack([xx, yy, zz], axis=-1).reshape(-1, 3)
    nrm = np.tile(np.array([0.0, 1.0, 0.0]), (len(pos), 1))
    uv = np.stack([(xx + 1) / 2, (zz + 1) / 2], axis=-1).reshape(-1, 2)
    idx = []
    for i in range(n):
        for j in range(n):
            a, bb, c, d = i * (n + 1) + j, (i + 1) * (n + 1) + j, (i + 1) * (n + 1) + j + 1, i * (n + 1) + j + 1
            idx += [a, d, bb, bb, d, c]
    sheet = b.add_mesh(pos, nrm, uv, np.array(idx))                      # 1,152 triangles
    strip = b.add_mesh(pos[: 2 * (n + 1)], nrm[: 2 * (n + 1)], uv[: 2 * (n + 1)], np.array(idx[: 6 * n]))   # 48 triangles
    cube = b.add_mesh(*cube_mesh(1.0))
    mats = [b.add_material(albedo=tuple(rng.uniform(0.3, 0.9, 3)), roughness=float(r)) for r in (0.0, 0.4, 1.0, 0.7)]
    light = b.add_material(albedo=(0, 0, 0), emissive=(6, 6, 6))
    S = lambda sx, sy, sz: np.diag([sx, sy, sz, 1.0])
    b.add_object(sheet, mats[0], S(3, 1, 3))
    b.add_object(sheet, mats[1], S(3, 1, 3))                                   # coincident with the first
    b.add_object(sheet, mats[2], translation(0, 0.6, 0) @ rotation_x(0.5) @ S(2, 1, 2))
    b.add_object(sheet, mats[3], translation(0.3, 1.0, -0.5) @ rotation_z(-0.7) @ S(1.5, 2, 1.5))
    for k in range(3):
        b.add_object(strip, mats[k], translation(0.4 * k - 0.4, 0.5 + 0.3 * k, 0.2 * k) @ rotation_y(0.9 * k) @ S(2, 1, 6))
    for k in range(4):
        b.add_object(cube, mats[k], translation(*rng.uniform(-1.2, 1.2, 3)) @ rotation_y(float(rng.uniform(0, 3))) @ S(0.5, 0.5 + 0.3 * k, 0.5))
    b.add_object(cube, light, translation(0.5, 2.6, 0.8) @ S(0.6, 0.2, 0.6))
    return b.build("overlap", Camera((1.5, 2.4, 4.5), (0, 0.5, 0)), sky_color=(0.25, 0.3, 0.4), width=width, height=height, samples=samples, bounces=bounces)


def instanced_boxes(n_objects=83334, width=1920, height=1080, samples=256, bounces=10, seed=0x5EED0001,
                    extent=40.0) -> Scene:
    """One 12-triangle cube mesh instanced `n_objects` times (83,334 -> 1,000,008 triangles) with seeded random
    translation in [-extent, extent]^3, uniform scale in [0.2, 0.6] and random rotation; 64 materials, every 50th
    object a black emissive (4,4,4) light; camera (60,45,60) -> origin."""
    b = SceneBuilder()
    cube = b.add_mesh(*cube_mesh(1.0))
    rng = np.random.Generator(np.random.PCG64(seed))
    mats = []
    for i in range(64):
        alb = rng.uniform(0.2, 0.9, 3)
        mats.append(b.add_material(albedo=tuple(alb), roughness=(0.0, 0.5, 1.0)[i % 3]))
    light = b.add_material(albedo=(0, 0, 0), emissive=(4, 4, 4))
    n = n_objects
    tr = rng.uniform(-extent, extent, (n, 3))
    sc = rng.uniform(0.2, 0.6, n)
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    x, y, z, w = q.T
    R = np.empty((n, 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - z * w); R[:, 0, 2] = 2 * (x * z + y * w)
    R[:, 1, 0] = 2 * (x * y + z * w); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - x * w)
    R[:, 2, 0] = 2 * (x * z - y * w); R[:, 2, 1] = 2 * (y * z + x * w); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    l2w = np.zeros((n, 4, 4))
    l2w[:, :3, :3] = R * sc[:, None, None]
    l2w[:, :3, 3] = tr
    l2w[:, 3, 3] = 1.0
    mat_ids = np.array([light if (i % 50) == 49 else mats[i % 64] for i in range(n)], dtype=np.uint32)
    b.add_objects_bulk(np.full(n, cube, dtype=np.uint32), mat_ids, l2w)
    cam_d = 1.5 * extent
    return b.build("instanced_boxes", Camera((cam_d, 0.75 * cam_d, cam_d), (0, 0, 0)), sky_color=(0.1, 0.2, 0.4),
                   width=width, height=height, samples=samples, bounces=bounces)


def cornell_4k(samples=4096) -> Scene:
    s = cornell_box(3840, 2160, samples, 10)
    s.name = "cornell_4k"
    return s


def texture_cube(width=512, height=512, samples=4, bounces=10) -> Scene:
    """examples/texture.rs:57-67 alone: a unit cube with the example.png albedo texture, grey sky."""
    b = SceneBuilder()
    texels, tw, th = load_example_png_f32()
    tex = b.add_texture(texels, tw, th, 4)
    cube = b.add_mesh(*cube_mesh(1.0))
    b.add_object(cube, b.add_material(albedo=(1, 1, 1), albedo_texture=tex), translation(0, 0, 0))
    return b.build("texture_cube", Camera((3, 3, 3), (0, 0, 0)), sky_color=(0.5, 0.5, 0.5),
                   width=width, height=height, samples=samples, bounces=bounces)


CONFIGS = {
    "cornell": cornell_box,
    "boxes": boxes,
    "scene_glb_textured": scene_glb_textured,
    "instanced_boxes": instanced_boxes,
    "cornell_4k": cornell_4k,
    "texture_cube": texture_cube,
}

#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the CPU restatement (oracle/).

The reference ships no tests or fixtures and cannot be built offline, so these vectors pin
the restatement itself (regression) and give the GPU tests data that does not need the
checker at run time. Re-run after any deliberate change of the arithmetic rules:

    python tools/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from path_tracing_amd import abi, scenes  # noqa: E402
from tests.oracle_lib import oracle_pcg3d, oracle_query, oracle_rand, oracle_render, oracle_sincos  # noqa: E402

OUT = ROOT / "tests" / "golden"


def golden_scenes():
    """name -> (scene, crop rect) ; small frames so the fixtures stay a few hundred KB."""
    return {
        "cornell": (scenes.cornell_box(64, 64, 8, 10), None),
        "boxes": (scenes.boxes(96, 54, 8, 10), None),
        "scene_glb_textured": (scenes.scene_glb_textured(96, 54, 4, 10), None),
        "texture_cube": (scenes.texture_cube(64, 64, 4, 10), None),
        "instanced_boxes": (scenes.instanced_boxes(n_objects=400, width=64, height=48, samples=2, extent=6.0), None),
    }


def rng_seeds():
    seeds = [(0, 0, 0), (1, 2, 3), (0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF, 0, 0xFFFFFFFF), (0x80000000, 1, 0x80000001)]
    for x, y in [(0, 0), (511, 511), (1919, 1079), (3839, 2159), (17, 4), (255, 256), (1, 0), (0, 1)]:
        seeds.append((x, y, x ^ y))
    g = np.random.Generator(np.random.PCG64(1234))
    while len(seeds) < 64:
        a, b = (int(v) for v in g.integers(0, 2 ** 32, 2, dtype=np.uint64))
        seeds.append((a, b, a ^ b))
    return np.array(seeds, dtype=np.uint32)


def camera_rays(scene, n, seed):
    """n rays from the scene camera through random pixel centres (the primary rays of those pixels), plus
    secondary-like rays started on the hit points, so both long and short distances are covered."""
    g = np.random.Generator(np.random.PCG64(seed))
    eye = np.asarray(scene.camera.eye, dtype=np.float64)
    rays = np.zeros(n, dtype=abi.RAY_DT)
    tgt = g.uniform(-1.5, 1.5, (n, 3))
    d = tgt - eye
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["pos"] = eye.astype(np.float32)
    rays["dir"] = d.astype(np.float32)
    # second half: origins inside the scene, random directions (not normalised on purpose for a few)
    h = n // 2
    rays["pos"][h:] = g.uniform(-0.9, 0.9, (n - h, 3)).astype(np.float32)
    d2 = g.normal(size=(n - h, 3))
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    d2[: (n - h) // 8] *= g.uniform(0.3, 1.0, ((n - h) // 8, 1))
    rays["dir"][h:] = d2.astype(np.float32)
    return rays


def edge_rays_cornell():
    """Hand-picked rays for the Cornell box (planes at +-1, half 1.1; light at y = 0.95, half 0.25)."""
    r = []
    # shared diagonal of the floor's two triangles (x == z): exact tie between triangle 0 and 1
    for x in (0.3, -0.5, 0.0, 0.75):
        r.append(((x, 0.5, x), (0, -1, 0)))
    # exactly on an outer edge / corner of the floor quad
    r.append(((1.1, 0.5, 0.2), (0, -1, 0)))
    r.append(((1.1, 0.5, 1.1), (0, -1, 0)))
    r.append(((-1.1, 0.5, -1.1), (0, -1, 0)))
    # back faces are culled: from below the floor looking up passes through it and hits the light / ceiling
    r.append(((0.0, -3.0, 0.0), (0, 1, 0)))
    r.append(((0.6, -3.0, 0.6), (0, 1, 0)))
    # origin exactly ON the flat AABB of a plane (0 * inf in the gate)
    r.append(((0.2, -1.0, 0.2), (0, 1, 0)))
    r.append(((0.2, -1.0, 0.2), (1, 0, 0)))
    # starts closer to the floor than T_MIN: 0 < dst_near < 1e-4 is rejected by the gate
    r.append(((0.2, -1.0 + 5e-5, 0.2), (0, -1, 0)))
    r.append(((0.2, -1.0 + 2e-4, 0.2), (0, -1, 0)))
    # edge-on to a wall (direction inside the plane of the left wall)
    r.append(((-1.0, 0.0, 0.0), (0, 1, 0)))
    r.append(((-1.0, 0.0, 0.0), (0, 0, 1)))
    # axis-parallel directions with zero components, and a grazing one
    r.append(((0.0, 0.0, 0.0), (1, 0, 0)))
    r.append(((0.0, 0.0, 0.0), (0, 0, -1)))
    r.append(((0.0, 0.0, 0.0), (1, 1e-6, 0)))
    # beyond the 1000 max distance / exactly at it
    r.append(((0.0, 999.0, 0.0), (0, -1, 0)))
    r.append(((0.0, 1001.5, 0.0), (0, -1, 0)))
    # non-unit directions scale t
    r.append(((0.0, 0.5, 0.0), (0, -0.25, 0)))
    r.append(((0.0, 0.5, 0.0), (0, -4.0, 0)))
    out = np.zeros(len(r), dtype=abi.RAY_DT)
    for i, (p, d) in enumerate(r):
        out["pos"][i], out["dir"][i] = p, d
    return out


def edge_rays_boxes():
    """examples/boxes.rs: cube edges/corners, shared edges between faces (ties across triangles of one object),
    inside-a-cube origins (AABB gate returns INFINITY), rays through the rotated cube."""
    r = []
    # blue cube at (0,0,1.5), unit size: towards its +Z face centre, edges and corner
    r.append(((0.0, 0.0, 5.0), (0, 0, -1)))
    r.append(((0.5, 0.0, 5.0), (0, 0, -1)))     # on the edge between +Z and +X faces
    r.append(((0.5, 0.5, 5.0), (0, 0, -1)))     # corner
    r.append(((0.25, 0.25, 5.0), (0, 0, -1)))   # on the face diagonal
    r.append(((0.0, 0.0, 1.5), (0, 0, 1)))      # origin inside the cube: only back faces ahead -> culled
    r.append(((0.0, 0.0, 1.5), (1, 0, 0)))
    # emissive stretched cubes
    r.append(((5.0, 0.0, 0.0), (-1, 0, 0)))
    r.append(((-5.0, 0.5, 0.1), (1, 0, 0)))
    # rotated green cube at the origin
    for d in [(-1, -1, -1), (-1, -0.9, -1.1), (-1, 0, 0), (0, -1, 0), (0, 0, -1)]:
        r.append(((3.0 * -d[0] if d[0] else 0.0, 3.0 * -d[1] if d[1] else 0.0, 3.0 * -d[2] if d[2] else 0.0), d))
    # ground plane y = -2 from above and from below (back face)
    r.append(((0.3, 3.0, 0.3), (0, -1, 0)))
    r.append(((0.3, -3.0, 0.3), (0, 1, 0)))
    r.append(((4.0, 0.0, 4.0), (0, -1, 0)))
    out = np.zeros(len(r), dtype=abi.RAY_DT)
    for i, (p, d) in enumerate(r):
        out["pos"][i], out["dir"][i] = p, d
    return out


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    # (i)/(ii) RNG known answers
    seeds = rng_seeds()
    st1 = np.array([oracle_pcg3d(s) for s in seeds], dtype=np.uint32)
    st2 = np.array([oracle_pcg3d(s) for s in st1], dtype=np.uint32)
    rnd = np.array([oracle_rand(s)[1] for s in seeds], dtype=np.float32)
    angles = np.concatenate([np.linspace(0, 2 * np.pi, 257), [0.0, np.pi / 4, np.pi / 2, np.pi, 3 * np.pi / 2, 6.2831855]]).astype(np.float32)
    sc = np.array([oracle_sincos(a) for a in angles], dtype=np.float32)
    np.savez_compressed(OUT / "rng_kat.npz", seeds=seeds, state_after_1=st1, state_after_2=st2,
                        rand_bits=rnd.view(np.uint32), sincos_angles_bits=angles.view(np.uint32), sincos_bits=sc.view(np.uint32))

    # (iii) rays -> hit records
    cornell = scenes.cornell_box(64, 64, 1, 1)
    bx = scenes.boxes(64, 64, 1, 1)
    glb = scenes.scene_glb_textured(64, 64, 1, 1)
    ray_sets = {
        "cornell": np.concatenate([edge_rays_cornell(), camera_rays(cornell, 400, 11)]),
        "boxes": np.concatenate([edge_rays_boxes(), camera_rays(bx, 400, 12)]),
        "scene_glb_textured": camera_rays(glb, 400, 13),
    }
    ray_scenes = {"cornell": cornell, "boxes": bx, "scene_glb_textured": glb}
    pack = {}
    for name, rays in ray_sets.items():
        hits = oracle_query(ray_scenes[name], rays)
        pack[name + "_rays"] = rays.view(np.float32).reshape(-1, 6)
        pack[name + "_hits"] = hits.view(np.uint32).reshape(-1, 12)
        n_hit = int((hits["object"] != abi.PT_U32_MAX).sum())
        print(f"rays {name}: {len(rays)} rays, {n_hit} hits")
    np.savez_compressed(OUT / "ray_hits.npz", **pack)

    # (iv)/(v) small frames: primary ids, t bits, ray counts and the f32 image
    for name, (scene, _) in golden_scenes().items():
        frame = scene.frame()
        r = oracle_render(scene, frame)
        np.savez_compressed(OUT / f"frame_{name}.npz", image_bits=r["image"].view(np.uint32),
                            primary_object=r["primary_object"], primary_flat_id=r["primary_flat_id"],
                            primary_t_bits=r["primary_t"].view(np.uint32), ray_count=r["ray_count"],
                            width=np.uint32(frame.width), height=np.uint32(frame.height),
                            samples=np.uint32(frame.settings.samples), bounces=np.uint32(frame.settings.bounces))
        print(f"frame {name}: {frame.width}x{frame.height} spp={frame.settings.samples} rays={int(r['ray_count'].sum())} mean={r['image'][..., :3].mean():.5f}")
    # a frame with time != 0 (seed multiplier 1.5) for the cornell box
    scene = scenes.cornell_box(64, 64, 4, 10)
    frame = scene.frame(time=0.5)
    r = oracle_render(scene, frame)
    np.savez_compressed(OUT / "frame_cornell_time0p5.npz", image_bits=r["image"].view(np.uint32), ray_count=r["ray_count"])


if __name__ == "__main__":
    main()

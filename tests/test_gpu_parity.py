"""GPU parity tests: the HIP path, called through the C ABI, against the committed golden vectors, against
the CPU restatement on the same seeded inputs, and - at full benchmark sizes - through properties that do not
need the checker (acceleration structure on/off bit-identity, partition union, determinism, ray accounting).

Bar: bit-exact for primary-hit object / flat triangle ids, hit distances, ray counts; the image is required to
agree within relative RMSE 1e-3 (the tolerance the goal states) and, because both sides use the same fixed f32
evaluation rules, is additionally expected to be bit-identical.
"""
import numpy as np
import pytest

from path_tracing_amd import abi, scenes
from tests.oracle_lib import oracle_query, oracle_render
from tools.make_golden import golden_scenes

pytestmark = pytest.mark.gpu

ACCELS = [abi.PT_ACCEL_BRUTE, abi.PT_ACCEL_AUTO, abi.PT_ACCEL_BVH]
RRMSE_TOL = 1e-3


def rrmse(a, b):
    a, b = a[..., :3].astype(np.float64), b[..., :3].astype(np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / max((b ** 2).sum(), 1e-30)))


def render_all(scene, frame, accel=abi.PT_ACCEL_AUTO, tie=abi.PT_TIE_REFERENCE):
    with abi.Context(accel=accel, tie_rule=tie) as ctx:
        ctx.upload_scene(scene)
        img = ctx.render(frame)
        out = dict(image=img, primary_object=ctx.read_aux(abi.PT_AUX_PRIMARY_OBJECT),
                   primary_flat_id=ctx.read_aux(abi.PT_AUX_PRIMARY_FLAT_ID), primary_t=ctx.read_aux(abi.PT_AUX_PRIMARY_T),
                   ray_count=ctx.read_aux(abi.PT_AUX_RAY_COUNT), stats=ctx.stats(), accel=ctx.check_accel())
    return out


def assert_same(got, want_img, want_obj, want_flat, want_t_bits, want_rays):
    assert np.array_equal(got["primary_object"], want_obj)
    assert np.array_equal(got["primary_flat_id"], want_flat)
    assert np.array_equal(got["primary_t"].view(np.uint32), want_t_bits)
    assert np.array_equal(got["ray_count"], want_rays)
    assert rrmse(got["image"], want_img) <= RRMSE_TOL
    assert np.array_equal(got["image"].view(np.uint32), want_img.view(np.uint32)), "image not bit-identical"
    assert got["accel"]["errors"] == 0


@pytest.mark.parametrize("accel", ACCELS)
@pytest.mark.parametrize("name", ["cornell", "boxes", "scene_glb_textured", "texture_cube", "instanced_boxes"])
def test_golden_frames(name, accel, golden_dir):
    scene, _ = golden_scenes()[name]
    g = np.load(golden_dir / f"frame_{name}.npz")
    got = render_all(scene, scene.frame(), accel)
    assert_same(got, g["image_bits"].view(np.float32), g["primary_object"], g["primary_flat_id"], g["primary_t_bits"], g["ray_count"])
    assert got["stats"]["rays_reference"] == int(g["ray_count"].astype(np.uint64).sum())


def test_golden_frame_nonzero_time(golden_dir):
    scene = scenes.cornell_box(64, 64, 4, 10)
    g = np.load(golden_dir / "frame_cornell_time0p5.npz")
    got = render_all(scene, scene.frame(time=0.5))
    assert np.array_equal(got["image"].view(np.uint32), g["image_bits"])
    assert np.array_equal(got["ray_count"], g["ray_count"])


@pytest.mark.parametrize("accel", ACCELS)
def test_golden_ray_hits(accel, golden_dir):
    g = np.load(golden_dir / "ray_hits.npz")
    scs = {"cornell": scenes.cornell_box(64, 64, 1, 1), "boxes": scenes.boxes(64, 64, 1, 1),
           "scene_glb_textured": scenes.scene_glb_textured(64, 64, 1, 1)}
    for name, scene in scs.items():
        rays = np.ascontiguousarray(g[name + "_rays"]).view(abi.RAY_DT).reshape(-1)
        with abi.Context(accel=accel) as ctx:
            ctx.upload_scene(scene)
            hits = ctx.query(rays)
        want = g[name + "_hits"]
        got = hits.view(np.uint32).reshape(-1, 12)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, f"{name}: {bad.size} rays differ, first {bad[:5]}"


@pytest.mark.parametrize("name,kw,time", [
    ("cornell", dict(width=128, height=128, samples=16), 0.0),
    ("cornell", dict(width=61, height=37, samples=3), 1.5),       # ragged size: partial 8x8 tiles
    ("boxes", dict(width=160, height=90, samples=16), 0.0),
    ("scene_glb_textured", dict(width=120, height=68, samples=4), 2.5),
])
def test_live_oracle_parity(name, kw, time):
    scene = scenes.CONFIGS[name](**kw)
    frame = scene.frame(time=time)
    ref = oracle_render(scene, frame)
    for accel in (abi.PT_ACCEL_AUTO, abi.PT_ACCEL_BVH):
        got = render_all(scene, frame, accel)
        assert_same(got, ref["image"], ref["primary_object"], ref["primary_flat_id"], ref["primary_t"].view(np.uint32), ref["ray_count"])


# (scene, size, variant of the restatement, bound on the whole image's relative RMSE)
DISTANCE_CASES = [
    ("cornell", dict(width=128, height=128, samples=16), "ieee", 1e-3),
    ("boxes", dict(width=160, height=90, samples=64), "ieee", 1e-3),
    ("boxes", dict(width=160, height=90, samples=64), "nofma", 1e-3),
    ("boxes", dict(width=160, height=90, samples=64), "nofma_libm", 1e-3),
    ("cornell", dict(width=96, height=96, samples=256), "ieee", 2e-2),
    # the Cornell box at the size and sample count its configuration states: of 262,144 pixels a dozen take one different
    # accept/reject decision and then differ at noise level (16 samples), which alone puts the image 5e-3 to 7e-3 away
    ("cornell", dict(width=512, height=512, samples=16), "ieee", 2e-2),
    ("cornell", dict(width=512, height=512, samples=16), "nofma", 2e-2),
    ("cornell", dict(width=512, height=512, samples=16), "nofma_libm", 2e-2),
]


@pytest.mark.parametrize("name,kw,variant,bound", DISTANCE_CASES)
def test_distance_to_other_legal_evaluations(name, kw, variant, bound, record_property):
    """The checker the kernels are bit-identical to fixes one evaluation of the WGSL: explicit fused multiply-adds, normalize()
    through a fixed inverse-square-root recipe, polynomial sin/cos. WGSL allows others. Three of them are built from the same
    restatement (oracle/Makefile): "ieee" (normalize with sqrt and division), "nofma" (that, and no fused multiply-add
    anywhere: the arithmetic of a rasteriser that does not contract) and "nofma_libm" (that, and the C library's sin/cos).
    The HIP image is compared with each: the primary hits must be the same object and triangle in every pixel, the pixels
    that traced the same paths must agree to rounding noise, and nearly all pixels must be such pixels. Where a handful of
    pixels take one different accept/reject decision somewhere along their segments, their RNG streams run apart and they
    differ at noise level: on frames with few samples per pixel that alone exceeds the goal's 1e-3 for the whole image, for
    ANY two implementations of the reference, so the bound asserted there is the measured distance with margin (stated
    per case above) and the 1e-3 is asserted where it holds."""
    scene = scenes.CONFIGS[name](**kw)
    frame = scene.frame()
    ref = oracle_render(scene, frame, variant=variant)
    got = render_all(scene, frame)
    assert np.array_equal(got["primary_object"], ref["primary_object"])
    assert np.array_equal(got["primary_flat_id"], ref["primary_flat_id"])
    # (the camera ray's direction is itself a normalize(): the evaluations differ in its last bit, and so may t)
    assert np.allclose(got["primary_t"], ref["primary_t"], rtol=1e-6, atol=0.0)
    d = rrmse(got["image"], ref["image"])
    same = got["ray_count"] == ref["ray_count"]
    record_property(f"rrmse_vs_{variant}", d)
    record_property("pixels_with_same_paths", float(same.mean()))
    print(f"{name} {kw} vs {variant}: rRMSE {d:.3e}, {same.mean():.6f} of the pixels trace the same paths, those agree to {rrmse(got['image'][same], ref['image'][same]):.2e}")
    assert d <= bound
    assert same.mean() >= 0.999
    assert rrmse(got["image"][same], ref["image"][same]) <= 1e-5


def materials_zoo():
    """Every texture format (1-4 channels and an unsupported count), emissive / normal-map / metallic-roughness
    textures, a non-square texture (row stride = height quirk), u = 1.0 reads one past the row."""
    b = scenes.SceneBuilder()
    g = np.random.Generator(np.random.PCG64(5))
    t1 = b.add_texture(g.uniform(0.2, 1, 8 * 8 * 1), 8, 8, 1)
    t2 = b.add_texture(g.uniform(0.2, 1, 8 * 4 * 2), 8, 4, 2)
    t3 = b.add_texture(g.uniform(0.2, 1, 4 * 8 * 3), 4, 8, 3)
    t4 = b.add_texture(g.uniform(0.2, 1, 16 * 16 * 4), 16, 16, 4)
    t5 = b.add_texture(g.uniform(0.2, 1, 4 * 4 * 5), 4, 4, 5)
    tn = b.add_texture(g.uniform(0.5, 1, 8 * 8 * 3), 8, 8, 3)
    cube = b.add_mesh(*scenes.cube_mesh(1.0))
    plane = b.add_mesh(*scenes.plane_mesh((0, 1, 0), 4.0))
    mats = [b.add_material(albedo=(1, 1, 1), albedo_texture=t1, roughness=0.3),
            b.add_material(albedo=(0.9, 0.9, 0.9), albedo_texture=t2, roughness=0.8),
            b.add_material(albedo=(1, 1, 1), albedo_texture=t3, normal_map_texture=tn, roughness=1.0),
            b.add_material(albedo=(0, 0, 0), emissive=(9, 9, 9), emissive_texture=t4),
            b.add_material(albedo=(0.8, 0.7, 0.6), albedo_texture=t5, metallic_roughness_texture=t1, roughness=0.0),
            b.add_material(albedo=(0.7, 0.7, 0.7), albedo_texture=t4, roughness=0.6)]
    for i in range(5):
        b.add_object(cube, mats[i], scenes.translation(-3.0 + 1.5 * i, 0.0, 0.0) @ scenes.rotation_y(0.3 * i) @ scenes.scaling(1, 1 + 0.1 * i, 1))
    b.add_object(plane, mats[5], scenes.translation(0, -0.7, 0))
    return b.build("zoo", scenes.Camera((0.5, 2.5, 5.0), (0, 0, 0)), sky_color=(0.3, 0.4, 0.5), width=96, height=64, samples=4, bounces=6)


@pytest.mark.parametrize("accel", ACCELS)
def test_materials_zoo_against_oracle(accel):
    scene = materials_zoo()
    frame = scene.frame()
    ref = oracle_render(scene, frame)
    got = render_all(scene, frame, accel)
    assert_same(got, ref["image"], ref["primary_object"], ref["primary_flat_id"], ref["primary_t"].view(np.uint32), ref["ray_count"])


def dead_scatter_scene(bounces):
    """Surfaces whose scatter result is never used: the last segment of a path (small bounce limits) and
    throughputs that become exactly zero (albedo channels that are exactly 0 meeting each other), next to the
    cases in which the kernel must NOT take that shortcut: an albedo texture, a normal map, channels that are
    tiny but not zero, albedo beyond the bound of the shortcut, roughness outside [0, 1]."""
    b = scenes.SceneBuilder()
    g = np.random.Generator(np.random.PCG64(11))
    tex = b.add_texture(np.where(g.uniform(0, 1, 8 * 8 * 3) < 0.5, 0.0, 1.0), 8, 8, 3)   # texels exactly 0 or 1
    tn = b.add_texture(g.uniform(0.5, 1, 8 * 8 * 3), 8, 8, 3)
    cube = b.add_mesh(*scenes.cube_mesh(1.0))
    plane = b.add_mesh(*scenes.plane_mesh((0, 1, 0), 4.0))
    mats = [b.add_material(albedo=(1, 0, 0), roughness=0.5), b.add_material(albedo=(0, 1, 0), roughness=1.0),
            b.add_material(albedo=(0, 0, 1), roughness=0.0), b.add_material(albedo=(1e-30, 0, 1e-38), roughness=0.3),
            b.add_material(albedo=(2e30, 0, 0), roughness=0.7), b.add_material(albedo=(0, 3e15, 0), roughness=3.0),
            b.add_material(albedo=(1, 1, 0), albedo_texture=tex, roughness=-0.5),
            b.add_material(albedo=(0, 1, 1), normal_map_texture=tn, roughness=0.2),
            b.add_material(albedo=(0, 0, 0), emissive=(3, 2, 1))]
    k = 0
    for ix in range(3):
        for iz in range(3):
            b.add_object(cube, mats[k], scenes.translation(-1.6 + 1.6 * ix, 0.0, -1.6 + 1.6 * iz) @ scenes.rotation_y(0.4 * k) @ scenes.scaling(1.0, 0.6 + 0.2 * k, 1.0))
            k += 1
    b.add_object(plane, b.add_material(albedo=(1, 1, 1), roughness=0.4), scenes.translation(0, -0.9, 0))
    return b.build("dead", scenes.Camera((2.5, 3.0, 4.5), (0, 0, 0)), sky_color=(0.3, 0.4, 0.5), width=96, height=64, samples=6, bounces=bounces)


@pytest.mark.parametrize("bounces", [1, 2, 3, 10])
def test_unused_scatters_against_oracle(bounces):
    scene = dead_scatter_scene(bounces)
    frame = scene.frame()
    ref = oracle_render(scene, frame)
    for accel in (abi.PT_ACCEL_AUTO, abi.PT_ACCEL_BVH):
        got = render_all(scene, frame, accel)
        # (the 2e30 albedo overflows to inf on paths that meet it twice: bits are compared, not an RMSE)
        assert np.array_equal(got["primary_flat_id"], ref["primary_flat_id"])
        assert np.array_equal(got["ray_count"], ref["ray_count"])
        assert np.array_equal(got["image"].view(np.uint32), ref["image"].view(np.uint32)), "image not bit-identical"


def test_in_range_sqrt_is_correctly_rounded():
    """cosine_sample's square roots use a shorter sequence than the compiler's sqrtf; the checker compares the two on
    the device for every f32 in [2^-96, 2^64) and for zero (tools/micro/sqrt_check.hip, built by build())."""
    import subprocess
    from pathlib import Path
    exe = Path(__file__).resolve().parent.parent / "tools" / "micro" / "sqrt_check"
    assert exe.exists(), "tools/micro/sqrt_check is not built (run __graft_entry__.build())"
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 differ" in r.stdout


def test_axis_parallel_rays_in_staged_scene_against_oracle():
    """Rays with zero direction components through the LDS-staged scene's box tests (which pick the near and far plane
    of an axis by the sign of the reciprocal direction): a camera on the z axis with an odd image size sends its centre
    row, column and pixel along the axes, and mirrors facing the axes send them straight back."""
    b = scenes.SceneBuilder()
    cube = b.add_mesh(*scenes.cube_mesh(1.0))
    plane = b.add_mesh(*scenes.plane_mesh((0, 1, 0), 4.0))
    mirror = b.add_material(albedo=(0.9, 0.9, 0.9), roughness=0.0)
    b.add_object(cube, mirror, scenes.translation(0, 0, 0))
    b.add_object(cube, b.add_material(albedo=(0.8, 0.3, 0.2), roughness=0.0), scenes.translation(0, 0, 9))    # behind the camera
    b.add_object(cube, b.add_material(albedo=(0.2, 0.8, 0.3), roughness=1.0), scenes.translation(2.0, 0, 0))
    b.add_object(cube, b.add_material(albedo=(0, 0, 0), emissive=(2, 2, 2)), scenes.translation(-2.0, 0, 0))
    b.add_object(plane, b.add_material(albedo=(0.5, 0.5, 0.5), roughness=0.0), scenes.translation(0, -0.5, 0))
    scene = b.build("axes", scenes.Camera((0, 0, 5), (0, 0, 0)), sky_color=(0.2, 0.3, 0.5), width=33, height=33, samples=6, bounces=8)
    frame = scene.frame()
    ref = oracle_render(scene, frame)
    for accel in ACCELS:
        got = render_all(scene, frame, accel)
        assert_same(got, ref["image"], ref["primary_object"], ref["primary_flat_id"], ref["primary_t"].view(np.uint32), ref["ray_count"])


def test_query_random_rays_against_oracle():
    scene = scenes.instanced_boxes(n_objects=3000, width=8, height=8, samples=1, extent=10.0)
    g = np.random.Generator(np.random.PCG64(99))
    n = 4096
    rays = np.zeros(n, dtype=abi.RAY_DT)
    rays["pos"] = g.uniform(-14, 14, (n, 3)).astype(np.float32)
    d = g.normal(size=(n, 3))
    rays["dir"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays["dir"][:64, 0] = 0.0      # axis-parallel components (infinite reciprocals)
    rays["dir"][64:128, 1] = 0.0
    want = oracle_query(scene, rays)
    for accel in ACCELS:
        with abi.Context(accel=accel) as ctx:
            ctx.upload_scene(scene)
            got = ctx.query(rays)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"accel {accel}"
    assert (want["object"] != abi.PT_U32_MAX).sum() > 500


def test_tie_rules():
    b = scenes.SceneBuilder()
    m = b.add_mesh(*scenes.plane_mesh((0, 1, 0), 1.0))
    mat = b.add_material()
    b.add_object(m, mat)
    b.add_object(m, mat)
    scene = b.build("tie", scenes.Camera((0, 3, 0), (0, 0, 0)))
    rays = np.zeros(2, dtype=abi.RAY_DT)
    rays["pos"] = [(0.25, 2.0, 0.25), (0.5, 2.0, -0.25)]
    rays["dir"] = [(0, -1, 0)] * 2
    want = oracle_query(scene, rays)
    for accel in ACCELS:
        with abi.Context(accel=accel, tie_rule=abi.PT_TIE_REFERENCE) as ctx:
            ctx.upload_scene(scene)
            h = ctx.query(rays)
        assert np.array_equal(h.view(np.uint32), want.view(np.uint32))
        assert list(h["object"]) == [1, 1] and h["triangle"][0] == 1
        with abi.Context(accel=accel, tie_rule=abi.PT_TIE_LOWEST) as ctx:
            ctx.upload_scene(scene)
            h = ctx.query(rays)
        assert list(h["object"]) == [0, 0] and h["triangle"][0] == 0
        assert np.array_equal(h["t"], want["t"])


# ---- full benchmark sizes: properties that need no checker -----------------------------------
def test_boxes_1080p_accel_invariance_and_accounting():
    scene = scenes.boxes(1920, 1080, 64, 10)
    frame = scene.frame()
    a = render_all(scene, frame, abi.PT_ACCEL_BRUTE)
    b = render_all(scene, frame, abi.PT_ACCEL_BVH)
    c = render_all(scene, frame, abi.PT_ACCEL_BRUTE)
    for other in (b, c):  # acceleration structure on/off and run-to-run: bit-identical
        assert np.array_equal(a["image"].view(np.uint32), other["image"].view(np.uint32))
        assert np.array_equal(a["primary_flat_id"], other["primary_flat_id"])
        assert np.array_equal(a["ray_count"], other["ray_count"])
    assert np.all(a["image"][..., 3] == 1.0) and np.isfinite(a["image"]).all()
    st = a["stats"]
    assert st["paths"] == 1920 * 1080 * 64
    assert st["rays_reference"] == int(a["ray_count"].astype(np.uint64).sum())
    assert st["rays"] == st["rays_reference"] - 1920 * 1080 * 63
    # a crop of the same frame against the CPU restatement
    rect = (900, 500, 932, 524)
    ref = oracle_render(scene, frame, rect=rect)
    x0, y0, x1, y1 = rect
    assert np.array_equal(a["image"][y0:y1, x0:x1].view(np.uint32), ref["image"].view(np.uint32))
    assert np.array_equal(a["primary_flat_id"][y0:y1, x0:x1], ref["primary_flat_id"])


def test_boxes_1080p_repeated_frames_bit_identical(monkeypatch):
    """Benchmark size: the first frame of a context runs in list order, later frames in the order the previous
    frame's ray counts suggest, with the samples cut into sweeps. All of them, and a context with sweeps and hints
    switched off, must produce the same bits, aux planes included."""
    scene = scenes.boxes(1920, 1080, 64, 10)
    frame = scene.frame()
    with monkeypatch.context() as mp:
        mp.setenv("PT_AMD_SWEEPS", "1")
        mp.setenv("PT_AMD_NO_COST_HINTS", "1")
        with abi.Context() as ctx:
            ctx.upload_scene(scene)
            ref = ctx.render(frame)
            ref_rays = ctx.read_aux(abi.PT_AUX_RAY_COUNT)
            ref_stats = ctx.stats()
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        for _ in range(3):
            img = ctx.render(frame)
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
            assert np.array_equal(ctx.read_aux(abi.PT_AUX_RAY_COUNT), ref_rays)
            assert ctx.stats()["rays"] == ref_stats["rays"] and ctx.stats()["rays_reference"] == int(ref_rays.astype(np.uint64).sum())
        # another `time` changes the seeds but is the same view: hints stay in use, the image is that of a fresh context
        img_t = ctx.render(scene.frame(time=1.5))
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        assert np.array_equal(ctx.render(scene.frame(time=1.5)).view(np.uint32), img_t.view(np.uint32))


def test_scene_glb_1080p_accel_invariance_and_oracle_crop(monkeypatch):
    """Benchmark resolution of the glb workload (fewer samples): exhaustive loop, object list + BLAS, and the repeated
    frame that stages the BLAS in LDS all give the same bits; a crop over Suzanne against the CPU restatement."""
    scene = scenes.scene_glb_textured(1920, 1080, 8, 10)
    frame = scene.frame()
    a = render_all(scene, frame, abi.PT_ACCEL_BRUTE)
    b = render_all(scene, frame, abi.PT_ACCEL_BVH)
    assert np.array_equal(a["image"].view(np.uint32), b["image"].view(np.uint32))
    assert np.array_equal(a["primary_flat_id"], b["primary_flat_id"])
    assert np.array_equal(a["primary_t"].view(np.uint32), b["primary_t"].view(np.uint32))
    assert np.array_equal(a["ray_count"], b["ray_count"])
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        for _ in range(3):   # from the second frame on the path kernel runs with the previous frame's ray counts as hints
            img = ctx.render(frame)
            assert np.array_equal(img.view(np.uint32), a["image"].view(np.uint32))
            assert np.array_equal(ctx.read_aux(abi.PT_AUX_RAY_COUNT), a["ray_count"])
    with monkeypatch.context() as mp:   # the most expensive chunks handed out whole instead of in parts of 16 entries
        mp.setenv("PT_AMD_SPLIT_CHUNKS", "0")
        with abi.Context() as ctx:
            ctx.upload_scene(scene)
            for _ in range(3):
                assert np.array_equal(ctx.render(frame).view(np.uint32), a["image"].view(np.uint32))
                assert np.array_equal(ctx.read_aux(abi.PT_AUX_RAY_COUNT), a["ray_count"])
    assert np.isfinite(a["image"]).all() and np.all(a["image"][..., 3] == 1.0)
    rect = (944, 520, 976, 544)
    ref = oracle_render(scene, frame, rect=rect)
    x0, y0, x1, y1 = rect
    assert (ref["primary_object"] != abi.PT_U32_MAX).any()
    assert np.array_equal(a["image"][y0:y1, x0:x1].view(np.uint32), ref["image"].view(np.uint32))
    assert np.array_equal(a["primary_flat_id"][y0:y1, x0:x1], ref["primary_flat_id"])
    assert np.array_equal(a["ray_count"][y0:y1, x0:x1], ref["ray_count"])


def test_cornell_4k_accel_invariance_and_oracle_crop(monkeypatch):
    """The 3840x2160 Cornell box (fewer samples than the 4096 of the benchmark configuration): the LDS-staged scene,
    the global-memory object list and the exhaustive loop agree bit for bit; repeated frames too; a crop against the
    CPU restatement; every path is accounted for."""
    scene = scenes.cornell_4k(16)
    frame = scene.frame()
    a = render_all(scene, frame, abi.PT_ACCEL_AUTO)
    b = render_all(scene, frame, abi.PT_ACCEL_BRUTE)
    with monkeypatch.context() as mp:
        mp.setenv("PT_AMD_NO_LDS_SCENE", "1")
        c = render_all(scene, frame, abi.PT_ACCEL_AUTO)
    for other in (b, c):
        assert np.array_equal(a["image"].view(np.uint32), other["image"].view(np.uint32))
        assert np.array_equal(a["primary_flat_id"], other["primary_flat_id"])
        assert np.array_equal(a["ray_count"], other["ray_count"])
    st = a["stats"]
    assert st["paths"] == 3840 * 2160 * 16
    assert st["rays_reference"] == int(a["ray_count"].astype(np.uint64).sum())
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        for _ in range(2):
            assert np.array_equal(ctx.render(frame).view(np.uint32), a["image"].view(np.uint32))
    rect = (1900, 1060, 1932, 1084)
    ref = oracle_render(scene, frame, rect=rect)
    x0, y0, x1, y1 = rect
    assert np.array_equal(a["image"][y0:y1, x0:x1].view(np.uint32), ref["image"].view(np.uint32))
    assert np.array_equal(a["ray_count"][y0:y1, x0:x1], ref["ray_count"])


@pytest.mark.parametrize("name,make,rect", [
    ("scene_glb_textured 1080p 256 spp", lambda: scenes.scene_glb_textured(1920, 1080, 256, 10), (944, 520, 976, 544)),
    ("cornell 3840x2160 4096 spp", lambda: scenes.cornell_4k(4096), (1900, 1060, 1924, 1076)),
])
def test_configurations_at_their_stated_sample_counts(name, make, rect):
    """The two benchmark configurations whose other tests use fewer samples, once at full size: the exhaustive loop and the
    accelerated path agree bit for bit on the whole frame, every path and ray is accounted for, and a crop agrees with the
    CPU restatement."""
    scene = make()
    frame = scene.frame()
    a = render_all(scene, frame, abi.PT_ACCEL_AUTO)
    b = render_all(scene, frame, abi.PT_ACCEL_BRUTE)
    assert np.array_equal(a["image"].view(np.uint32), b["image"].view(np.uint32))
    assert np.array_equal(a["ray_count"], b["ray_count"])
    assert np.array_equal(a["primary_flat_id"], b["primary_flat_id"])
    assert np.isfinite(a["image"]).all() and np.all(a["image"][..., 3] == 1.0)
    st = a["stats"]
    assert st["paths"] == frame.width * frame.height * frame.settings.samples
    assert st["rays_reference"] == int(a["ray_count"].astype(np.uint64).sum())
    assert st["rays"] == st["rays_reference"] - frame.width * frame.height * (frame.settings.samples - 1)
    x0, y0, x1, y1 = rect
    ref = oracle_render(scene, frame, rect=rect)
    assert (ref["primary_object"] != abi.PT_U32_MAX).any()
    assert np.array_equal(a["image"][y0:y1, x0:x1].view(np.uint32), ref["image"].view(np.uint32))
    assert np.array_equal(a["ray_count"][y0:y1, x0:x1], ref["ray_count"])
    assert np.array_equal(a["primary_flat_id"][y0:y1, x0:x1], ref["primary_flat_id"])


def test_million_triangles_tlas_vs_loop_and_oracle_crop():
    scene = scenes.instanced_boxes(83334, 1920, 1080, 1, 10)
    assert scene.n_triangles == 1000008
    frame = scene.frame(width=480, height=270)
    a = render_all(scene, frame, abi.PT_ACCEL_AUTO)
    assert a["accel"]["errors"] == 0 and a["accel"]["tlas_depth"] > 0
    b = render_all(scene, frame, abi.PT_ACCEL_BRUTE)
    assert np.array_equal(a["image"].view(np.uint32), b["image"].view(np.uint32))
    assert np.array_equal(a["primary_flat_id"], b["primary_flat_id"])
    assert np.array_equal(a["primary_t"].view(np.uint32), b["primary_t"].view(np.uint32))
    assert np.array_equal(a["ray_count"], b["ray_count"])
    rect = (232, 128, 248, 140)
    ref = oracle_render(scene, frame, rect=rect)
    x0, y0, x1, y1 = rect
    assert np.array_equal(a["image"][y0:y1, x0:x1].view(np.uint32), ref["image"].view(np.uint32))
    assert np.array_equal(a["primary_flat_id"][y0:y1, x0:x1], ref["primary_flat_id"])
    # full 1080p through the TLAS: finite, alpha 1, most pixels see geometry
    full = render_all(scene, scene.frame(), abi.PT_ACCEL_AUTO)
    assert np.isfinite(full["image"]).all() and np.all(full["image"][..., 3] == 1.0)
    assert (full["primary_object"] != abi.PT_U32_MAX).mean() > 0.3


@pytest.mark.parametrize("name,spp", [("cornell", 32), ("boxes", 32), ("instanced_boxes", 8), ("scene_glb_textured", 8)])
def test_schedule_invariance(name, spp, golden_dir, monkeypatch):
    """The order in which k_paths takes the pixels (sample sweeps, cost hints from the previous frame, whole-chunk
    or per-lane refill) never shows in a result: repeated frames of one context and contexts with other schedules
    all reproduce the golden frame bit for bit."""
    scene, _ = golden_scenes()[name]
    scene.samples = spp
    frame = scene.frame()
    ref = oracle_render(scene, frame)
    want = (ref["image"], ref["primary_object"], ref["primary_flat_id"], ref["primary_t"].view(np.uint32), ref["ray_count"])

    def run(n_frames):
        with abi.Context() as ctx:
            ctx.upload_scene(scene)
            for _ in range(n_frames):   # from the second frame on the previous frame's ray counts order the work
                img = ctx.render(frame)
                got = dict(image=img, primary_object=ctx.read_aux(abi.PT_AUX_PRIMARY_OBJECT),
                           primary_flat_id=ctx.read_aux(abi.PT_AUX_PRIMARY_FLAT_ID), primary_t=ctx.read_aux(abi.PT_AUX_PRIMARY_T),
                           ray_count=ctx.read_aux(abi.PT_AUX_RAY_COUNT), accel=ctx.check_accel())
                assert_same(got, *want)

    run(3)   # (the glb scene: from the second frame on its big mesh's BLAS is staged in LDS)
    for env in ({"PT_AMD_SWEEPS": "1"}, {"PT_AMD_SWEEPS": "8"}, {"PT_AMD_NO_COST_HINTS": "1"}, {"PT_AMD_REFILL_MIN": "1"},
                {"PT_AMD_NO_LDS_SCENE": "1"}, {"PT_AMD_NO_LDS_SCENE": "1", "PT_AMD_SWEEPS": "3"}, {"PT_AMD_NO_LDS_BLAS": "1"}, {"PT_AMD_WALK_QUOTA": "1"}, {"PT_AMD_WALK_QUOTA": "64"},
                {"PT_AMD_SPLIT_CHUNKS": "0"}, {"PT_AMD_SPLIT_CHUNKS": "0", "PT_AMD_NO_LDS_SCENE": "1"}, {"PT_AMD_SPLIT_CHUNKS": "1", "PT_AMD_NO_LDS_SCENE": "1"},
                {"PT_AMD_LIST_REGS": "roomy", "PT_AMD_NO_LDS_SCENE": "1"}, {"PT_AMD_LIST_REGS": "roomy", "PT_AMD_SPLIT_FRACTION": "1", "PT_AMD_SPLIT_LOAD": "0"},
                {"PT_AMD_REFILL_MIN": "1", "PT_AMD_SPLIT_FRACTION": "1", "PT_AMD_SPLIT_LOAD": "0", "PT_AMD_NO_LDS_SCENE": "1"},
                {"PT_AMD_REFILL_MIN": "16", "PT_AMD_SPLIT_FRACTION": "1", "PT_AMD_SPLIT_LOAD": "0"}):
        with monkeypatch.context() as mp:
            for k, v in env.items():
                mp.setenv(k, v)
            run(2)


@pytest.mark.parametrize("name,env", [
    ("glb", {}),                                   # object list, the big mesh's BLAS staged in LDS
    ("glb", {"PT_AMD_NO_LDS_BLAS": "1"}),          # object list, every BLAS in global memory
    ("boxes", {"PT_AMD_NO_LDS_SCENE": "1"}),       # object list, meshes too small for a BLAS: the triangle loop
    ("overlap", {"PT_AMD_NO_LDS_SCENE": "1"}),     # twelve overlapping objects with and without BLAS, coincident surfaces
    ("overlap", {"PT_AMD_NO_LDS_SCENE": "1", "PT_AMD_NO_LDS_BLAS": "1"}),
])
def test_four_lanes_to_a_ray_changes_no_result(name, env, monkeypatch):
    """Chunks handed out in parts of 16 entries are traced with four lanes to a ray (one child box, one leaf triangle,
    every fourth triangle of a small mesh per lane). With every chunk split (PT_AMD_SPLIT_FRACTION=1) from the second
    frame on, the frames stay those of the exhaustive loop bit for bit and of the CPU restatement on a crop, and the
    per-wave records show that such trace steps ran."""
    if name == "glb":
        scene, rect = scenes.scene_glb_textured(320, 180, 16, 10), (150, 80, 182, 104)
    elif name == "boxes":
        scene, rect = scenes.boxes(320, 180, 16, 10), (150, 80, 182, 104)
    else:
        scene, rect = scenes.overlapping_meshes(), (120, 64, 144, 80)
    frame = scene.frame()
    want = render_all(scene, frame, abi.PT_ACCEL_BRUTE)
    ref = oracle_render(scene, frame, rect=rect)
    x0, y0, x1, y1 = rect
    assert np.array_equal(want["image"][y0:y1, x0:x1].view(np.uint32), ref["image"].view(np.uint32))
    assert np.array_equal(want["ray_count"][y0:y1, x0:x1], ref["ray_count"])
    with monkeypatch.context() as mp:
        mp.setenv("PT_AMD_SPLIT_FRACTION", "1")
        for k, v in env.items():
            mp.setenv(k, v)
        with abi.Context() as ctx:
            ctx.upload_scene(scene)
            quad_steps = []
            for i in range(4):
                ctx.set_stats_detail(i == 3)
                img = ctx.render(frame)
                assert np.array_equal(img.view(np.uint32), want["image"].view(np.uint32)), f"frame {i}"
                assert np.array_equal(ctx.read_aux(abi.PT_AUX_RAY_COUNT), want["ray_count"])
                assert np.array_equal(ctx.read_aux(abi.PT_AUX_PRIMARY_FLAT_ID), want["primary_flat_id"])
                if i == 3:
                    quad_steps.append(int(ctx.wave_log()[:, 12].astype(np.uint64).sum()))
            assert quad_steps[0] > 0, "no trace step ran with four lanes to a ray: the test does not cover that path"
            assert ctx.check_accel()["errors"] == 0


def test_split_chunks_with_a_short_last_chunk_at_1080p(monkeypatch):
    """1080p frames of the glb scene at 16 spp with `time` advancing, on one context: from the second frame on the most
    expensive chunks are handed out in parts. The path list's last chunk is short here (its fourth part is empty), the
    case in which a wave goes on from an empty part to a whole chunk. Every frame must be the one the exhaustive loop
    renders without any splitting, in every pixel and ray count."""
    scene = scenes.scene_glb_textured(1920, 1080, 16, 10)
    times = [0.5, 1.5, 2.5, 3.5]
    want = []
    with monkeypatch.context() as mp:
        mp.setenv("PT_AMD_SPLIT_CHUNKS", "0")
        with abi.Context(accel=abi.PT_ACCEL_BRUTE) as ctx:
            ctx.upload_scene(scene)
            for t in times:
                want.append((ctx.render(scene.frame(time=t)).copy(), ctx.read_aux(abi.PT_AUX_RAY_COUNT).copy()))
    # (the defaults twice; then the kernel build for four waves per SIMD where the default picks the one with more registers
    # for such a frame, every heavy chunk in parts, and as few workgroups as the library ever keeps)
    for env in ({}, {}, {"PT_AMD_LIST_REGS": "tight"}, {"PT_AMD_SPLIT_FRACTION": "1", "PT_AMD_SPLIT_LOAD": "0"}, {"PT_AMD_GROUP_MARGIN": "1"}):
        with monkeypatch.context() as mp:
            for k, v in env.items():
                mp.setenv(k, v)
            with abi.Context() as ctx:
                ctx.upload_scene(scene)
                for j, t in enumerate(times):
                    img = ctx.render(scene.frame(time=t))
                    assert np.array_equal(img.view(np.uint32), want[j][0].view(np.uint32)), f"{env}, frame {j}"
                    assert np.array_equal(ctx.read_aux(abi.PT_AUX_RAY_COUNT), want[j][1]), f"{env}, frame {j}"


def test_hints_from_another_view_change_no_result():
    """The schedule hints are taken from the context's previous frame even when the camera has moved since: the frame
    must be what a fresh context renders."""
    scene = scenes.boxes(320, 180, 16, 10)
    frame_a = scene.frame(time=0.5)
    cam_b = scenes.Camera(tuple(np.asarray(scene.camera.eye) + np.array([0.4, 0.1, -0.3])), scene.camera.target, scene.camera.fov_y)
    wfv, vfc = cam_b.matrices(320, 180)
    frame_b = abi.make_frame(wfv, vfc, 320, 180, time=1.5, bounces=10, samples=16, sky_color=scene.sky_color)
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        want = ctx.render(frame_b)
        want_rays = ctx.read_aux(abi.PT_AUX_RAY_COUNT)
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        ctx.render(frame_a)
        got = ctx.render(frame_b)       # ordered by the ray counts of frame_a
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        assert np.array_equal(ctx.read_aux(abi.PT_AUX_RAY_COUNT), want_rays)


def test_partition_union_equals_full_frame():
    scene = scenes.boxes(200, 120, 4, 10)
    full = render_all(scene, scene.frame())
    acc = np.zeros_like(full["image"])
    rays = 0
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        for k in range(3):
            part = ctx.render(scene.frame(part_index=k, part_count=3))
            owned = part[..., 3] == 1.0
            assert not np.any(owned & (acc[..., 3] == 1.0))  # disjoint
            acc += part
            rays += ctx.stats()["rays"]
    assert np.array_equal(acc.view(np.uint32), full["image"].view(np.uint32))
    assert rays == full["stats"]["rays"]


def test_accumulation_is_mean_of_frames():
    scene = scenes.cornell_box(64, 64, 4, 10)
    times = [0.0, 1.5, 2.5]
    frames = [oracle_render(scene, scene.frame(time=t))["image"] for t in times]
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        outs = []
        for i, t in enumerate(times):
            flags = abi.PT_FRAME_ACCUMULATE | (abi.PT_FRAME_RESET if i == 0 else 0)
            outs.append(ctx.render(scene.frame(time=t, flags=flags)))
    acc = np.zeros_like(frames[0][..., :3])
    for i, f in enumerate(frames):
        acc = acc + f[..., :3]
        want = acc / np.float32(i + 1)
        assert np.array_equal(outs[i][..., :3].view(np.uint32), want.view(np.uint32))
        assert np.all(outs[i][..., 3] == 1.0)


def test_accumulation_restarts_after_a_scene_upload():
    """Uploads between accumulated frames: the identical object array (what extract_visible sends every frame,
    reference src/extract.rs:297-346) leaves the running sum alone; a changed scene starts a new one, without the
    caller passing PT_FRAME_RESET."""
    scene = scenes.cornell_box(48, 40, 3, 6)
    moved = scenes.cornell_box(48, 40, 3, 6)
    moved.objects = moved.objects.copy()
    moved.objects["local_to_world"][6, 13] -= 0.25    # the light, a little lower
    moved.objects["world_to_local"][6, 13] += 0.25
    times = [0.0, 1.5, 2.5, 3.5]
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        outs = [ctx.render(scene.frame(time=times[0], flags=abi.PT_FRAME_ACCUMULATE | abi.PT_FRAME_RESET))]
        ctx.upload_objects(scene.objects)                 # same bytes: the sum goes on
        outs.append(ctx.render(scene.frame(time=times[1], flags=abi.PT_FRAME_ACCUMULATE)))
        ctx.upload_objects(moved.objects)                 # another scene: a new sum
        outs.append(ctx.render(moved.frame(time=times[2], flags=abi.PT_FRAME_ACCUMULATE)))
        outs.append(ctx.render(moved.frame(time=times[3], flags=abi.PT_FRAME_ACCUMULATE)))
    f = [oracle_render(scene, scene.frame(time=times[0]))["image"][..., :3], oracle_render(scene, scene.frame(time=times[1]))["image"][..., :3],
         oracle_render(moved, moved.frame(time=times[2]))["image"][..., :3], oracle_render(moved, moved.frame(time=times[3]))["image"][..., :3]]
    want = [f[0] / np.float32(1), (f[0] + f[1]) / np.float32(2), f[2] / np.float32(1), (f[2] + f[3]) / np.float32(2)]
    for k in range(4):
        assert np.array_equal(outs[k][..., :3].view(np.uint32), want[k].view(np.uint32)), f"frame {k}"
    assert not np.array_equal(f[1], f[2])


# ---- edge cases and error behaviour -------------------------------------------------------------
def test_output_f16_and_pinned_host_buffer():
    """PT_FRAME_OUTPUT_F16 writes the f32 result rounded to nearest-even binary16 (the reference's Rgba16Float
    target, src/shader.rs:321); a page-locked staging buffer from pt_host_alloc only changes how fast the copy is."""
    scene = scenes.cornell_box(64, 48, 4, 10)
    with abi.Context() as ctx:
        ctx.upload_scene(scene)
        ref = ctx.render(scene.frame())
        want16 = ref.astype(np.float16)
        got16 = ctx.render(scene.frame(flags=abi.PT_FRAME_OUTPUT_F16))
        assert got16.dtype == np.float16 and np.array_equal(got16.view(np.uint16), want16.view(np.uint16))
        staging = ctx.host_array(ref.shape, np.float32)
        staging16 = ctx.host_array(ref.shape, np.float16)
        for _ in range(2):
            out = ctx.render(scene.frame(), out=staging)
            assert out is staging and np.array_equal(staging.view(np.uint32), ref.view(np.uint32))
            ctx.render(scene.frame(flags=abi.PT_FRAME_OUTPUT_F16), out=staging16)
            assert np.array_equal(staging16.view(np.uint16), want16.view(np.uint16))
        with pytest.raises(ValueError):
            ctx.render(scene.frame(), out=staging16)   # wrong element type for this frame


def test_empty_scene_renders_sky():
    b = scenes.SceneBuilder()
    scene = b.build("empty", scenes.Camera((3, 3, 3), (0, 0, 0)), sky_color=(0.1, 0.2, 0.4), width=40, height=24, samples=3, bounces=5)
    for accel in ACCELS:
        got = render_all(scene, scene.frame(), accel)
        ref = oracle_render(scene, scene.frame())
        assert np.array_equal(got["image"].view(np.uint32), ref["image"].view(np.uint32))
        assert np.all(got["primary_object"] == abi.PT_U32_MAX) and np.all(got["ray_count"] == 3)


def test_zero_bounces_and_zero_samples():
    scene = scenes.boxes(32, 16, 3, 4)
    got = render_all(scene, scene.frame(bounces=0))
    assert got["stats"]["rays"] == 0 and np.all(got["image"][..., :3] == 0) and np.all(got["ray_count"] == 0)
    got = render_all(scene, scene.frame(samples=0))
    assert np.all(np.isnan(got["image"][..., :3])) and got["stats"]["rays"] == 0


def test_mesh_without_triangles_and_single_triangle():
    b = scenes.SceneBuilder()
    tri = b.add_mesh([(0, 0, 0), (1, 0, 0), (0, 0, -1)], [(0, 1, 0)] * 3, [(0, 0), (1, 0), (0, 1)], [0, 1, 2])
    empty = b.add_mesh([(0, 0, 0)], [(0, 1, 0)], [(0, 0)], [])
    mat = b.add_material(albedo=(0.5, 0.5, 0.5))
    b.add_object(empty, mat)
    b.add_object(tri, mat, scenes.scaling(2, 2, 2))
    scene = b.build("one", scenes.Camera((1, 4, 1), (0.5, 0, -0.5)), sky_color=(1, 1, 1), width=48, height=48, samples=2, bounces=3)
    ref = oracle_render(scene, scene.frame())
    for accel in ACCELS:
        got = render_all(scene, scene.frame(), accel)
        assert_same(got, ref["image"], ref["primary_object"], ref["primary_flat_id"], ref["primary_t"].view(np.uint32), ref["ray_count"])
    assert (ref["primary_object"] == 1).sum() > 10


def test_error_codes():
    scene = scenes.cornell_box(16, 16, 1, 2)
    with abi.Context() as ctx:
        with pytest.raises(abi.PtError) as e:
            ctx.render(scene.frame())
        assert e.value.code == abi.PT_ERR_NOT_READY
        bad = scene.indices.copy()
        bad[3] = 1000
        with pytest.raises(abi.PtError) as e:
            ctx.upload_meshes(scene.meshes, bad, scene.vertices)
        assert e.value.code == abi.PT_ERR_INVALID_SCENE
        badm = scene.meshes.copy()
        badm["tri_count"][0] = 50
        with pytest.raises(abi.PtError) as e:
            ctx.upload_meshes(badm, scene.indices, scene.vertices)
        assert e.value.code == abi.PT_ERR_INVALID_SCENE
        tight = scene.meshes.copy()
        tight["aabb_max"][0, 0] = 0.5       # the box no longer encloses the mesh
        with pytest.raises(abi.PtError) as e:
            ctx.upload_meshes(tight, scene.indices, scene.vertices)
        assert e.value.code == abi.PT_ERR_INVALID_SCENE
        nanv = scene.vertices.copy()
        nanv["position"][2, 1] = np.nan
        with pytest.raises(abi.PtError) as e:
            ctx.upload_meshes(scene.meshes, scene.indices, nanv)
        assert e.value.code == abi.PT_ERR_INVALID_SCENE
        ctx.upload_meshes(scene.meshes, scene.indices, scene.vertices)
        badmat = scene.materials.copy()
        badmat["albedo_texture"][0] = 3
        with pytest.raises(abi.PtError) as e:
            ctx.upload_materials(badmat, scene.textures, scene.texture_data)
        assert e.value.code == abi.PT_ERR_INVALID_SCENE
        ctx.upload_materials(scene.materials, scene.textures, scene.texture_data)
        bado = scene.objects.copy()
        bado["mesh"][1] = 77
        with pytest.raises(abi.PtError) as e:
            ctx.upload_objects(bado)
        assert e.value.code == abi.PT_ERR_INVALID_SCENE
        bado = scene.objects.copy()
        bado["mat"][1] = 77
        ctx.upload_objects(bado)  # materials are checked when the frame is rendered
        with pytest.raises(abi.PtError) as e:
            ctx.render(scene.frame())
        assert e.value.code == abi.PT_ERR_INVALID_SCENE
        ctx.upload_objects(scene.objects)
        with pytest.raises(abi.PtError) as e:
            ctx.render(scene.frame(part_index=2, part_count=2))
        assert e.value.code == abi.PT_ERR_INVALID_ARGUMENT
        img = ctx.render(scene.frame())  # the context is still usable after errors
        assert np.isfinite(img).all()


def test_object_update_refits_tlas_with_identical_results():
    """pt_upload_objects with an unchanged object count keeps the TLAS topology and refits its boxes (the per-frame
    update of extract_visible, reference src/extract.rs:297-346). The image must not depend on it."""
    a = scenes.instanced_boxes(n_objects=600, width=96, height=64, samples=2, extent=6.0)
    b = scenes.instanced_boxes(n_objects=600, width=96, height=64, samples=2, extent=6.0, seed=1234)   # every object somewhere else
    frame = a.frame()
    fresh = render_all(b, frame)
    assert not fresh["stats"]["accel_used"] & 4
    with abi.Context() as ctx:
        ctx.upload_scene(a)
        ctx.render(frame)
        for k in range(3):      # refit, refit, refit
            ctx.upload_objects(b.objects if k % 2 == 0 else a.objects)
        ctx.upload_materials(b.materials, b.textures, b.texture_data)   # the two scenes draw different material colours
        ctx.upload_objects(b.objects)
        img = ctx.render(frame)
        st = ctx.stats()
        assert st["accel_used"] & 1 and st["accel_used"] & 4, "expected a refitted TLAS"
        assert ctx.check_accel()["errors"] == 0
        assert np.array_equal(img.view(np.uint32), fresh["image"].view(np.uint32))
        assert np.array_equal(ctx.read_aux(abi.PT_AUX_PRIMARY_FLAT_ID), fresh["primary_flat_id"])
        assert np.array_equal(ctx.read_aux(abi.PT_AUX_RAY_COUNT), fresh["ray_count"])
        # and against the CPU restatement, which knows nothing of trees
        ref = oracle_render(b, frame)
        assert np.array_equal(img.view(np.uint32), ref["image"].view(np.uint32))
        assert np.array_equal(ctx.read_aux(abi.PT_AUX_PRIMARY_OBJECT), ref["primary_object"])
        assert np.array_equal(ctx.read_aux(abi.PT_AUX_PRIMARY_FLAT_ID), ref["primary_flat_id"])
        assert np.array_equal(ctx.read_aux(abi.PT_AUX_PRIMARY_T).view(np.uint32), ref["primary_t"].view(np.uint32))
        assert np.array_equal(ctx.read_aux(abi.PT_AUX_RAY_COUNT), ref["ray_count"])
        # the same array again is recognised: no rebuild, the refit flag of the last real update stays
        ctx.upload_objects(b.objects)
        assert np.array_equal(ctx.render(frame).view(np.uint32), ref["image"].view(np.uint32))
        assert ctx.stats()["accel_used"] & 4
        # a different object count builds afresh
        ctx.upload_objects(b.objects[:500])
        ctx.render(frame)
        assert not ctx.stats()["accel_used"] & 4


def test_mesh_with_a_deep_blas_against_the_loop_and_the_oracle():
    """A mesh of 20,000 triangles (a bumpy sphere): its wide BLAS is made level by level over many workgroups, not by the
    one-workgroup kernel the small meshes get. Tree checks, the whole frame against the exhaustive loop, a crop against
    the CPU restatement."""
    nu, nv = 100, 100
    u = np.linspace(0.0, 2.0 * np.pi, nu + 1)
    v = np.linspace(0.02, np.pi - 0.02, nv + 1)
    uu, vv = np.meshgrid(u, v, indexing="ij")
    r = 1.0 + 0.08 * np.sin(7.0 * uu) * np.sin(5.0 * vv)
    pos = np.stack([r * np.sin(vv) * np.cos(uu), r * np.cos(vv), r * np.sin(vv) * np.sin(uu)], axis=-1).reshape(-1, 3)
    nrm = pos / np.linalg.norm(pos, axis=1, keepdims=True)
    uv = np.stack([uu / (2.0 * np.pi), vv / np.pi], axis=-1).reshape(-1, 2)
    idx = []
    for i in range(nu):
        for j in range(nv):
            a, b, c, d = i * (nv + 1) + j, (i + 1) * (nv + 1) + j, (i + 1) * (nv + 1) + j + 1, i * (nv + 1) + j + 1
            idx += [a, d, b, b, d, c]   # counter-clockwise seen from outside
    b = scenes.SceneBuilder()
    mesh = b.add_mesh(pos, nrm, uv, np.array(idx))
    floor = b.add_mesh(*scenes.cube_mesh(1.0))
    b.add_object(mesh, b.add_material(albedo=(0.8, 0.6, 0.3), roughness=0.4), scenes.translation(0.0, 0.3, 0.0))
    b.add_object(floor, b.add_material(albedo=(0.5, 0.5, 0.5)), scenes.translation(0.0, -1.6, 0.0) @ np.diag([8.0, 1.0, 8.0, 1.0]))
    b.add_object(floor, b.add_material(albedo=(0, 0, 0), emissive=(6, 6, 6)), scenes.translation(2.5, 3.0, 1.0))
    scene = b.build("bumpy_sphere", scenes.Camera((0, 1.5, 5), (0, 0, 0)), sky_color=(0.3, 0.4, 0.6), width=160, height=120, samples=3, bounces=6)
    assert scene.n_triangles == 20024
    frame = scene.frame()
    got = render_all(scene, frame, abi.PT_ACCEL_AUTO)
    assert got["accel"]["errors"] == 0 and got["stats"]["accel_used"] & 2
    brute = render_all(scene, frame, abi.PT_ACCEL_BRUTE)
    assert np.array_equal(got["image"].view(np.uint32), brute["image"].view(np.uint32))
    assert np.array_equal(got["primary_flat_id"], brute["primary_flat_id"])
    assert np.array_equal(got["primary_t"].view(np.uint32), brute["primary_t"].view(np.uint32))
    assert np.array_equal(got["ray_count"], brute["ray_count"])
    assert (got["primary_object"] == 0).mean() > 0.1
    rect = (70, 50, 90, 62)
    x0, y0, x1, y1 = rect
    ref = oracle_render(scene, frame, rect=rect)
    assert np.array_equal(got["image"][y0:y1, x0:x1].view(np.uint32), ref["image"].view(np.uint32))
    assert np.array_equal(got["primary_flat_id"][y0:y1, x0:x1], ref["primary_flat_id"])


def test_many_refits_then_a_fresh_build():
    """Refits keep a topology that fits the moved objects worse and worse, so every 16th update builds afresh; the trees
    must be valid after every update and the image that of a fresh context."""
    a = scenes.instanced_boxes(n_objects=400, width=64, height=48, samples=2, extent=5.0)
    b = scenes.instanced_boxes(n_objects=400, width=64, height=48, samples=2, extent=5.0, seed=99)
    frame = a.frame()
    want = {0: render_all(a, frame)["image"], 1: None}
    with abi.Context() as ctx:
        ctx.upload_scene(a)
        ctx.render(frame)
        flags = []
        for k in range(18):
            ctx.upload_objects((b if k % 2 == 0 else a).objects)
            assert ctx.check_accel()["errors"] == 0, k
            ctx.render(frame)
            flags.append(bool(ctx.stats()["accel_used"] & 4))
        assert flags == [True] * 15 + [False] + [True] * 2
        img = ctx.render(frame)     # the 18th upload was scene a's objects (materials of a were never replaced)
    assert np.array_equal(img.view(np.uint32), want[0].view(np.uint32))


def test_reupload_order_independent():
    scene = scenes.boxes(64, 36, 2, 4)
    ref = render_all(scene, scene.frame())
    with abi.Context() as ctx:
        ctx.upload_objects(scene.objects)      # objects first, meshes later
        ctx.upload_materials(scene.materials, scene.textures, scene.texture_data)
        ctx.upload_meshes(scene.meshes, scene.indices, scene.vertices)
        img = ctx.render(scene.frame())
        assert np.array_equal(img.view(np.uint32), ref["image"].view(np.uint32))
        other = scenes.cornell_box(64, 36, 2, 4)
        ctx.upload_scene(other)                # replace everything on a live context
        img2 = ctx.render(other.frame())
    assert np.array_equal(img2.view(np.uint32), render_all(other, other.frame())["image"].view(np.uint32))


def test_plugin_mirror_renders_like_the_raw_abi():
    from path_tracing_amd.plugin import Globals, RayTracePlugin, RayTraceSettings, View
    scene = scenes.boxes(96, 54, 3, 5)
    wfv, vfc = scene.camera.matrices(96, 54)
    plugin = RayTracePlugin().build().finish()
    try:
        plugin.meta.set_scene(scene)
        view = View(wfv, vfc, (0, 0, 96, 54))
        settings = RayTraceSettings(bounces=5, samples=3, sky_color=(0.1, 0.2, 0.4, 1.0))
        assert plugin.node.run(view, Globals(0.0), None) is None     # camera without RayTraceSettings: node does not run
        img = plugin.node.run(view, Globals(0.0), settings)
        # objects change every frame in the reference (extract_visible); re-upload and render again
        plugin.meta.set_objects(scene.objects)
        img2 = plugin.node.run(view, Globals(0.0), settings)
        # the same node with two frames in flight, the objects handed over before each
        bufs = [plugin.ctx.host_array((54, 96, 4), np.float32) for _ in range(2)]
        assert not plugin.node.submit(view, Globals(0.0), None, 0, bufs[0])
        for k in range(2):
            plugin.meta.set_objects(scene.objects)
            assert plugin.node.submit(view, Globals(0.5 + k), settings, k, bufs[k])
        flight = []
        for k in range(2):
            plugin.node.wait(k)
            flight.append(bufs[k].copy())
    finally:
        plugin.close()
    ref = oracle_render(scene, scene.frame())
    assert np.array_equal(img.view(np.uint32), ref["image"].view(np.uint32))
    assert np.array_equal(img2.view(np.uint32), img.view(np.uint32))
    for k in range(2):
        assert np.array_equal(flight[k].view(np.uint32), oracle_render(scene, scene.frame(time=0.5 + k))["image"].view(np.uint32))


def test_every_texture_format_branch_reaches_the_kernels():
    """Five images, one per branch of TextureData::append_texture (reference src/data.rs:105-147), go through the plugin
    mirror's extract_textures step (RayTraceMeta.set_images) into the HIP path; the frame must be the restatement's for the
    same flattened pool, bit for bit. Rgba16Float texels come out divided by 255 and Rgb9e5Ufloat through the reference's
    own byte arithmetic (both quirks kept), so the pictures are dim, but they must be the SAME dim pictures."""
    from path_tracing_amd.plugin import Globals, RayTracePlugin, RayTraceSettings, View
    g = np.random.Generator(np.random.PCG64(11))
    w, h = 8, 8
    rgba8 = g.integers(40, 256, w * h * 4, dtype=np.uint8)
    rgba16 = (g.uniform(60.0, 250.0, w * h * 4).astype(np.float16)).view(np.uint8)   # / 255 in the flattening: ends up in [0.2, 1]
    # (bytes chosen so that the half patterns the reference's byte arithmetic builds are positive and below one)
    rgb9e5 = np.stack([g.integers(20, 30, w * h, dtype=np.uint8), g.integers(10, 15, w * h, dtype=np.uint8), g.integers(5, 8, w * h, dtype=np.uint8),
                       g.integers(0, 256, w * h, dtype=np.uint8)], axis=1).reshape(-1)
    r8 = g.integers(40, 256, w * h, dtype=np.uint8)
    rg8 = g.integers(40, 256, w * h * 2, dtype=np.uint8)
    images = [(rgba8, w, h, "Rgba8UnormSrgb"), (rgba16, w, h, "Rgba16Float"), (rgb9e5, w, h, "Rgb9e5Ufloat"), (r8, w, h, "R8Unorm"),
              (rg8, w, h, "Rg8Unorm")]
    b = scenes.SceneBuilder()
    cube = b.add_mesh(*scenes.cube_mesh(1.0))
    plane = b.add_mesh(*scenes.plane_mesh((0, 1, 0), 5.0))
    for k in range(5):   # texture k on cube k (as albedo), and the 16-bit one also as an emissive texture on a sixth
        b.add_object(cube, b.add_material(albedo=(1, 1, 1), albedo_texture=k, roughness=0.6), scenes.translation(-3.0 + 1.5 * k, 0.0, 0.0))
    b.add_object(cube, b.add_material(albedo=(0.5, 0.5, 0.5), emissive=(1, 1, 1), emissive_texture=1), scenes.translation(0.0, 1.5, -1.0))
    b.add_object(plane, b.add_material(albedo=(0.7, 0.7, 0.7), roughness=0.3), scenes.translation(0.0, -0.5, 0.0))
    scene = b.build("formats", scenes.Camera((0.5, 2.5, 6.0), (0, 0, 0)), sky_color=(0.6, 0.7, 0.9), width=160, height=96, samples=6, bounces=5)
    wfv, vfc = scene.camera.matrices(scene.width, scene.height)
    plugin = RayTracePlugin().build().finish()
    try:
        plugin.meta.set_meshes(scene.meshes, scene.indices, scene.vertices)
        plugin.meta.set_images(scene.materials, images)
        plugin.meta.set_objects(scene.objects)
        img = plugin.node.run(View(wfv, vfc, (0, 0, scene.width, scene.height)), Globals(1.5),
                              RayTraceSettings(bounces=scene.bounces, samples=scene.samples, sky_color=tuple(scene.sky_color) + (1.0,)))
        prim = plugin.ctx.read_aux(abi.PT_AUX_PRIMARY_OBJECT)
        textures, pool = plugin.meta.textures, plugin.meta.texture_data
    finally:
        plugin.close()
    assert [int(t["format"]) for t in textures] == [4, 4, 3, 1, 2] and pool.size == w * h * (4 + 4 + 3 + 1 + 2)
    import dataclasses
    flat = dataclasses.replace(scene, textures=textures, texture_data=pool)
    ref = oracle_render(flat, flat.frame(time=1.5))
    assert np.array_equal(prim, ref["primary_object"])
    assert all((prim == k).sum() > 50 for k in range(6)), "every textured cube must be in view"
    assert np.array_equal(img.view(np.uint32), ref["image"].view(np.uint32))
    # the formats really differ: each cube's pixels have their own mean colour
    means = [img[prim == k][:, :3].mean(axis=0) for k in range(5)]
    assert len({tuple(np.round(m, 3)) for m in means}) == 5


def test_distributed_renderer_single_rank_device_tensor():
    """torch owns the device memory here. torch bundles its own HIP runtime, which must be initialised before
    libpt_amd.so pulls in the system one, so this check runs in a fresh process (tests/gpu_torch_worker.py)."""
    import subprocess
    import sys
    from pathlib import Path
    worker = Path(__file__).resolve().parent / "gpu_torch_worker.py"
    r = subprocess.run([sys.executable, str(worker)], capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_cpp_host_mirror_cli(tmp_path):
    """The compiled C++ mirror of the plugin surface (path_tracing_amd/host) renders the Cornell box through the C ABI."""
    import subprocess
    from pathlib import Path
    exe = Path(__file__).resolve().parent.parent / "path_tracing_amd" / "host" / "pt_render_cli"
    assert exe.exists(), "build() did not produce the C++ host demo"
    out = tmp_path / "cornell.f32"
    r = subprocess.run([str(exe), "48", "32", "3", "6", str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    img = np.fromfile(out, dtype=np.float32).reshape(32, 48, 4)
    scene = scenes.cornell_box(48, 32, 3, 6)
    ref = oracle_render(scene, scene.frame())["image"]
    assert rrmse(img, ref) <= RRMSE_TOL
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    # the same host with three frames in flight on the slots of its one context: six frames, `time` advancing
    r = subprocess.run([str(exe), "48", "32", "3", "6", str(out), "3", "6"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "6 frames with 3 in flight" in r.stdout, r.stdout + r.stderr
    frames = np.fromfile(out, dtype=np.float32).reshape(7, 32, 48, 4)
    assert np.array_equal(frames[0].view(np.uint32), ref.view(np.uint32))
    for n in range(6):
        want = oracle_render(scene, scene.frame(time=0.5 + n))["image"]
        assert np.array_equal(frames[1 + n].view(np.uint32), want.view(np.uint32)), f"frame {n}" 

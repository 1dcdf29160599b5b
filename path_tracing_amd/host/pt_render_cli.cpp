// pt_render_cli — renders the Cornell box of reference examples/cornellbox.rs:59-113 through the C++ plugin mirror
// and writes a raw little-endian f32 RGBA file plus a checksum line. Used as the compiled-host smoke test.
//   pt_render_cli <width> <height> <samples> <bounces> <out.f32> [frames in flight K [frames N]]
// With K given it also renders N frames (default 6) with `time` = 0.5, 1.5, ... keeping K of them in flight on the slots
// of the one context (RayTraceNode::submit / wait, page-locked targets) and appends them to the output file in order.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "raytrace_plugin.hpp"

using namespace path_tracing;

static void identity(float* m) { std::memset(m, 0, 64); m[0] = m[5] = m[10] = m[15] = 1.0f; }

// two-triangle square of half-size `half` through the origin in the plane spanned by s,t (s x t = normal)
static void add_plane(RayTraceMeta& meta, const float n[3], const float s[3], const float t[3], float half, uint32_t mat, const float pos[3]) {
    PtMesh me{};
    me.ihead = (uint32_t)meta.indices.size();
    me.vhead = (uint32_t)meta.vertices.size();
    me.tri_count = 2;
    const int corners[4][2] = {{-1, -1}, {1, -1}, {1, 1}, {-1, 1}};
    for (int k = 0; k < 3; ++k) { me.aabb_min[k] = 1e30f; me.aabb_max[k] = -1e30f; }
    for (auto& c : corners) {
        PtVertex v{};
        for (int k = 0; k < 3; ++k) {
            v.position[k] = half * (c[0] * s[k] + c[1] * t[k]);
            v.normal[k] = n[k];
            me.aabb_min[k] = std::fmin(me.aabb_min[k], v.position[k]);
            me.aabb_max[k] = std::fmax(me.aabb_max[k], v.position[k]);
        }
        v.uv[0] = (c[0] + 1) * 0.5f;
        v.uv[1] = (c[1] + 1) * 0.5f;
        meta.vertices.push_back(v);
    }
    const uint32_t idx[6] = {0, 1, 2, 2, 3, 0};
    for (uint32_t i : idx) meta.indices.push_back(i);
    meta.meshes.push_back(me);
    PtObject o{};
    identity(o.local_to_world);
    identity(o.world_to_local);
    for (int k = 0; k < 3; ++k) { o.local_to_world[12 + k] = pos[k]; o.world_to_local[12 + k] = -pos[k]; }
    o.mat = mat;
    o.mesh = (uint32_t)meta.meshes.size() - 1;
    meta.objects.push_back(o);
}

static PtMaterial material(float r, float g, float b, float er, float eg, float eb) {
    PtMaterial m{};
    m.albedo[0] = r; m.albedo[1] = g; m.albedo[2] = b;
    m.emissive[0] = er; m.emissive[1] = eg; m.emissive[2] = eb;
    m.roughness = 0.5f; m.metallic = 0.0f; m.reflectance = 0.5f;  // StandardMaterial defaults
    m.albedo_texture = m.emissive_texture = m.metallic_roughness_texture = m.normal_map_texture = PT_U32_MAX;
    return m;
}

int main(int argc, char** argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: %s width height samples bounces out.f32\n", argv[0]); return 2; }
    const uint32_t W = (uint32_t)std::atoi(argv[1]), H = (uint32_t)std::atoi(argv[2]);
    RayTraceSettings settings;
    settings.samples = (uint32_t)std::atoi(argv[3]);
    settings.bounces = (uint32_t)std::atoi(argv[4]);
    try {
        RayTracePlugin plugin;
        plugin.build();
        std::vector<float> img;
        View view{};
        Globals globals;
        if (plugin.node.run(view, globals, &settings, img)) { std::fprintf(stderr, "node ran before finish()\n"); return 1; }
        plugin.finish();
        RayTraceMeta& meta = plugin.meta;
        meta.materials = {material(1, 1, 1, 0, 0, 0), material(1, 0, 0, 0, 0, 0), material(0, 1, 0, 0, 0, 0), material(0, 0, 0, 1, 1, 1)};
        const float X[3] = {1, 0, 0}, Y[3] = {0, 1, 0}, Z[3] = {0, 0, 1}, NX[3] = {-1, 0, 0}, NY[3] = {0, -1, 0}, NZ[3] = {0, 0, -1};
        const float p0[3] = {0, -1, 0}, p1[3] = {0, 1, 0}, p2[3] = {0, 0, -1}, p3[3] = {0, 0, 1}, p4[3] = {-1, 0, 0}, p5[3] = {1, 0, 0}, p6[3] = {0, 0.95f, 0};
        add_plane(meta, Y, Z, X, 1.1f, 0, p0);
        add_plane(meta, NY, X, Z, 1.1f, 0, p1);
        add_plane(meta, Z, X, Y, 1.1f, 0, p2);
        add_plane(meta, NZ, Y, X, 1.1f, 0, p3);
        add_plane(meta, X, Y, Z, 1.1f, 1, p4);
        add_plane(meta, NX, Z, Y, 1.1f, 2, p5);
        add_plane(meta, NY, X, Z, 0.25f, 3, p6);
        // camera (3,3,3) looking at the origin, fov pi/4, near 0.1 (column-major matrices)
        const double eye[3] = {3, 3, 3};
        double back[3] = {3, 3, 3}, l = std::sqrt(27.0);
        for (double& v : back) v /= l;
        double right[3] = {back[2], 0.0, -back[0]};  // up x back with up = +Y
        l = std::sqrt(right[0] * right[0] + right[2] * right[2]);
        for (double& v : right) v /= l;
        const double up[3] = {back[1] * right[2] - back[2] * right[1], back[2] * right[0] - back[0] * right[2], back[0] * right[1] - back[1] * right[0]};
        identity(view.world_from_view);
        for (int k = 0; k < 3; ++k) {
            view.world_from_view[k] = (float)right[k];
            view.world_from_view[4 + k] = (float)up[k];
            view.world_from_view[8 + k] = (float)back[k];
            view.world_from_view[12 + k] = (float)eye[k];
        }
        const double f = 1.0 / std::tan(0.5 * 0.78539816339744831), aspect = (double)W / H;
        std::memset(view.view_from_clip, 0, 64);
        view.view_from_clip[0] = (float)(aspect / f);
        view.view_from_clip[5] = (float)(1.0 / f);
        view.view_from_clip[11] = (float)(1.0 / 0.1);
        view.view_from_clip[14] = -1.0f;
        view.viewport[2] = (float)W;
        view.viewport[3] = (float)H;
        if (!plugin.node.run(view, globals, nullptr, img) && !plugin.node.run(view, globals, &settings, img)) { std::fprintf(stderr, "node did not run\n"); return 1; }
        PtStats st{};
        plugin.check(pt_get_stats(plugin.context(), &st));
        std::FILE* fo = std::fopen(argv[5], "wb");
        if (!fo) { std::perror("fopen"); return 1; }
        std::fwrite(img.data(), sizeof(float), img.size(), fo);
        std::fclose(fo);
        double sum = 0;
        for (size_t i = 0; i < img.size(); i += 4) sum += img[i] + img[i + 1] + img[i + 2];
        std::printf("ok %ux%u spp=%u bounces=%u rays=%llu trace_ms=%.3f mean=%.6f\n", W, H, settings.samples, settings.bounces,
                    (unsigned long long)st.rays, st.trace_ms, sum / (3.0 * W * H));
        if (argc > 6) {
            const uint32_t K = (uint32_t)std::atoi(argv[6]), N = argc > 7 ? (uint32_t)std::atoi(argv[7]) : 6u;
            if (K < 1 || K > PT_MAX_FRAME_SLOTS) { std::fprintf(stderr, "frames in flight must be 1..%u\n", PT_MAX_FRAME_SLOTS); return 2; }
            const size_t n_floats = (size_t)W * H * 4;
            std::vector<float*> staging(K, nullptr);
            for (uint32_t k = 0; k < K; ++k) {
                void* p = nullptr;
                plugin.check(pt_host_alloc(plugin.context(), n_floats * sizeof(float), &p));
                staging[k] = static_cast<float*>(p);
            }
            fo = std::fopen(argv[5], "ab");
            if (!fo) { std::perror("fopen"); return 1; }
            for (uint32_t n = 0; n < N + K; ++n) {
                const uint32_t slot = n % K;
                if (n >= K) {   // the frame this slot has held since K submits ago
                    plugin.node.wait(slot);
                    std::fwrite(staging[slot], sizeof(float), n_floats, fo);
                }
                if (n < N) {
                    Globals g;
                    g.time = 0.5f + (float)n;
                    meta.objects_dirty = true;   // extract_visible hands the objects over every frame
                    if (!plugin.node.submit(view, g, &settings, slot, staging[slot])) { std::fprintf(stderr, "submit refused\n"); return 1; }
                }
            }
            std::fclose(fo);
            std::printf("ok %u frames with %u in flight\n", N, K);
        }
        // error path: an object pointing past the mesh table must come back as an exception, not a crash
        meta.objects[0].mesh = 99;
        meta.objects_dirty = true;
        try {
            plugin.node.run(view, globals, &settings, img);
            std::fprintf(stderr, "bad scene was accepted\n");
            return 1;
        } catch (const PtException& e) {
            if (e.code != PT_ERR_INVALID_SCENE) { std::fprintf(stderr, "unexpected code %d\n", e.code); return 1; }
        }
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
}

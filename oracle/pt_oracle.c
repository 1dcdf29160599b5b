/*
 * pt_oracle.c — CPU restatement of the reference's path-tracing shader.
 *
 * TEST INFRASTRUCTURE ONLY. This file is the checker that the HIP kernels are
 * compared against. Nothing in the product (the C-ABI library, its kernels, the
 * host layer) includes, links or calls it; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg do.
 *
 * PARITY UNPINNED: the reference (PixelDoted/path-tracing) ships no tests,
 * golden vectors or fixtures, and it cannot be built in this environment (Rust
 * nightly + Bevy 0.15.0-rc.2 + a Vulkan ICD are absent), so this restatement is
 * checked only by an independent numpy re-derivation of its integer/IEEE-exact
 * pieces (tests/naive_ref.py) and by review against the WGSL it cites. The Bevy
 * WGSL helpers the shader imports (bevy_pbr::lighting, bevy_render::maths,
 * fullscreen vertex shader) are not part of the reference tree; they are restated
 * here from their published definitions for bevy_pbr / bevy_render 0.15.0-rc.2.
 *
 * The loop nest is deliberately the reference's own: per pixel, `samples` x
 * `bounces`, and per ray a brute-force loop over all objects and all triangles
 * (src/query.wgsl:55-125) with the closest-so-far kept in a private global.
 *
 * Numerical policy (WGSL leaves fusion, dot/cross evaluation order and the
 * accuracy of sin/cos/pow/normalize to the implementation; this file fixes
 * one evaluation so that every discrete decision is reproducible):
 *   - IEEE-754 binary32 everywhere, round-to-nearest-even, subnormals kept;
 *     compiled with -ffp-contract=off so the ONLY fused operations are the
 *     explicit fmaf() calls below;
 *   - dot(a,b)   = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
 *   - cross(a,b) = ( fma(a.y,b.z, -(a.z*b.y)), fma(a.z,b.x, -(a.x*b.z)),
 *                    fma(a.x,b.y, -(a.y*b.x)) )
 *   - M*(v,0)    : r_i = fma(M[2][i],v.z, fma(M[1][i],v.y, M[0][i]*v.x));
 *     M*(v,1) adds M[3][i] last
 *   - a + b*s    = fma(b,s,a) component-wise; mix(a,b,t) = fma(b,t, a*(1-t))
 *   - reflect(I,N) = fma(-(2*dot(N,I)), N, I)
 *   - normalize(v) = v * inverseSqrt(dot(v,v)), inverseSqrt(d) by a fixed recipe:
 *     y = bits(0x5f375a86 - (bits(d) >> 1)); three times y = y * fma(-((0.5*d)*y), y, 1.5)
 *     (at most 1.92 ulp from exact for d in [2^-100, 2^100], inside the 2 ulp WGSL allows
 *     inverseSqrt; GPUs running the reference use a hardware approximation here)
 *   - min/max = fminf/fmaxf; u32(f) truncates and saturates (NaN -> 0)
 *   - sin/cos  : Cody-Waite reduction by pi/2 + degree-7/8 polynomials (below)
 *   - pow(x,5) = ((x*x)*(x*x))*x
 *   - reads past the end of texture_data return 0.0 (the reference relies on
 *     wgpu's robust buffer access, whose result is unspecified).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/pt_amd.h"

/* Alternative evaluations, built as separate libraries (oracle/Makefile) only to measure how far other legal evaluations
 * of the same WGSL are from the one the kernels implement:
 *   -DPTO_IEEE_NORMALIZE  normalize(v) = v * (1 / sqrt(dot(v,v))), both correctly rounded
 *   -DPTO_NO_FMA          no fused multiply-add anywhere: every fma(a,b,c) below is a*b rounded, then + c rounded
 *                         (what a software rasteriser without FMA contraction computes)
 *   -DPTO_LIBM_TRIG       sin/cos from the C library instead of the fixed polynomials */
#ifdef PTO_NO_FMA
#define PTO_FMA(a, b, c) ((a) * (b) + (c))
#else
#define PTO_FMA(a, b, c) fmaf((a), (b), (c))
#endif

/* reference src/math.wgsl:3-6 */
#define EPSILON 4.88e-4f
#define WGSL_INFINITY 10000000.0f
#define U32_MAX 4294967295u
#define T_MIN 0.0001f
/* bevy_render::maths::PI */
#define PI_F 3.14159265358979323846f

typedef struct { float x, y, z; } v3;

typedef struct PtoScene {
    const PtObject* objects;    uint32_t n_objects;
    const PtMesh* meshes;       uint32_t n_meshes;
    const uint32_t* indices;    uint32_t n_indices;
    const PtVertex* vertices;   uint32_t n_vertices;
    const PtMaterial* materials;uint32_t n_materials;
    const PtTexture* textures;  uint32_t n_textures;
    const float* texture_data;  uint32_t n_texture_data;
} PtoScene;

/* reference src/query.wgsl:45-52 (+ the triangle that produced the record) */
typedef struct {
    float t;
    v3 p, n;
    float u, v; /* uv */
    uint32_t tri; /* index of the last accepted triangle inside its mesh */
} HitRecord;

typedef struct {
    const PtoScene* sc;
    HitRecord hit_record;   /* WGSL `var<private> hit_record` */
    uint32_t rng_state[3];  /* WGSL `var<private> rng_state`  */
} Invocation;

/* ---- vector helpers (the numerical policy above) -------------------------- */
static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline float dot3(v3 a, v3 b) { return PTO_FMA(a.z, b.z, PTO_FMA(a.y, b.y, a.x * b.x)); }
static inline v3 cross3(v3 a, v3 b) {
    return V3(PTO_FMA(a.y, b.z, -(a.z * b.y)), PTO_FMA(a.z, b.x, -(a.x * b.z)), PTO_FMA(a.x, b.y, -(a.y * b.x)));
}
static inline v3 sub3(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 add3(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 mul3(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scale3(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 neg3(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline v3 madd3(v3 a, v3 b, float s) { /* a + b*s */
    return V3(PTO_FMA(b.x, s, a.x), PTO_FMA(b.y, s, a.y), PTO_FMA(b.z, s, a.z));
}
/* inverseSqrt by a fixed recipe (see the numerical policy above) */
static inline float rsqrt_fixed(float d) {
    union { float f; uint32_t u; } c;
    c.f = d;
    c.u = 0x5f375a86u - (c.u >> 1);
    float y = c.f;
    const float hd = 0.5f * d;
    y = y * PTO_FMA(-(hd * y), y, 1.5f);
    y = y * PTO_FMA(-(hd * y), y, 1.5f);
    y = y * PTO_FMA(-(hd * y), y, 1.5f);
    return y;
}
float pto_rsqrt(float d) { return rsqrt_fixed(d); }
static inline v3 normalize3(v3 a) {
#ifdef PTO_IEEE_NORMALIZE /* alternative evaluation, only to measure how far apart two legal evaluations of normalize() are */
    float r = 1.0f / sqrtf(dot3(a, a));
#else
    float r = rsqrt_fixed(dot3(a, a));
#endif
    return V3(a.x * r, a.y * r, a.z * r);
}
static inline v3 xform_dir(const float* m, v3 v) {
    return V3(PTO_FMA(m[8], v.z, PTO_FMA(m[4], v.y, m[0] * v.x)),
              PTO_FMA(m[9], v.z, PTO_FMA(m[5], v.y, m[1] * v.x)),
              PTO_FMA(m[10], v.z, PTO_FMA(m[6], v.y, m[2] * v.x)));
}
static inline v3 xform_point(const float* m, v3 v) {
    v3 r = xform_dir(m, v);
    return V3(r.x + m[12], r.y + m[13], r.z + m[14]);
}
/* WGSL u32(f32): truncate toward zero, saturate; NaN -> 0 */
static inline uint32_t f2u(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

/* sin and cos of one angle, fixed algorithm (see header). */
void pto_sincos(float x, float* s_out, float* c_out) {
    float kf = rintf(x * 0.636619772367581343f);
    int k = (fabsf(kf) < 1073741824.0f) ? (int)kf : 0;
    float y = PTO_FMA(-kf, 1.57079637050628662109375f, x);
    y = PTO_FMA(-kf, -4.37113900018624283e-8f, y);
    float z = y * y;
    float ps = PTO_FMA(PTO_FMA(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
    float s = PTO_FMA(ps * z, y, y);
    float pc = PTO_FMA(PTO_FMA(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
    float c = PTO_FMA(pc * z, z, PTO_FMA(-0.5f, z, 1.0f));
    switch (k & 3) {
        case 0: *s_out = s;  *c_out = c;  break;
        case 1: *s_out = c;  *c_out = -s; break;
        case 2: *s_out = -s; *c_out = -c; break;
        default:*s_out = -c; *c_out = s;  break;
    }
}

/* ---- RNG: reference src/raytrace.wgsl:60-99 -------------------------------- */

/* src/raytrace.wgsl:60-78 */
static inline void pcg3d(uint32_t* st, uint32_t out[3]) {
    uint32_t x = st[0] * 1664525u + 1013904223u;
    uint32_t y = st[1] * 1664525u + 1013904223u;
    uint32_t z = st[2] * 1664525u + 1013904223u;
    x += y * z; y += z * x; z += x * y;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16;
    x += y * z; y += z * x; z += x * y;
    st[0] = x; st[1] = y; st[2] = z;
    out[0] = x; out[1] = y; out[2] = z;
}
void pto_pcg3d(uint32_t state[3]) { uint32_t o[3]; pcg3d(state, o); }

/* src/raytrace.wgsl:80-82 : abs(fract(vec3<f32>(pcg3d()) / 3141.592653589793)) */
static inline void rand3(uint32_t* st, float out[3]) {
    uint32_t v[3];
    pcg3d(st, v);
    for (int i = 0; i < 3; ++i) {
        float q = (float)v[i] / 3141.592653589793f;
        out[i] = fabsf(q - floorf(q));
    }
}
void pto_rand(uint32_t state[3], float out[3]) { rand3(state, out); }

/* src/raytrace.wgsl:88-95 */
static inline v3 cosine_sample(Invocation* iv) {
    float rng[3];
    rand3(iv->rng_state, rng);
    float phi = 2.0f * PI_F * rng[0];
    float sqr_sin_theta = rng[1];
    float sin_theta = sqrtf(sqr_sin_theta);
    float cos_theta = sqrtf(1.0f - sqr_sin_theta);
    float s, c;
#ifdef PTO_LIBM_TRIG
    s = sinf(phi);
    c = cosf(phi);
#else
    pto_sincos(phi, &s, &c);
#endif
    return V3(sin_theta * c, sin_theta * s, cos_theta);
}

/* src/raytrace.wgsl:102-116 ; returns the three columns t, b, n */
static inline void hugues_moller(v3 n, v3* t_out, v3* b_out) {
    float ax = fabsf(n.x), ay = fabsf(n.y), az = fabsf(n.z);
    v3 t;
    if (ax <= ay && ax <= az) t = V3(0.0f, -n.z, n.y);
    else if (ay <= ax && ay <= az) t = V3(-n.z, 0.0f, n.x);
    else t = V3(-n.y, n.x, 0.0f);
    t = normalize3(t);
    *t_out = t;
    *b_out = normalize3(cross3(n, t));
}

/* ---- closest hit: reference src/query.wgsl:55-147 --------------------------- */

/* src/query.wgsl:127-147 */
static inline float hit_box(const float* bmin, const float* bmax, v3 pos, v3 dir) {
    float ix = 1.0f / dir.x, iy = 1.0f / dir.y, iz = 1.0f / dir.z;
    float tminx = (bmin[0] - pos.x) * ix, tminy = (bmin[1] - pos.y) * iy, tminz = (bmin[2] - pos.z) * iz;
    float tmaxx = (bmax[0] - pos.x) * ix, tmaxy = (bmax[1] - pos.y) * iy, tmaxz = (bmax[2] - pos.z) * iz;
    float t1x = fminf(tminx, tmaxx), t1y = fminf(tminy, tmaxy), t1z = fminf(tminz, tmaxz);
    float t2x = fmaxf(tminx, tmaxx), t2y = fmaxf(tminy, tmaxy), t2z = fmaxf(tminz, tmaxz);
    float dst_near = fmaxf(fmaxf(t1x, t1y), t1z);
    float dst_far = fminf(fminf(t2x, t2y), t2z);
    int hit = (dst_far >= dst_near) && (dst_far > 0.0f);
    if (hit) return (dst_near > 0.0f) ? dst_near : WGSL_INFINITY;
    return 0.0f;
}

/* src/query.wgsl:66-125 */
static int hit_mesh(Invocation* iv, uint32_t object_index, float t_min, v3 rpos, v3 rdir) {
    const PtoScene* sc = iv->sc;
    const PtObject* object = &sc->objects[object_index];
    const PtMesh* mesh = &sc->meshes[object->mesh];
    int hit = 0;

    /* :73-74 */
    v3 pos = xform_point(object->world_to_local, rpos);
    v3 dir = xform_dir(object->world_to_local, rdir);

    /* :77-80 */
    float t_aabb = hit_box(mesh->aabb_min, mesh->aabb_max, pos, dir);
    if (t_aabb < t_min) return 0;

    for (uint32_t i = 0; i < mesh->tri_count; ++i) {
        uint32_t i3 = i * 3u;
        uint32_t ai = sc->indices[mesh->ihead + i3];
        uint32_t bi = sc->indices[mesh->ihead + i3 + 1u];
        uint32_t ci = sc->indices[mesh->ihead + i3 + 2u];
        const PtVertex* va = &sc->vertices[mesh->vhead + ai];
        const PtVertex* vb = &sc->vertices[mesh->vhead + bi];
        const PtVertex* vc = &sc->vertices[mesh->vhead + ci];
        v3 pa = V3(va->position[0], va->position[1], va->position[2]);
        v3 pb = V3(vb->position[0], vb->position[1], vb->position[2]);
        v3 pc = V3(vc->position[0], vc->position[1], vc->position[2]);

        /* :95-107 Moller-Trumbore */
        v3 edge_ab = sub3(pb, pa);
        v3 edge_ac = sub3(pc, pa);
        v3 n = cross3(edge_ab, edge_ac);
        v3 ao = sub3(pos, pa);
        v3 dao = cross3(ao, dir);

        float det = -dot3(dir, n);            /* dot(-dir, n) */
        float inv_det = 1.0f / det;

        float t = dot3(ao, n) * inv_det;
        float u = dot3(edge_ac, dao) * inv_det;
        float v = (-dot3(edge_ab, dao)) * inv_det; /* dot(-edge_ab, dao) */
        float w = 1.0f - u - v;

        /* :109 */
        if (det < EPSILON || t < t_min || t > iv->hit_record.t || u < 0.0f || v < 0.0f || w < 0.0f) continue;

        /* :113-121 */
        v3 _p = madd3(pos, dir, t);
        v3 na = V3(va->normal[0], va->normal[1], va->normal[2]);
        v3 nb = V3(vb->normal[0], vb->normal[1], vb->normal[2]);
        v3 nc = V3(vc->normal[0], vc->normal[1], vc->normal[2]);
        v3 _n = V3(PTO_FMA(nc.x, v, PTO_FMA(nb.x, u, na.x * w)),
                   PTO_FMA(nc.y, v, PTO_FMA(nb.y, u, na.y * w)),
                   PTO_FMA(nc.z, v, PTO_FMA(nb.z, u, na.z * w)));
        float uvx = PTO_FMA(vc->uv[0], v, PTO_FMA(vb->uv[0], u, va->uv[0] * w));
        float uvy = PTO_FMA(vc->uv[1], v, PTO_FMA(vb->uv[1], u, va->uv[1] * w));

        iv->hit_record.t = t;
        iv->hit_record.p = xform_point(object->local_to_world, _p);
        iv->hit_record.n = normalize3(xform_dir(object->local_to_world, _n));
        iv->hit_record.u = uvx;
        iv->hit_record.v = uvy;
        iv->hit_record.tri = i;
        hit = 1;
    }
    return hit;
}

/* src/query.wgsl:55-64 */
static uint32_t hit_all(Invocation* iv, v3 pos, v3 dir) {
    uint32_t hit = U32_MAX;
    for (uint32_t o = 0; o < iv->sc->n_objects; ++o) {
        if (hit_mesh(iv, o, T_MIN, pos, dir)) hit = o;
    }
    return hit;
}

/* ---- textures: reference src/raytrace.wgsl:120-153 --------------------------- */
static inline float texel(const PtoScene* sc, uint32_t i) {
    return (i < sc->n_texture_data) ? sc->texture_data[i] : 0.0f;
}
static v3 sample_texture(const PtoScene* sc, uint32_t idx, float u, float v) {
    const PtTexture* tx = &sc->textures[idx];
    float x = u * (float)tx->width;
    float y = v * (float)tx->height;
    /* note: row stride is `height`, as in the reference (:124) */
    uint32_t i = tx->offset + (f2u(x) + f2u(y) * tx->height) * tx->format;
    switch (tx->format) {
        case 1u: { float r = texel(sc, i); return V3(r, r, r); }
        case 2u: return V3(texel(sc, i), texel(sc, i + 1u), 0.0f);
        case 3u: return V3(texel(sc, i), texel(sc, i + 1u), texel(sc, i + 2u));
        case 4u: {
            float a = texel(sc, i + 3u);
            return V3(texel(sc, i) * a, texel(sc, i + 1u) * a, texel(sc, i + 2u) * a);
        }
        default: return V3(1.0f, 1.0f, 1.0f);
    }
}

/* ---- BRDF: reference src/raytrace.wgsl:157-203 -------------------------------- */
static inline float saturate(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
static inline float pow5(float x) { float x2 = x * x; return (x2 * x2) * x; }
/* bevy_pbr::lighting::F_Schlick with f0 = 1: f0 + (f90 - f0) * pow(1 - u, 5) */
static inline float f_schlick_1(float f90, float u) { return PTO_FMA(f90 - 1.0f, pow5(1.0f - u), 1.0f); }

static void calculate_brdf(Invocation* iv, v3 ray_dir, const PtMaterial* material, v3* out_dir, v3* out_color) {
    const PtoScene* sc = iv->sc;
    v3 N = iv->hit_record.n;
    /* :158 */
    v3 tt, bb;
    hugues_moller(N, &tt, &bb);
    v3 cs = cosine_sample(iv);
    /* mat3x3(t,b,n) * cs = t*cs.x + b*cs.y + n*cs.z */
    v3 lam = V3(PTO_FMA(N.x, cs.z, PTO_FMA(bb.x, cs.y, tt.x * cs.x)),
                PTO_FMA(N.y, cs.z, PTO_FMA(bb.y, cs.y, tt.y * cs.x)),
                PTO_FMA(N.z, cs.z, PTO_FMA(bb.z, cs.y, tt.z * cs.x)));
    v3 lambertian_ray = normalize3(lam);
    /* :159 reflect(I,N) = I - 2*dot(N,I)*N */
    float k2 = 2.0f * dot3(N, ray_dir);
    v3 reflection_ray = normalize3(madd3(ray_dir, N, -k2));
    /* :160 mix(a,b,t) = a*(1-t) + b*t */
    float rough_in = material->roughness;
    float om = 1.0f - rough_in;
    v3 new_ray_dir = V3(PTO_FMA(lambertian_ray.x, rough_in, reflection_ray.x * om),
                        PTO_FMA(lambertian_ray.y, rough_in, reflection_ray.y * om),
                        PTO_FMA(lambertian_ray.z, rough_in, reflection_ray.z * om));

    /* :162-170 */
    v3 albedo = V3(material->albedo[0], material->albedo[1], material->albedo[2]);
    if (material->albedo_texture != U32_MAX)
        albedo = mul3(albedo, sample_texture(sc, material->albedo_texture, iv->hit_record.u, iv->hit_record.v));
    /* the metallic-roughness texture is sampled and discarded (:168-170) */

    /* :174-181 */
    v3 V = neg3(ray_dir);
    v3 L = new_ray_dir;
    float NdotV = fmaxf(dot3(N, V), 0.0001f);

    /* :189 perceptualRoughnessToRoughness: clamp(r, 0.089, 1.0)^2 */
    float cr = fminf(fmaxf(rough_in, 0.089f), 1.0f);
    float roughness = cr * cr;

    /* :196 derive_lighting_input(N,V,L) */
    v3 H = normalize3(add3(L, V));
    float NdotL = saturate(dot3(N, L));
    float LdotH = saturate(dot3(L, H));

    /* :199 Fd_Burley */
    float f90 = PTO_FMA((2.0f * roughness) * LdotH, LdotH, 0.5f);
    float light_scatter = f_schlick_1(f90, NdotL);
    float view_scatter = f_schlick_1(f90, NdotV);
    float fd = (light_scatter * view_scatter) * (1.0f / PI_F);

    v3 color = scale3(albedo, fd);
    *out_dir = new_ray_dir;
    *out_color = scale3(color, PI_F); /* :202 */
}

/* ---- entry: reference src/raytrace.wgsl:207-277 ------------------------------- */

typedef struct PtoAux {
    uint32_t* primary_object;  /* may be NULL */
    uint32_t* primary_flat_id;
    float* primary_t;
    uint32_t* ray_count;       /* hit_all calls per pixel */
} PtoAux;

static void fragment(const PtoScene* sc, const PtFrame* fr, const uint32_t* tri_base,
                     uint32_t px, uint32_t py, float out[4],
                     uint32_t* prim_obj, uint32_t* prim_flat, float* prim_t, uint32_t* n_rays) {
    Invocation iv;
    iv.sc = sc;
    memset(&iv.hit_record, 0, sizeof iv.hit_record);

    float W = (float)fr->width, H = (float)fr->height;
    /* fullscreen triangle interpolated at the pixel centre */
    float uvx = ((float)px + 0.5f) / W;
    float uvy = ((float)py + 0.5f) / H;

    /* :210 rng_setup(in.uv * view.viewport.zw * (globals.time + 1.0)) */
    float tp1 = fr->time + 1.0f;
    float sx = (uvx * W) * tp1, sy = (uvy * H) * tp1;
    iv.rng_state[0] = f2u(sx);
    iv.rng_state[1] = f2u(sy);
    iv.rng_state[2] = f2u(sx) ^ f2u(sy);

    /* :212 */
    float dx = uvx * 2.0f - 1.0f;
    float dy = (uvy * 2.0f - 1.0f) * -1.0f;

    /* :215-217 */
    const float* wfv = fr->world_from_view;
    const float* vfc = fr->view_from_clip;
    v3 initial_origin = V3(wfv[12], wfv[13], wfv[14]);
    v3 temp = V3(PTO_FMA(vfc[4], dy, vfc[0] * dx) + vfc[8] + vfc[12],
                 PTO_FMA(vfc[5], dy, vfc[1] * dx) + vfc[9] + vfc[13],
                 PTO_FMA(vfc[6], dy, vfc[2] * dx) + vfc[10] + vfc[14]);
    v3 initial_direction = xform_dir(wfv, normalize3(temp));

    uint32_t bounces = fr->settings.bounces, samples = fr->settings.samples;
    v3 sky = V3(fr->settings.sky_color[0], fr->settings.sky_color[1], fr->settings.sky_color[2]);

    *prim_obj = U32_MAX; *prim_flat = U32_MAX; *prim_t = 1000.0f; *n_rays = 0;
    int have_primary = 0;

    v3 pixel_color = V3(0, 0, 0);
    for (uint32_t sample = 0; sample < samples; ++sample) {
        v3 rpos = initial_origin, rdir = initial_direction;
        v3 ray_color = V3(1, 1, 1);
        v3 color = V3(0, 0, 0);

        for (uint32_t bounce = 0; bounce < bounces; ++bounce) {
            iv.hit_record.t = 1000.0f;
            uint32_t hit = hit_all(&iv, rpos, rdir);
            (*n_rays)++;
            if (!have_primary) {
                have_primary = 1;
                *prim_obj = hit;
                *prim_t = iv.hit_record.t;
                *prim_flat = (hit != U32_MAX) ? tri_base[hit] + iv.hit_record.tri : U32_MAX;
            }
            if (hit != U32_MAX) {
                const PtObject* object = &sc->objects[hit];
                const PtMaterial* material = &sc->materials[object->mat];

                /* :239-244 */
                v3 emissive = V3(material->emissive[0], material->emissive[1], material->emissive[2]);
                if (material->emissive_texture != U32_MAX)
                    emissive = sample_texture(sc, material->emissive_texture, iv.hit_record.u, iv.hit_record.v);
                color = V3(PTO_FMA(ray_color.x, emissive.x, color.x), PTO_FMA(ray_color.y, emissive.y, color.y),
                           PTO_FMA(ray_color.z, emissive.z, color.z));
                v3 alb = V3(material->albedo[0], material->albedo[1], material->albedo[2]);
                if (dot3(alb, alb) < EPSILON) break;

                /* :251-253 */
                if (material->normal_map_texture != U32_MAX)
                    iv.hit_record.n = mul3(iv.hit_record.n,
                        sample_texture(sc, material->normal_map_texture, iv.hit_record.u, iv.hit_record.v));

                /* :256-260 */
                v3 ndir, bcol;
                calculate_brdf(&iv, rdir, material, &ndir, &bcol);
                rdir = ndir;
                rpos = madd3(iv.hit_record.p, rdir, 0.001f);
                ray_color = mul3(ray_color, bcol);
            } else {
                color = V3(PTO_FMA(ray_color.x, sky.x, color.x), PTO_FMA(ray_color.y, sky.y, color.y),
                           PTO_FMA(ray_color.z, sky.z, color.z));
                break;
            }
            /* :266-269 */
            float p = fmaxf(ray_color.x, fmaxf(ray_color.y, ray_color.z));
            if (p < EPSILON) break;
        }
        pixel_color = add3(pixel_color, color);
    }
    float ns = (float)samples;
    out[0] = pixel_color.x / ns;
    out[1] = pixel_color.y / ns;
    out[2] = pixel_color.z / ns;
    out[3] = 1.0f;
}

static uint32_t* build_tri_base(const PtoScene* sc) {
    uint32_t* tb = (uint32_t*)malloc(sizeof(uint32_t) * (sc->n_objects ? sc->n_objects : 1));
    uint32_t acc = 0;
    for (uint32_t o = 0; o < sc->n_objects; ++o) {
        tb[o] = acc;
        acc += sc->meshes[sc->objects[o].mesh].tri_count;
    }
    return tb;
}

/* Same range checks as the product's upload path, so a bad fixture fails here too. */
int pto_validate(const PtoScene* sc) {
    for (uint32_t m = 0; m < sc->n_meshes; ++m) {
        const PtMesh* me = &sc->meshes[m];
        if ((uint64_t)me->ihead + 3ull * me->tri_count > sc->n_indices) return -2;
        for (uint32_t i = 0; i < 3u * me->tri_count; ++i)
            if ((uint64_t)me->vhead + sc->indices[me->ihead + i] >= sc->n_vertices) return -2;
    }
    for (uint32_t o = 0; o < sc->n_objects; ++o) {
        if (sc->objects[o].mesh >= sc->n_meshes) return -2;
        if (sc->objects[o].mat >= sc->n_materials) return -2;
    }
    for (uint32_t m = 0; m < sc->n_materials; ++m) {
        const PtMaterial* ma = &sc->materials[m];
        uint32_t t[4] = {ma->albedo_texture, ma->emissive_texture, ma->metallic_roughness_texture, ma->normal_map_texture};
        for (int k = 0; k < 4; ++k)
            if (t[k] != U32_MAX && t[k] >= sc->n_textures) return -2;
    }
    return 0;
}

/*
 * Render the pixel rectangle [x0,x1) x [y0,y1) of a fr->width x fr->height frame.
 * out_rgba and the aux arrays are (x1-x0)*(y1-y0) pixels, row-major.
 * n_threads <= 0 uses the OpenMP default.
 */
int pto_render(const PtoScene* sc, const PtFrame* fr, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
               float* out_rgba, const PtoAux* aux, int n_threads) {
    if (!sc || !fr || !out_rgba || x1 > fr->width || y1 > fr->height || x0 > x1 || y0 > y1) return -1;
    int rc = pto_validate(sc);
    if (rc) return rc;
    uint32_t* tri_base = build_tri_base(sc);
    uint32_t cw = x1 - x0;
    long rows = (long)(y1 - y0);
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    (void)n_threads;
#pragma omp parallel for schedule(dynamic, 1)
    for (long r = 0; r < rows; ++r) {
        uint32_t py = y0 + (uint32_t)r;
        for (uint32_t px = x0; px < x1; ++px) {
            size_t k = (size_t)r * cw + (px - x0);
            uint32_t po, pf, nr; float pt;
            fragment(sc, fr, tri_base, px, py, out_rgba + 4 * k, &po, &pf, &pt, &nr);
            if (aux) {
                if (aux->primary_object) aux->primary_object[k] = po;
                if (aux->primary_flat_id) aux->primary_flat_id[k] = pf;
                if (aux->primary_t) aux->primary_t[k] = pt;
                if (aux->ray_count) aux->ray_count[k] = nr;
            }
        }
    }
    free(tri_base);
    return 0;
}

/* hit_all for a list of rays, hit_record.t preset to 1000.0 (src/raytrace.wgsl:230). */
int pto_query(const PtoScene* sc, const PtRay* rays, uint32_t n_rays, PtHit* hits, int n_threads) {
    if (!sc || (!rays && n_rays) || (!hits && n_rays)) return -1;
    int rc = pto_validate(sc);
    if (rc) return rc;
    uint32_t* tri_base = build_tri_base(sc);
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    (void)n_threads;
#pragma omp parallel for schedule(dynamic, 64)
    for (long i = 0; i < (long)n_rays; ++i) {
        Invocation iv;
        iv.sc = sc;
        memset(&iv.hit_record, 0, sizeof iv.hit_record);
        iv.hit_record.t = 1000.0f;
        v3 pos = V3(rays[i].pos[0], rays[i].pos[1], rays[i].pos[2]);
        v3 dir = V3(rays[i].dir[0], rays[i].dir[1], rays[i].dir[2]);
        uint32_t hit = hit_all(&iv, pos, dir);
        PtHit* h = &hits[i];
        memset(h, 0, sizeof *h);
        h->object = hit;
        h->t = iv.hit_record.t;
        if (hit != U32_MAX) {
            h->triangle = iv.hit_record.tri;
            h->flat_id = tri_base[hit] + iv.hit_record.tri;
            h->p[0] = iv.hit_record.p.x; h->p[1] = iv.hit_record.p.y; h->p[2] = iv.hit_record.p.z;
            h->n[0] = iv.hit_record.n.x; h->n[1] = iv.hit_record.n.y; h->n[2] = iv.hit_record.n.z;
            h->uv[0] = iv.hit_record.u; h->uv[1] = iv.hit_record.v;
        } else {
            h->triangle = U32_MAX;
            h->flat_id = U32_MAX;
        }
    }
    free(tri_base);
    return 0;
}

int pto_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

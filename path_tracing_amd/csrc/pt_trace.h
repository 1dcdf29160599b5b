// pt_trace.h — device functions of the two stages of the hot path:
//   closest_hit_*, object_test, tri_test, hit_box_exact   replace hit_all / hit_mesh / hit_box (reference src/query.wgsl:55-147)
//   calculate_brdf, sample_texture                        replace the shading helpers          (reference src/raytrace.wgsl:88-203)
//
// Closest hit keeps the reference's per-candidate arithmetic exactly (ray moved to the
// object's local space with world_to_local, the quirky mesh-AABB gate, Moeller-Trumbore
// with the un-normalised back-face threshold) and only changes WHICH candidates are
// visited: objects through a TLAS and triangles through a per-mesh BLAS whose slab tests
// are padded so that they can only cull what the exact test would reject. The winner is
// chosen by an order-independent rule, (smaller t) or (equal t and later/earlier
// (object, triangle)), which reproduces the reference's sequential `t > hit_record.t`
// reject test for any visiting order.
#pragma once

#include "pt_scene.h"

struct Best {
    float t;
    uint32_t obj;
    uint32_t tri;   // the triangle: counted from the start of the triangle arrays (so that its attributes can be fetched
                    // without a look at the object record), except for scenes staged in LDS, where such a look costs
                    // nothing and the triangle is counted within its mesh (one register less through the triangle loops)
    float u, v;
};

struct HitInfo {
    uint32_t mat;   // material of the object that was hit (what shading needs of it), PT_NONE: nothing was hit
    v3 p, n;
    float u, v;  // hit_record.uv
};

#ifdef PT_EXEC_COUNTS
// debug build: how often the wave executes each major block (the first active lane counts)
#define PT_COUNT(cnt, k) do { if ((uint32_t)(threadIdx.x & 63) == (uint32_t)__builtin_ctzll(__ballot(1))) (cnt).exec[k]++; } while (0)
// the same, adding the number of lanes for which `cond` holds
#define PT_COUNT_LANES(cnt, k, cond) do { const uint32_t n_ = (uint32_t)__popcll(__ballot(cond)); if ((uint32_t)(threadIdx.x & 63) == (uint32_t)__builtin_ctzll(__ballot(1))) (cnt).exec[k] += n_; } while (0)
// wave cycles since `t0` (and restart the clock)
#define PT_COUNT_CYCLES(cnt, k, t0) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); if ((uint32_t)(threadIdx.x & 63) == (uint32_t)__builtin_ctzll(__ballot(1))) (cnt).exec[k] += (uint32_t)(now_ - (t0)); (t0) = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PT_COUNT(cnt, k) do { } while (0)
#define PT_COUNT_LANES(cnt, k, cond) do { } while (0)
#define PT_COUNT_CYCLES(cnt, k, t0) do { } while (0)
#endif
struct LaneCounters {
    uint32_t xforms, box_tests, tri_tests, shades, hits;
#ifdef PT_EXEC_COUNTS
    uint32_t exec[16];
#endif
    uint32_t passes;  // wave-uniform: object-test passes the wave made in its last closest_hit (a measure of how expensive its rays are)
};

// Per-lane traversal stack in LDS: entry [level][lane], so a wave's push/pop touches 64 consecutive banks.
struct LdsStack {
    uint32_t* base;  // &s_stack[wave][0][lane]
    PT_DEV void put(int level, uint32_t v) { base[level * 64] = v; }
    PT_DEV uint32_t get(int level) const { return base[level * 64]; }
    PT_DEV bool all_in_lds(int) const { return true; }
    PT_DEV void put_lds(int level, uint32_t v) { base[level * 64] = v; }
    PT_DEV uint32_t get_lds(int level) const { return base[level * 64]; }
};
// The same with only the first `lds_levels` entries in LDS and the rest in a column of global memory per lane: a walk
// rarely holds more than a dozen entries, but the trees allow several times that, and LDS per workgroup decides how many
// waves a compute unit holds. A lane reads back only what it wrote itself, so no ordering beyond program order is needed.
struct SpillStack {
    uint32_t* base;      // &s_stack[wave][0][lane]
    uint32_t* overflow;  // this lane's column: entry k at overflow[k * stride]
    int lds_levels;
    uint32_t stride;
    // (The address spaces are spelled out: given two branches that access memory through generic pointers, the compiler
    // merges them into one flat access through a chosen pointer, which takes a slot in the LDS and in the vector-memory
    // queue whichever memory the lane touches.)
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    typedef __attribute__((address_space(1))) uint32_t glb_u32;
    PT_DEV void put(int level, uint32_t v) {
        if (level < lds_levels) ((lds_u32*)base)[level * 64] = v;
        else ((glb_u32*)overflow)[(size_t)(level - lds_levels) * stride] = v;
    }
    PT_DEV uint32_t get(int level) const {
        if (level < lds_levels) return ((const lds_u32*)base)[level * 64];
        return ((const glb_u32*)overflow)[(size_t)(level - lds_levels) * stride];
    }
    // Wave-uniform: no lane of the wave touches a level at or above `top` that lies outside LDS, so the plain LDS
    // accessors below may be used (the usual case; the per-access choice above costs a dozen instructions each time).
    PT_DEV bool all_in_lds(int top) const { return !__any(top > lds_levels); }
    PT_DEV void put_lds(int level, uint32_t v) { ((lds_u32*)base)[level * 64] = v; }
    PT_DEV uint32_t get_lds(int level) const { return ((const lds_u32*)base)[level * 64]; }
};

PT_DEV bool candidate_wins(float t, uint32_t obj, uint32_t tri, const Best& b, uint32_t tie_lowest) {
    if (t < b.t) return true;
    if (t == b.t) {
        if (b.obj == PT_NONE) return true;
        if (!tie_lowest) return (obj > b.obj) || (obj == b.obj && tri > b.tri);
        return (obj < b.obj) || (obj == b.obj && tri < b.tri);
    }
    return false;
}

// reference src/query.wgsl:95-121, with edges and normal precomputed at upload by the same operations.
// Returns true when `best` was replaced.
template <bool STATS>
PT_DEV bool tri_test(const float4 q0, const float4 q1, const float4 q2, v3 o, v3 d, uint32_t obj, uint32_t tri,
                     uint32_t tie_lowest, Best& best, LaneCounters& cnt) {
    if (STATS) cnt.tri_tests++;
    v3 a = V3(q0.x, q0.y, q0.z);
    v3 e_ab = V3(q0.w, q1.x, q1.y);
    v3 e_ac = V3(q1.z, q1.w, q2.x);
    v3 n = V3(q2.y, q2.z, q2.w);
    v3 ao = sub3(o, a);
    v3 dao = cross3(ao, d);
    float det = -dot3(d, n);
    if (det < PT_EPSILON) return false;
    float un = dot3(e_ac, dao);
    float vn = -dot3(e_ab, dao);
    // Early out without the division: for det in [EPSILON, 2^60) the reciprocal is a positive normal number above
    // 2^-60, so a numerator below -2^-60 gives a product below -2^-120, which cannot round to zero: u (or v) < 0
    // and the reference rejects. Anything outside these ranges takes the exact path below.
    if ((un < -0x1p-60f || vn < -0x1p-60f) && det < 0x1p60f) return false;
    float inv_det = pt_div(1.0f, det);
    float t = dot3(ao, n) * inv_det;
    float u = un * inv_det;
    float v = vn * inv_det;
    float w = 1.0f - u - v;
    if (t < PT_T_MIN || u < 0.0f || v < 0.0f || w < 0.0f) return false;
    if (!candidate_wins(t, obj, tri, best, tie_lowest)) return false;
    best.t = t;
    best.obj = obj;
    best.tri = tri;
    best.u = u;
    best.v = v;
    return true;
}

// reference src/query.wgsl:127-147
PT_DEV float hit_box_exact(const float* bmin, const float* bmax, v3 pos, v3 dir) {
    v3 inv = V3(pt_div(1.0f, dir.x), pt_div(1.0f, dir.y), pt_div(1.0f, dir.z));
    float tminx = (bmin[0] - pos.x) * inv.x, tminy = (bmin[1] - pos.y) * inv.y, tminz = (bmin[2] - pos.z) * inv.z;
    float tmaxx = (bmax[0] - pos.x) * inv.x, tmaxy = (bmax[1] - pos.y) * inv.y, tmaxz = (bmax[2] - pos.z) * inv.z;
    float t1x = pt_min(tminx, tmaxx), t1y = pt_min(tminy, tmaxy), t1z = pt_min(tminz, tmaxz);
    float t2x = pt_max(tminx, tmaxx), t2y = pt_max(tminy, tmaxy), t2z = pt_max(tminz, tmaxz);
    float dst_near = pt_max(pt_max(t1x, t1y), t1z);
    float dst_far = pt_min(pt_min(t2x, t2y), t2z);
    bool hit = (dst_far >= dst_near) && (dst_far > 0.0f);
    if (hit) return (dst_near > 0.0f) ? dst_near : PT_WGSL_INFINITY;
    return 0.0f;
}

// Conservative ray/box test used only for culling. The box is grown by `pad` on every side (folded into ka/kb),
// the reciprocals come from v_rcp_f32 (1 ulp, far inside the pad) and are clamped to +-2^100 so that axis-parallel
// rays stay finite, NaNs drop out of min/max (= "no constraint"), and the interval is clipped to [0, tmax].
struct SlabRay {
    v3 inv;     // clamped ~1/dir
    v3 ka, kb;  // -(o+pad)*inv, -(o-pad)*inv
};
PT_DEV SlabRay make_slab_ray(v3 o, v3 d, float pad) {
    SlabRay s;
    const float big = 0x1p100f;
    s.inv = V3(pt_min(pt_max(__builtin_amdgcn_rcpf(d.x), -big), big), pt_min(pt_max(__builtin_amdgcn_rcpf(d.y), -big), big),
               pt_min(pt_max(__builtin_amdgcn_rcpf(d.z), -big), big));
    s.ka = V3(-(o.x + pad) * s.inv.x, -(o.y + pad) * s.inv.y, -(o.z + pad) * s.inv.z);
    s.kb = V3(-(o.x - pad) * s.inv.x, -(o.y - pad) * s.inv.y, -(o.z - pad) * s.inv.z);
    return s;
}
PT_DEV bool slab_test(const SlabRay& s, float4 lo, float4 hi, float tmax, float& tnear) {
    float ax = pt_fma(lo.x, s.inv.x, s.ka.x), bx = pt_fma(hi.x, s.inv.x, s.kb.x);
    float ay = pt_fma(lo.y, s.inv.y, s.ka.y), by = pt_fma(hi.y, s.inv.y, s.kb.y);
    float az = pt_fma(lo.z, s.inv.z, s.ka.z), bz = pt_fma(hi.z, s.inv.z, s.kb.z);
    float tn = pt_max(pt_max(pt_min(ax, bx), pt_min(ay, by)), pt_max(pt_min(az, bz), 0.0f));
    float tf = pt_min(pt_min(pt_max(ax, bx), pt_max(ay, by)), pt_min(pt_max(az, bz), tmax));
    tnear = tn;
    return tn <= tf;
}

// The same test against a box stored as (lo, hi, lo, 0) per axis (SceneView::obj_slab): the lane reads (near, far) of
// an axis as two consecutive words starting at lo when its reciprocal direction is non-negative and at hi when it is
// negative, so the six minimum/maximum operations that sort the plane distances of slab_test are not needed. For
// inv >= 0 the fused products satisfy a <= b (lo <= hi, pad >= 0, rounding is monotonic), for inv < 0 the reverse:
// the interval is the one slab_test computes, and so is every culling decision.
struct SlabRaySel {
    v3 inv, kn, kf;         // reciprocal direction; the constants of the near and of the far plane per axis
    uint32_t ox, oy, oz;    // word offsets into an object's 12 floats
};
PT_DEV SlabRaySel make_slab_ray_sel(const SlabRay& s) {
    SlabRaySel r;
    const bool nx = (__float_as_uint(s.inv.x) >> 31) != 0u, ny = (__float_as_uint(s.inv.y) >> 31) != 0u, nz = (__float_as_uint(s.inv.z) >> 31) != 0u;
    r.inv = s.inv;
    r.kn = V3(nx ? s.kb.x : s.ka.x, ny ? s.kb.y : s.ka.y, nz ? s.kb.z : s.ka.z);
    r.kf = V3(nx ? s.ka.x : s.kb.x, ny ? s.ka.y : s.kb.y, nz ? s.ka.z : s.kb.z);
    r.ox = nx ? 1u : 0u;
    r.oy = ny ? 5u : 4u;
    r.oz = nz ? 9u : 8u;
    return r;
}
PT_DEV bool slab_test_sel(const SlabRaySel& s, const float* box, float tmax, float& tnear) {
    const float* px = box + s.ox;
    const float* py = box + s.oy;
    const float* pz = box + s.oz;
    const float nx = pt_fma(px[0], s.inv.x, s.kn.x), fx = pt_fma(px[1], s.inv.x, s.kf.x);
    const float ny = pt_fma(py[0], s.inv.y, s.kn.y), fy = pt_fma(py[1], s.inv.y, s.kf.y);
    const float nz = pt_fma(pz[0], s.inv.z, s.kn.z), fz = pt_fma(pz[1], s.inv.z, s.kf.z);
    const float tn = pt_max(pt_max(nx, ny), pt_max(nz, 0.0f));
    const float tf = pt_min(pt_min(fx, fy), pt_min(fz, tmax));
    tnear = tn;
    return tn <= tf;
}

// ---- the 4-wide trees ---------------------------------------------------------------------------------------------
// Ray constants for the child-box tests of a WNode: per axis the reciprocal direction, the constants of the near and of
// the far plane (chosen by the sign of the direction, as in SlabRaySel) and that sign.
struct WideRay {
    v3 inv, kn, kf;
    bool nx, ny, nz;
};
PT_DEV WideRay make_wide_ray(const SlabRay& s) {
    WideRay r;
    r.nx = (__float_as_uint(s.inv.x) >> 31) != 0u;
    r.ny = (__float_as_uint(s.inv.y) >> 31) != 0u;
    r.nz = (__float_as_uint(s.inv.z) >> 31) != 0u;
    r.inv = s.inv;
    r.kn = V3(r.nx ? s.kb.x : s.ka.x, r.ny ? s.kb.y : s.ka.y, r.nz ? s.kb.z : s.ka.z);
    r.kf = V3(r.nx ? s.ka.x : s.kb.x, r.ny ? s.ka.y : s.kb.y, r.nz ? s.ka.z : s.kb.z);
    return r;
}

#define PT_WALK_DONE 0x7FFFFFFFu
// One visit of a wide node (its four 16-byte parts in q0..q3): tests the child boxes, continues with the child the ray
// enters first and pushes the other children it hits so that the nearer one is popped first. Returns the next node
// reference (a node, a leaf, or PT_WALK_DONE when nothing is hit and the stack is back at `sp0`).
// A plane distance is  (origin + q * step - ray_origin -+ pad) * inv  =  fma(q, step * inv, fma(origin, inv, k)):
// step is a power of two, so step * inv is exact, and the two fused operations stay well inside the pads (the ray's, in
// k, and the 2^-20 of the node's size its boxes were grown by when they were put on the grid).
// The three stack writes are unconditional: entries that are not pushed land just above the new top, so the stack has
// three words of headroom (stack_depth_needed() on the host side).
template <bool STATS, class Stk>
PT_DEV uint32_t wide_visit(const uint4 q0, const uint4 q1, const uint4 q2, const uint4 q3, const WideRay& wr, float tmax,
                           Stk stk, int& sp, int sp0, LaneCounters& cnt) {
    const uint32_t exps = q0.w;
    const float sx = __uint_as_float((exps << 23) & 0x7F800000u) * wr.inv.x;
    const float sy = __uint_as_float((exps << 15) & 0x7F800000u) * wr.inv.y;
    const float sz = __uint_as_float((exps << 7) & 0x7F800000u) * wr.inv.z;
    const float ox = __uint_as_float(q0.x), oy = __uint_as_float(q0.y), oz = __uint_as_float(q0.z);
    const float bnx = pt_fma(ox, wr.inv.x, wr.kn.x), bfx = pt_fma(ox, wr.inv.x, wr.kf.x);
    const float bny = pt_fma(oy, wr.inv.y, wr.kn.y), bfy = pt_fma(oy, wr.inv.y, wr.kf.y);
    const float bnz = pt_fma(oz, wr.inv.z, wr.kn.z), bfz = pt_fma(oz, wr.inv.z, wr.kf.z);
    // q2 = (qlo.x, qlo.y, qlo.z, qhi.x), q3 = (qhi.y, qhi.z, -, -)
    const uint32_t wnx = wr.nx ? q2.w : q2.x, wfx = wr.nx ? q2.x : q2.w;
    const uint32_t wny = wr.ny ? q3.x : q2.y, wfy = wr.ny ? q2.y : q3.x;
    const uint32_t wnz = wr.nz ? q3.y : q2.z, wfz = wr.nz ? q2.z : q3.y;
    uint32_t key[4], ref[4] = {q1.x, q1.y, q1.z, q1.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float nx = pt_fma((float)((wnx >> (8 * k)) & 0xFFu), sx, bnx), fx = pt_fma((float)((wfx >> (8 * k)) & 0xFFu), sx, bfx);
        const float ny = pt_fma((float)((wny >> (8 * k)) & 0xFFu), sy, bny), fy = pt_fma((float)((wfy >> (8 * k)) & 0xFFu), sy, bfy);
        const float nz = pt_fma((float)((wnz >> (8 * k)) & 0xFFu), sz, bnz), fz = pt_fma((float)((wfz >> (8 * k)) & 0xFFu), sz, bfz);
        const float tn = pt_max(pt_max(nx, ny), pt_max(nz, 0.0f));
        const float tf = pt_min(pt_min(fx, fy), pt_min(fz, tmax));
        const bool hit = (ref[k] != PT_NONE) && (tn <= tf);
        key[k] = hit ? __float_as_uint(tn) : 0xFFFFFFFFu;   // tn >= 0: its bit pattern orders like its value
        if (STATS && ref[k] != PT_NONE) cnt.box_tests++;
    }
    // sort the four (entry distance, child) pairs, nearest first
#define PT_CE(a, b) { const bool sw_ = key[b] < key[a]; const uint32_t ka_ = sw_ ? key[b] : key[a], kb_ = sw_ ? key[a] : key[b]; \
                      const uint32_t ra_ = sw_ ? ref[b] : ref[a], rb_ = sw_ ? ref[a] : ref[b]; key[a] = ka_; key[b] = kb_; ref[a] = ra_; ref[b] = rb_; }
    PT_CE(0, 1) PT_CE(2, 3) PT_CE(0, 2) PT_CE(1, 3) PT_CE(1, 2)
#undef PT_CE
    const int n_hit = (key[0] != 0xFFFFFFFFu) + (key[1] != 0xFFFFFFFFu) + (key[2] != 0xFFFFFFFFu) + (key[3] != 0xFFFFFFFFu);
    const int n_push = n_hit > 1 ? n_hit - 1 : 0;
    // (levels sp - 1 .. sp + 5 are touched below)
    if (stk.all_in_lds(sp + 6)) {
#pragma unroll
        for (int j = 1; j < 4; ++j) stk.put_lds(sp + ((key[j] != 0xFFFFFFFFu) ? n_push - j : n_push + j - 1), ref[j]);
        sp += n_push;
        if (n_hit) return ref[0];
        return (sp > sp0) ? stk.get_lds(--sp) : PT_WALK_DONE;
    }
#pragma unroll
    for (int j = 1; j < 4; ++j) stk.put(sp + ((key[j] != 0xFFFFFFFFu) ? n_push - j : n_push + j - 1), ref[j]);
    sp += n_push;
    if (n_hit) return ref[0];
    return (sp > sp0) ? stk.get(--sp) : PT_WALK_DONE;
}

// Per-lane stack walk of a wide tree in "while-while" form: every lane first descends through nodes until it holds a
// leaf (or runs out of work), then the lanes that hold a leaf process it together. This keeps the expensive leaf code
// (object or triangle test) running with as many lanes as possible. The descent is cut after BURST nodes: with
// incoherent rays a few lanes on long descents would otherwise hold back every lane that already has its leaf.
#ifndef PT_BURST_BLAS
#define PT_BURST_BLAS 1
#endif
#ifndef PT_BURST_TLAS
#define PT_BURST_TLAS 8
#endif
// `fetch(ref, q0, q1, q2, q3)` reads the four 16-byte parts of a node. `tmax` is read through the
// reference every time, so leaves that shorten the ray tighten the culling at once. `leaf(ref, sp)` gets the child
// reference without PT_LEAF_BIT.
PT_DEV void node_parts(const WNode* n, uint4& q0, uint4& q1, uint4& q2, uint4& q3) {
    const uint4* N = reinterpret_cast<const uint4*>(n);
    q0 = N[0]; q1 = N[1]; q2 = N[2]; q3 = N[3];
}
template <bool STATS, int BURST, class NodeFn, class LeafFn, class Stk>
PT_DEV void wide_walk(NodeFn fetch, uint32_t root, const WideRay& wr, const float& tmax, Stk stk, int sp0, bool active,
                      LaneCounters& cnt, LeafFn leaf) {
    uint32_t node = active ? root : PT_WALK_DONE;
    int sp = sp0;
    while (__any(node != PT_WALK_DONE)) {
        for (int burst = 0; burst < BURST && node != PT_WALK_DONE && !(node & PT_LEAF_BIT); ++burst) {
            uint4 q0, q1, q2, q3;
            fetch(node, q0, q1, q2, q3);
            node = wide_visit<STATS, Stk>(q0, q1, q2, q3, wr, tmax, stk, sp, sp0, cnt);
        }
        if (node != PT_WALK_DONE && (node & PT_LEAF_BIT)) {
            leaf(node & ~PT_LEAF_BIT, sp);
            node = (sp > sp0) ? stk.get(--sp) : PT_WALK_DONE;
        }
    }
}

// The same walk in resumable form, for kernels that keep a lane's walk alive across rounds of their main loop: the
// state is (current node, stack height); the stack itself stays in the lane's LDS column. The call returns when
// every lane is done or when at least `quota` of the lanes that were walking at entry have finished, so that those
// can go on (shade, take a new ray) instead of idling until the longest walk of the wave ends.
struct WalkState {
    uint32_t node;  // PT_WALK_DONE: no walk in progress
    int sp;
};
template <bool STATS, int BURST, class NodeFn, class LeafFn, class Stk>
PT_DEV void wide_walk_resume(NodeFn fetch, const WideRay& wr, const float& tmax, Stk stk, WalkState& ws, uint32_t quota,
                             LaneCounters& cnt, LeafFn leaf) {
    uint32_t node = ws.node;
    int sp = ws.sp;
    const uint32_t walking = (uint32_t)__popcll(__ballot(node != PT_WALK_DONE));
    const uint32_t stop_at = walking > quota ? walking - quota : 0u;   // leave the loop when this many (or fewer) still walk
    for (;;) {
        const uint32_t still = (uint32_t)__popcll(__ballot(node != PT_WALK_DONE));
        if (still <= stop_at) break;
        for (int burst = 0; burst < BURST && node != PT_WALK_DONE && !(node & PT_LEAF_BIT); ++burst) {
            PT_COUNT(cnt, 8); PT_COUNT_LANES(cnt, 13, true);
#ifdef PT_EXEC_COUNTS
            // (debug build: node-visit iterations that run with at most 16 and at most 32 lanes)
            { const uint32_t al_ = (uint32_t)__popcll(__ballot(1)); if (al_ <= 16u) PT_COUNT(cnt, 7); if (al_ <= 32u) PT_COUNT(cnt, 11); }
#endif
            uint4 q0, q1, q2, q3;
            fetch(node, q0, q1, q2, q3);
            node = wide_visit<STATS, Stk>(q0, q1, q2, q3, wr, tmax, stk, sp, 0, cnt);
        }
        if (node != PT_WALK_DONE && (node & PT_LEAF_BIT)) {
            PT_COUNT(cnt, 9); PT_COUNT_LANES(cnt, 14, true);
            leaf(node & ~PT_LEAF_BIT, sp);
            node = (sp > 0) ? stk.get(--sp) : PT_WALK_DONE;
        }
    }
    ws.node = node;
    ws.sp = sp;
}

// hit_mesh for one object (reference src/query.wgsl:66-125). Each lane brings its own object; lanes whose objects
// share a mesh are processed together so that the mesh and triangle records come through the scalar cache.
// The reference gates the whole mesh on hit_box() BEFORE looking at triangles; the gate only ever rejects, so it
// is evaluated afterwards and only for objects that produced a winning candidate (the exact arithmetic with its
// three divisions is then needed once per accepted object instead of once per visited object).
// Meshes without a BLAS and more than this many triangles are first tested against their padded local box.
#define PT_LOCAL_BOX_MIN_TRIS 4u
template <bool STATS, bool PERLANE, bool LB = false, class Stk>
PT_DEV void object_test(const SceneView& sc, uint32_t obj, v3 ro, v3 rd, Best& best, Stk stk, int sp0,
                        LaneCounters& cnt) {
    const DObject* ob = &sc.objects[obj];
    const Affine w2l = ob->w2l;
    const uint32_t mesh_id = ob->mesh;
    const v3 lo = xform_point(w2l, ro);   // reference :73
    const v3 ld = xform_dir(w2l, rd);     // reference :74
    if (STATS) cnt.xforms++;
    Best cand = best;
    bool changed = false;
    if (PERLANE) {
        // scene records live in LDS: every lane reads its own mesh and triangles, no grouping by mesh needed
        const DMesh* me = &sc.meshes[mesh_id];
        const uint32_t root = me->blas_root, tri_first = me->tri_first;
        // (the culling ray is only built where it is used: three reciprocals and a dozen operations that the
        // two-triangle meshes of a Cornell box never need)
        const float pad = 0x1p-18f * (pt_max(pt_max(__builtin_fabsf(lo.x), __builtin_fabsf(lo.y)), __builtin_fabsf(lo.z)) + me->max_abs);
        if (root == PT_NONE) {
            // no BLAS: the padded mesh box stands in for its root (the same test a BLAS walk starts with)
            const uint32_t n = me->tri_count;
            if (STATS) cnt.box_tests++;
            if (n > PT_LOCAL_BOX_MIN_TRIS) {
                float tn;
                const SlabRay sr = make_slab_ray(lo, ld, pad);
                if (!slab_test(sr, make_float4(me->bmin[0], me->bmin[1], me->bmin[2], 0.0f), make_float4(me->bmax[0], me->bmax[1], me->bmax[2], 0.0f), best.t, tn))
                    return;
            }
            for (uint32_t i = 0; i < n; ++i) {   // (scenes staged in LDS number a hit's triangle within its mesh: see Best)
                PT_COUNT(cnt, 10);
                const DTri* T = &sc.tris[tri_first + i];
                changed |= tri_test<STATS>(T->q0, T->q1, T->q2, lo, ld, obj, i, sc.tie_lowest, best, cnt);
            }
        } else {
            const WideRay wr = make_wide_ray(make_slab_ray(lo, ld, pad));
            const WNode* nodes = sc.blas_nodes;
            const DTri* tris = sc.tris + tri_first;
            const uint32_t* ids = sc.tri_ids + tri_first;
            wide_walk<STATS, PT_BURST_BLAS>([&](uint32_t r, uint4& q0, uint4& q1, uint4& q2, uint4& q3) { node_parts(nodes + r, q0, q1, q2, q3); }, root, wr, best.t, stk, sp0, true, cnt,
                                            [&](uint32_t leaf, int) {
                const uint32_t pos = leaf & PT_WLEAF_POS_MASK, n = (leaf >> PT_WLEAF_COUNT_SHIFT) + 1u;
                // (one triangle at a time here: the records are in LDS, and two in flight cost this variant registers)
                for (uint32_t j = 0; j < n; ++j) {
                    const DTri* T = &tris[pos + j];
                    changed |= tri_test<STATS>(T->q0, T->q1, T->q2, lo, ld, obj, ids[pos + j], sc.tie_lowest, best, cnt);
                }
            });
        }
        if (changed) {
            PT_COUNT(cnt, 11);
            const float t_aabb = hit_box_exact(me->bmin, me->bmax, lo, ld);   // reference :77-80
            if (t_aabb < PT_T_MIN) best = cand;   // the gate rejects the object: back to the closest hit before it (rare)
        }
        return;
    }
    bool pending = true;
    while (__any(pending)) {
        if (pending) {
            const uint32_t m = __builtin_amdgcn_readfirstlane(mesh_id);
            // Compare against a laundered copy: with `mesh_id == m` the compiler would substitute the per-lane id for
            // the uniform one inside the branch and fetch the mesh and triangle records with vector loads again.
            uint32_t m_cmp = m;
            asm volatile("" : "+s"(m_cmp));
            if (mesh_id == m_cmp) {
                pending = false;
                const DMesh* me = &sc.meshes[m];
                const uint32_t root = me->blas_root, tri_first = me->tri_first;
                const float pad = 0x1p-18f * (pt_max(pt_max(__builtin_fabsf(lo.x), __builtin_fabsf(lo.y)), __builtin_fabsf(lo.z)) + me->max_abs);
                const SlabRay sr = make_slab_ray(lo, ld, pad);
                if (root == PT_NONE) {
                    const uint32_t n = me->tri_count;
                    float tn;
                    if (STATS) cnt.box_tests++;
                    const bool inside = n <= PT_LOCAL_BOX_MIN_TRIS ||
                        slab_test(sr, make_float4(me->bmin[0], me->bmin[1], me->bmin[2], 0.0f), make_float4(me->bmax[0], me->bmax[1], me->bmax[2], 0.0f), cand.t, tn);
                    if (__any(inside)) {
                        // two triangles per step: their records (scalar loads, the mesh being the same for all lanes here)
                        // are on their way together (four per step cost more scalar registers than the waiting they save)
                        const uint32_t g_end = tri_first + n;
                        for (uint32_t g = tri_first; g < g_end; g += 2u) {
                            PT_COUNT(cnt, 10); PT_COUNT_LANES(cnt, 15, inside);
                            const bool two = g + 1u < g_end;
                            const DTri A = sc.tris[g], B = sc.tris[two ? g + 1u : g];
                            if (inside) {
                                changed |= tri_test<STATS>(A.q0, A.q1, A.q2, lo, ld, obj, g, sc.tie_lowest, cand, cnt);
                                if (two) changed |= tri_test<STATS>(B.q0, B.q1, B.q2, lo, ld, obj, g + 1u, sc.tie_lowest, cand, cnt);
                            }
                        }
                    }
                } else {
                    const WideRay wr = make_wide_ray(sr);
                    auto walk = [&](const WNode* nodes, const DTri* tris, const uint32_t* ids) {
                        wide_walk<STATS, PT_BURST_BLAS>([&](uint32_t r, uint4& q0, uint4& q1, uint4& q2, uint4& q3) { node_parts(nodes + r, q0, q1, q2, q3); }, root, wr, cand.t, stk, sp0, true, cnt,
                                                        [&](uint32_t leaf, int) {
                            const uint32_t pos = leaf & PT_WLEAF_POS_MASK, n = (leaf >> PT_WLEAF_COUNT_SHIFT) + 1u;
                            // (the first two triangles of a leaf are fetched together: one trip to memory instead of two)
                            const uint32_t p1 = pos + (n > 1u ? 1u : 0u);
                            const DTri A = tris[pos], B = tris[p1];
                            const uint32_t ia = ids[pos], ib = ids[p1];
                            changed |= tri_test<STATS>(A.q0, A.q1, A.q2, lo, ld, obj, tri_first + ia, sc.tie_lowest, cand, cnt);
                            if (n > 1u) changed |= tri_test<STATS>(B.q0, B.q1, B.q2, lo, ld, obj, tri_first + ib, sc.tie_lowest, cand, cnt);
                            for (uint32_t j = 2; j < n; ++j) {
                                const DTri* T = &tris[pos + j];
                                changed |= tri_test<STATS>(T->q0, T->q1, T->q2, lo, ld, obj, tri_first + ids[pos + j], sc.tie_lowest, cand, cnt);
                            }
                        });
                    };
                    // (separate calls, so that each one's loads are plain LDS or plain global-memory loads)
                    if (LB && m == sc.lb_mesh) {
                        // this mesh's BLAS sits in LDS: a node visit costs an LDS read instead of a trip to the L2
                        if (sc.lb_tris) walk(sc.lb_nodes - sc.lb_node_base, sc.lb_tris, sc.lb_tri_ids);
                        else walk(sc.lb_nodes - sc.lb_node_base, sc.tris + tri_first, sc.tri_ids + tri_first);
                    } else {
                        walk(sc.blas_nodes, sc.tris + tri_first, sc.tri_ids + tri_first);
                    }
                }
                if (changed) {
                    const float t_aabb = hit_box_exact(me->bmin, me->bmax, lo, ld);   // reference :77-80
                    if (!(t_aabb < PT_T_MIN)) best = cand;
                }
            }
        }
    }
}

PT_DEV SlabRay world_slab_ray(v3 ro, v3 rd) {
    const float pad = 0x1p-16f * pt_max(pt_max(__builtin_fabsf(ro.x), __builtin_fabsf(ro.y)), __builtin_fabsf(ro.z));
    return make_slab_ray(ro, rd, pad);
}

// hit_all without a TLAS: the wave tests the padded world boxes of 32 objects at a time together (scalar loads),
// each lane collects its own candidates in a bit mask and then works through them.
// An object whose world box the ray misses cannot contain an accepted triangle, so skipping it changes nothing.
// With sc.cull_boxes == 0 (PT_ACCEL_BRUTE) every object is a candidate: the reference's exhaustive loop.
template <bool STATS, bool PERLANE, bool LB = false, class Stk>
PT_DEV void closest_hit_list(const SceneView& sc, v3 ro, v3 rd, bool active, Best& best, Stk stk, LaneCounters& cnt) {
#ifdef PT_EXEC_COUNTS
    unsigned long long tc = __builtin_amdgcn_s_memtime();
#endif
    const SlabRay sr = world_slab_ray(ro, rd);
    const uint32_t n = sc.n_objects;
    const bool cull = PERLANE || sc.cull_boxes != 0;
    SlabRaySel ss;
    if (PERLANE) ss = make_slab_ray_sel(sr);
    // (scenes staged in LDS have at most 32 objects: one group, and culling is always on)
    const uint32_t n_groups = PERLANE ? (n ? 1u : 0u) : (n + 31u) / 32u;
    for (uint32_t g = 0; g < n_groups; ++g) {
        const uint32_t base = PERLANE ? 0u : g * 32u;
        const uint32_t m = PERLANE ? n : ((n - base < 32u) ? n - base : 32u);
        uint32_t mask = 0;
        uint32_t near_k = 0;      // the candidate whose box the ray enters first is tested first: it usually holds the
        float near_t = PT_T_MAX;  // closest hit, and the others are then culled by distance
        if (cull) {
            for (uint32_t k = 0; k < m; ++k) {
                PT_COUNT(cnt, 8);
                float tn;
                if (STATS) cnt.box_tests++;
                if (PERLANE ? slab_test_sel(ss, sc.obj_slab + 12u * (base + k), best.t, tn)
                            : slab_test(sr, sc.obj_lo[base + k], sc.obj_hi[base + k], best.t, tn)) {
                    mask |= 1u << k;
                    if (tn < near_t) { near_t = tn; near_k = k; }
                }
            }
        } else {
            mask = (m == 32u) ? 0xFFFFFFFFu : ((1u << m) - 1u);
        }
        if (!active) mask = 0;
        PT_COUNT_CYCLES(cnt, 13, tc);   // list scenes: cycles of the box loop
        // Every lane that has candidates takes part in the first pass, with the candidate it enters first; that one
        // needs no second look at its box (nothing has shortened the ray since the box loop).
        bool first = cull;   // wave-uniform
        while (__any(mask != 0)) {
            cnt.passes++;
            PT_COUNT(cnt, 9);
            if (mask != 0) {
                const uint32_t k = first ? near_k : (uint32_t)__builtin_ctz(mask);
                const uint32_t o = base + k;
                mask &= ~(1u << k);
                float tn;
                if (!cull || first || (PERLANE ? slab_test_sel(ss, sc.obj_slab + 12u * o, best.t, tn) : slab_test(sr, sc.obj_lo[o], sc.obj_hi[o], best.t, tn))) object_test<STATS, PERLANE, LB>(sc, o, ro, rd, best, stk, 0, cnt);
            }
            first = false;
        }
        PT_COUNT_CYCLES(cnt, 14, tc);   // list scenes: cycles of the candidate passes
    }
}

// ---- hit_all for 16 rays per wave: four lanes to a ray ----------------------------------------------------------------
// When a wave holds few rays (k_paths hands the most expensive pixels out in portions of 16), lanes 4j .. 4j+3 (a quad)
// work on the ray of lane 4j together: each lane tests one child box of a BLAS node, one triangle of a leaf, every
// fourth triangle of a mesh without a BLAS. A node visit is then ~70 instructions per lane instead of ~165, and the
// pixels handed out this way are the frame's longest chains of dependent steps. The candidates, their arithmetic and the
// winner rule are those of the one-lane-per-ray code, so the result is the same closest hit.
// All 64 lanes are active in these functions; ray, closest hit and control flow are identical in the lanes of a quad,
// only what is marked "own" differs. Cross-lane moves are DPP quad permutations (no LDS traffic).
template <int CTRL>
PT_DEV uint32_t quad_perm_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xF, 0xF, true); }
template <int CTRL>
PT_DEV float quad_perm_f32(float v) { return __uint_as_float(quad_perm_u32<CTRL>(__float_as_uint(v))); }
#define PT_QUAD_BCAST0 0x00   // [0,0,0,0]
#define PT_QUAD_XOR1 0xB1     // [1,0,3,2]
#define PT_QUAD_XOR2 0x4E     // [2,3,0,1]
#define PT_QUAD_ROT1 0x39     // [1,2,3,0]
#define PT_QUAD_ROT2 0x4E     // [2,3,0,1]
#define PT_QUAD_ROT3 0x93     // [3,0,1,2]
PT_DEV float quad_min_f32(float v) {
    v = pt_min(v, quad_perm_f32<PT_QUAD_XOR1>(v));
    return pt_min(v, quad_perm_f32<PT_QUAD_XOR2>(v));
}
PT_DEV uint32_t quad_max_u32(uint32_t v) {
    uint32_t w = quad_perm_u32<PT_QUAD_XOR1>(v);
    v = v > w ? v : w;
    w = quad_perm_u32<PT_QUAD_XOR2>(v);
    return v > w ? v : w;
}
// The better of the quad's four closest-hit candidates (same object), in every lane.
PT_DEV void quad_best(Best& b, uint32_t tie_lowest) {
#define PT_QUAD_MERGE(CTRL) { \
        const float pt_ = quad_perm_f32<CTRL>(b.t), pu_ = quad_perm_f32<CTRL>(b.u), pv_ = quad_perm_f32<CTRL>(b.v); \
        const uint32_t po_ = quad_perm_u32<CTRL>(b.obj), pr_ = quad_perm_u32<CTRL>(b.tri); \
        if (po_ != PT_NONE && candidate_wins(pt_, po_, pr_, b, tie_lowest)) { b.t = pt_; b.obj = po_; b.tri = pr_; b.u = pu_; b.v = pv_; } }
    PT_QUAD_MERGE(PT_QUAD_XOR1)
    PT_QUAD_MERGE(PT_QUAD_XOR2)
#undef PT_QUAD_MERGE
}
// One visit of a wide node by a quad; `q` is the lane's number in its quad, its own child is child q. The stack column is
// the quad leader's. Same contract as wide_visit().
template <bool STATS, class Stk>
PT_DEV uint32_t quad_visit(const uint4* N, uint32_t q, const WideRay& wr, float tmax, Stk stk, int& sp, int sp0, LaneCounters& cnt) {
    const uint4 q0 = N[0], q2 = N[2], q3 = N[3];
    const uint32_t ref = reinterpret_cast<const uint32_t*>(N)[4u + q];   // own child
    const uint32_t exps = q0.w;
    const float sx = __uint_as_float((exps << 23) & 0x7F800000u) * wr.inv.x;
    const float sy = __uint_as_float((exps << 15) & 0x7F800000u) * wr.inv.y;
    const float sz = __uint_as_float((exps << 7) & 0x7F800000u) * wr.inv.z;
    const float ox = __uint_as_float(q0.x), oy = __uint_as_float(q0.y), oz = __uint_as_float(q0.z);
    const uint32_t sh = 8u * q;
    const uint32_t wnx = wr.nx ? q2.w : q2.x, wfx = wr.nx ? q2.x : q2.w;
    const uint32_t wny = wr.ny ? q3.x : q2.y, wfy = wr.ny ? q2.y : q3.x;
    const uint32_t wnz = wr.nz ? q3.y : q2.z, wfz = wr.nz ? q2.z : q3.y;
    const float nx = pt_fma((float)((wnx >> sh) & 0xFFu), sx, pt_fma(ox, wr.inv.x, wr.kn.x)), fx = pt_fma((float)((wfx >> sh) & 0xFFu), sx, pt_fma(ox, wr.inv.x, wr.kf.x));
    const float ny = pt_fma((float)((wny >> sh) & 0xFFu), sy, pt_fma(oy, wr.inv.y, wr.kn.y)), fy = pt_fma((float)((wfy >> sh) & 0xFFu), sy, pt_fma(oy, wr.inv.y, wr.kf.y));
    const float nz = pt_fma((float)((wnz >> sh) & 0xFFu), sz, pt_fma(oz, wr.inv.z, wr.kn.z)), fz = pt_fma((float)((wfz >> sh) & 0xFFu), sz, pt_fma(oz, wr.inv.z, wr.kf.z));
    const float tn = pt_max(pt_max(nx, ny), pt_max(nz, 0.0f));
    const float tf = pt_min(pt_min(fx, fy), pt_min(fz, tmax));
    const bool hit = (ref != PT_NONE) && (tn <= tf);
    if (STATS && ref != PT_NONE) cnt.box_tests++;
    // own key: entry distance with the quad lane in the two lowest bits (distinct keys, order = order of wide_visit's sort
    // up to those two bits); 0xFFFFFFFF for a miss
    const uint32_t key = hit ? ((__float_as_uint(tn) & ~3u) | q) : 0xFFFFFFFFu;
    const uint32_t k1 = quad_perm_u32<PT_QUAD_ROT1>(key), k2 = quad_perm_u32<PT_QUAD_ROT2>(key), k3 = quad_perm_u32<PT_QUAD_ROT3>(key);
    const int rank = (int)(k1 < key) + (int)(k2 < key) + (int)(k3 < key);   // children the ray enters before the own one
    const int n_hit = (int)hit + (int)(k1 != 0xFFFFFFFFu) + (int)(k2 != 0xFFFFFFFFu) + (int)(k3 != 0xFFFFFFFFu);
    const int n_push = n_hit > 1 ? n_hit - 1 : 0;
    if (hit && rank > 0) stk.put(sp + n_push - rank, ref);   // the farthest goes deepest
    sp += n_push;
    const uint32_t first = quad_max_u32((hit && rank == 0) ? ref + 1u : 0u);   // (+1: reference 0 is a node)
    if (n_hit) return first - 1u;
    return (sp > sp0) ? stk.get(--sp) : PT_WALK_DONE;
}
// hit_mesh for one object by the quads of a wave (see object_test); `best` stays identical in the lanes of a quad.
template <bool STATS, bool LB, class Stk>
PT_DEV void object_test_quad(const SceneView& sc, uint32_t obj, v3 ro, v3 rd, uint32_t q, Best& best, Stk stk, LaneCounters& cnt) {
    const DObject* ob = &sc.objects[obj];
    const Affine w2l = ob->w2l;
    const uint32_t mesh_id = ob->mesh;
    const v3 lo = xform_point(w2l, ro);
    const v3 ld = xform_dir(w2l, rd);
    if (STATS && q == 0u) cnt.xforms++;
    bool pending = true;
    while (__any(pending)) {
        if (pending) {
            const uint32_t m = __builtin_amdgcn_readfirstlane(mesh_id);
            uint32_t m_cmp = m;
            asm volatile("" : "+s"(m_cmp));   // (see object_test)
            if (mesh_id == m_cmp) {
                pending = false;
                const DMesh* me = &sc.meshes[m];
                const uint32_t root = me->blas_root, tri_first = me->tri_first;
                const float pad = 0x1p-18f * (pt_max(pt_max(__builtin_fabsf(lo.x), __builtin_fabsf(lo.y)), __builtin_fabsf(lo.z)) + me->max_abs);
                const SlabRay sr = make_slab_ray(lo, ld, pad);
                Best cand = best;   // own: the closest hit with the triangles this lane tests
                bool changed = false;
                if (root == PT_NONE) {
                    const uint32_t n = me->tri_count;
                    float tn;
                    if (STATS && q == 0u) cnt.box_tests++;
                    const bool inside = n <= PT_LOCAL_BOX_MIN_TRIS ||
                        slab_test(sr, make_float4(me->bmin[0], me->bmin[1], me->bmin[2], 0.0f), make_float4(me->bmax[0], me->bmax[1], me->bmax[2], 0.0f), cand.t, tn);
                    if (inside) {
                        for (uint32_t g = tri_first + q; g < tri_first + n; g += 4u) {   // own: every fourth triangle
                            const DTri* T = &sc.tris[g];
                            changed |= tri_test<STATS>(T->q0, T->q1, T->q2, lo, ld, obj, g, sc.tie_lowest, cand, cnt);
                        }
                    }
                } else {
                    const WideRay wr = make_wide_ray(sr);
                    Stk qstk = stk;
                    qstk.base -= q;   // the quad leader's column
                    auto walk = [&](const WNode* nodes, const DTri* tris, const uint32_t* ids) {
                        uint32_t node = root;
                        int sp = 0;
                        float tmax = cand.t;   // the quad's closest hit so far (identical in its lanes)
                        while (__any(node != PT_WALK_DONE)) {
                            if (node != PT_WALK_DONE && !(node & PT_LEAF_BIT))
                                node = quad_visit<STATS, Stk>(reinterpret_cast<const uint4*>(nodes + node), q, wr, tmax, qstk, sp, 0, cnt);
                            if (node != PT_WALK_DONE && (node & PT_LEAF_BIT)) {
                                const uint32_t leaf = node & ~PT_LEAF_BIT;
                                const uint32_t pos = leaf & PT_WLEAF_POS_MASK, n = (leaf >> PT_WLEAF_COUNT_SHIFT) + 1u;
                                if (q < n) {   // own: triangle q of the leaf (a leaf has at most four)
                                    const DTri* T = &tris[pos + q];
                                    changed |= tri_test<STATS>(T->q0, T->q1, T->q2, lo, ld, obj, tri_first + ids[pos + q], sc.tie_lowest, cand, cnt);
                                }
                                tmax = quad_min_f32(cand.t);
                                node = (sp > 0) ? qstk.get(--sp) : PT_WALK_DONE;
                            }
                        }
                    };
                    if (LB && m == sc.lb_mesh) {
                        if (sc.lb_tris) walk(sc.lb_nodes - sc.lb_node_base, sc.lb_tris, sc.lb_tri_ids);
                        else walk(sc.lb_nodes - sc.lb_node_base, sc.tris + tri_first, sc.tri_ids + tri_first);
                    } else {
                        walk(sc.blas_nodes, sc.tris + tri_first, sc.tri_ids + tri_first);
                    }
                }
                // the quad's winner, in all of its lanes; then the reference's gate, as in object_test
                const unsigned long long cm = __ballot(changed);
                const bool any_changed = ((cm >> ((threadIdx.x & 63u) & ~3u)) & 0xFull) != 0ull;
                quad_best(cand, sc.tie_lowest);
                if (any_changed) {
                    const float t_aabb = hit_box_exact(me->bmin, me->bmax, lo, ld);
                    if (!(t_aabb < PT_T_MIN)) best = cand;
                }
            }
        }
    }
}
// hit_all for the rays of the quads' leaders (see closest_hit_list; object lists only). ro, rd, active: the leader's,
// replicated in its quad by the caller.
template <bool STATS, bool LB, class Stk>
PT_DEV void closest_hit_list_quad(const SceneView& sc, v3 ro, v3 rd, bool active, Best& best, Stk stk, LaneCounters& cnt) {
    const uint32_t q = threadIdx.x & 3u;
    const SlabRay sr = world_slab_ray(ro, rd);
    const uint32_t n = sc.n_objects;
    const bool cull = sc.cull_boxes != 0;
    const uint32_t n_groups = (n + 31u) / 32u;
    for (uint32_t g = 0; g < n_groups; ++g) {
        const uint32_t base = g * 32u;
        const uint32_t m = (n - base < 32u) ? n - base : 32u;
        uint32_t mask = 0, near_k = 0;
        float near_t = PT_T_MAX;
        if (cull) {
            for (uint32_t k = 0; k < m; ++k) {
                float tn;
                if (STATS && q == 0u) cnt.box_tests++;
                if (slab_test(sr, sc.obj_lo[base + k], sc.obj_hi[base + k], best.t, tn)) {
                    mask |= 1u << k;
                    if (tn < near_t) { near_t = tn; near_k = k; }
                }
            }
        } else {
            mask = (m == 32u) ? 0xFFFFFFFFu : ((1u << m) - 1u);
        }
        if (!active) mask = 0;
        bool first = cull;
        while (__any(mask != 0)) {
            cnt.passes++;
            if (mask != 0) {
                const uint32_t k = first ? near_k : (uint32_t)__builtin_ctz(mask);
                const uint32_t o = base + k;
                mask &= ~(1u << k);
                float tn;
                if (!cull || first || slab_test(sr, sc.obj_lo[o], sc.obj_hi[o], best.t, tn)) object_test_quad<STATS, LB>(sc, o, ro, rd, q, best, stk, cnt);
            }
            first = false;
        }
    }
}

// A TLAS node: from the copy of the tree's top levels in LDS when the kernel made one, else from global memory. Two
// branches rather than one load through a chosen pointer: that one would be a flat load, which takes a slot in the LDS
// and in the vector-memory queue for every lane, whichever memory the lane reads.
// (The address spaces are spelled out: given two branches that load through generic pointers, the compiler merges them
// into the one flat load again.)
typedef uint32_t pt_u32x4 __attribute__((ext_vector_type(4)));
PT_DEV void tlas_fetch(const SceneView& sc, uint32_t r, uint4& q0, uint4& q1, uint4& q2, uint4& q3) {
    pt_u32x4 a, b, c, d;
    if (r < sc.tlas_top_n) {
        const __attribute__((address_space(3))) pt_u32x4* N = (const __attribute__((address_space(3))) pt_u32x4*)(sc.tlas_top + r);
        a = N[0]; b = N[1]; c = N[2]; d = N[3];
    } else {
        const __attribute__((address_space(1))) pt_u32x4* N = (const __attribute__((address_space(1))) pt_u32x4*)(sc.tlas_nodes + r);
        a = N[0]; b = N[1]; c = N[2]; d = N[3];
    }
    q0 = make_uint4(a.x, a.y, a.z, a.w); q1 = make_uint4(b.x, b.y, b.z, b.w);
    q2 = make_uint4(c.x, c.y, c.z, c.w); q3 = make_uint4(d.x, d.y, d.z, d.w);
}

// hit_all through the TLAS: each lane walks the object tree on its own.
template <bool STATS, class Stk>
PT_DEV void closest_hit_tlas(const SceneView& sc, v3 ro, v3 rd, bool active, Best& best, Stk stk, LaneCounters& cnt) {
    if (sc.tlas_root == PT_NONE) return;
    const WideRay wr = make_wide_ray(world_slab_ray(ro, rd));
    wide_walk<STATS, PT_BURST_TLAS>([&](uint32_t r, uint4& q0, uint4& q1, uint4& q2, uint4& q3) { tlas_fetch(sc, r, q0, q1, q2, q3); }, sc.tlas_root, wr, best.t, stk, 0, active, cnt,
                    [&](uint32_t obj, int sp) { object_test<STATS, false>(sc, obj, ro, rd, best, stk, sp, cnt); });
}

// MODE: how hit_all finds its candidates.
#define PT_MODE_LIST 0  // object list culled with world boxes, scene records in global memory
#define PT_MODE_TLAS 1  // TLAS walk
#define PT_MODE_LDS 2   // object list, every scene record staged in LDS (small scenes)
#define PT_MODE_LIST_LB 3  // object list, the BLAS of one mesh staged in LDS (k_paths only)
template <int MODE, bool STATS, class Stk>
PT_DEV void closest_hit(const SceneView& sc, v3 ro, v3 rd, bool active, Best& best, Stk stk, LaneCounters& cnt) {
    best.t = PT_T_MAX;
    best.obj = PT_NONE;
    best.tri = PT_NONE;
    best.u = 0.0f;
    best.v = 0.0f;
    if (MODE == PT_MODE_TLAS) closest_hit_tlas<STATS>(sc, ro, rd, active, best, stk, cnt);
    else closest_hit_list<STATS, MODE == PT_MODE_LDS, MODE == PT_MODE_LIST_LB>(sc, ro, rd, active, best, stk, cnt);
}

// The hit_record of the winner (reference src/query.wgsl:113-120), evaluated once per ray.
// GLOBAL_TRI: how Best::tri counts (see there).
template <bool GLOBAL_TRI>
PT_DEV HitInfo finalize_hit(const SceneView& sc, v3 ro, v3 rd, const Best& b) {
    HitInfo h;
    h.mat = PT_NONE;
    h.p = V3(0, 0, 0);
    h.n = V3(0, 0, 0);
    h.u = 0.0f;
    h.v = 0.0f;
    if (b.obj == PT_NONE) return h;
    const DObject* ob = &sc.objects[b.obj];
    const Affine w2l = ob->w2l, l2w = sc.obj_l2w[b.obj];
    v3 lo = xform_point(w2l, ro);
    v3 ld = xform_dir(w2l, rd);
    const DTriAttr* A = &sc.tri_attrs[GLOBAL_TRI ? b.tri : ob->tri_first + b.tri];
    h.mat = ob->mat;
    float4 a0 = A->q0, a1 = A->q1, a2 = A->q2, a3 = A->q3;
    float u = b.u, v = b.v;
    float w = 1.0f - u - v;
    v3 _p = madd3(lo, ld, b.t);
    v3 _n = V3(pt_fma(a1.z, v, pt_fma(a0.w, u, a0.x * w)), pt_fma(a1.w, v, pt_fma(a1.x, u, a0.y * w)),
               pt_fma(a2.x, v, pt_fma(a1.y, u, a0.z * w)));
    h.u = pt_fma(a3.y, v, pt_fma(a2.w, u, a2.y * w));
    h.v = pt_fma(a3.z, v, pt_fma(a3.x, u, a2.z * w));
    h.p = xform_point(l2w, _p);
    h.n = normalize3(xform_dir(l2w, _n));
    return h;
}

// ---- shading: reference src/raytrace.wgsl:88-203 ----------------------------------

PT_DEV float texel(const SceneView& sc, uint32_t i) { return (i < sc.n_texture_data) ? sc.texture_data[i] : 0.0f; }

// src/raytrace.wgsl:120-153 (row stride is `height`, as in the reference)
PT_DEV v3 sample_texture(const SceneView& sc, uint32_t idx, float u, float v) {
    DTexture tx = sc.textures[idx];
    float x = u * (float)tx.width;
    float y = v * (float)tx.height;
    uint32_t i = tx.offset + (pt_f2u(x) + pt_f2u(y) * tx.height) * tx.format;
    switch (tx.format) {
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

// bevy_pbr::lighting::F_Schlick with f0 = 1
PT_DEV float f_schlick_1(float f90, float u) { return pt_fma(f90 - 1.0f, pt_pow5(1.0f - u), 1.0f); }

// src/raytrace.wgsl:157-203, in two parts. N is hit_record.n after the normal-map multiply.
// brdf_frame: everything that depends only on the hit and the incoming ray (tangent frame, reflected ray, albedo,
// N.V): the same for every sample of a pixel when the hit is the primary hit, so k_paths keeps it per pixel.
// brdf_sample: the part that draws random numbers. calculate_brdf = both, the operations and their order unchanged.
struct BrdfFrame {
    v3 tt, bb;     // hugues_moller(N) columns 0 and 1 (:102-116)
    v3 refl_om;    // normalize(reflect(ray_dir, N)) * (1 - roughness)
    float NdotV;
    v3 albedo;
};
PT_DEV BrdfFrame brdf_frame(const SceneView& sc, v3 ray_dir, v3 N, float hu, float hv, const DMaterial& m) {
    BrdfFrame f;
    float ax = __builtin_fabsf(N.x), ay = __builtin_fabsf(N.y), az = __builtin_fabsf(N.z);
    v3 tt;
    if (ax <= ay && ax <= az) tt = V3(0.0f, -N.z, N.y);
    else if (ay <= ax && ay <= az) tt = V3(-N.z, 0.0f, N.x);
    else tt = V3(-N.y, N.x, 0.0f);
    f.tt = normalize3(tt);
    f.bb = normalize3(cross3(N, f.tt));
    float k2 = 2.0f * dot3(N, ray_dir);
    v3 reflection_ray = normalize3(madd3(ray_dir, N, -k2));
    float om = 1.0f - m.roughness;
    f.refl_om = V3(reflection_ray.x * om, reflection_ray.y * om, reflection_ray.z * om);
    v3 albedo = V3(m.albedo[0], m.albedo[1], m.albedo[2]);
    if (m.albedo_texture != PT_NONE) albedo = mul3(albedo, sample_texture(sc, m.albedo_texture, hu, hv));
    f.albedo = albedo;
    f.NdotV = pt_max(dot3(N, neg3(ray_dir)), 0.0001f);
    return f;
}
PT_DEV void brdf_sample(Rng& rng, v3 ray_dir, v3 N, float rough_in, const BrdfFrame& f, v3& out_dir, v3& out_color) {
    // cosine_sample (:88-95)
    pcg3d(rng);
    float r0 = rand_component(rng.x), r1 = rand_component(rng.y);
    float phi = 2.0f * PT_PI * r0;
    float sin_theta = pt_sqrt_in_range(r1);          // r1 is 0 or >= 2^-24, 1 - r1 is in [2^-24, 1]
    float cos_theta = pt_sqrt_in_range(1.0f - r1);
    float sn, cs;
    pt_sincos(phi, sn, cs);
    v3 c = V3(sin_theta * cs, sin_theta * sn, cos_theta);
    v3 lam = V3(pt_fma(N.x, c.z, pt_fma(f.bb.x, c.y, f.tt.x * c.x)), pt_fma(N.y, c.z, pt_fma(f.bb.y, c.y, f.tt.y * c.x)),
                pt_fma(N.z, c.z, pt_fma(f.bb.z, c.y, f.tt.z * c.x)));
    v3 lambertian_ray = normalize3(lam);
    v3 L = V3(pt_fma(lambertian_ray.x, rough_in, f.refl_om.x), pt_fma(lambertian_ray.y, rough_in, f.refl_om.y),
              pt_fma(lambertian_ray.z, rough_in, f.refl_om.z));
    v3 V = neg3(ray_dir);
    float cr = pt_min(pt_max(rough_in, 0.089f), 1.0f);
    float roughness = cr * cr;
    v3 H = normalize3(add3(L, V));
    float NdotL = pt_saturate(dot3(N, L));
    float LdotH = pt_saturate(dot3(L, H));
    float f90 = pt_fma((2.0f * roughness) * LdotH, LdotH, 0.5f);
    float fd = (f_schlick_1(f90, NdotL) * f_schlick_1(f90, f.NdotV)) * (1.0f / PT_PI);
    out_dir = L;
    out_color = scale3(scale3(f.albedo, fd), PT_PI);
}
PT_DEV void calculate_brdf(const SceneView& sc, Rng& rng, v3 ray_dir, v3 N, float hu, float hv, const DMaterial& m,
                           v3& out_dir, v3& out_color) {
    const BrdfFrame f = brdf_frame(sc, ray_dir, N, hu, hv, m);
    brdf_sample(rng, ray_dir, N, m.roughness, f, out_dir, out_color);
}

// True only when calculate_brdf at this hit is certain to leave max(ray_color * brdf.color) == 0, so that the path
// ends on the throughput cutoff (src/raytrace.wgsl:266-269) whatever direction is sampled. brdf.color is
// (albedo * Fd_Burley) * PI per channel. Fd_Burley is finite whenever N.V is: every other input of it is clamped to
// [0, 1] (NaNs included) and roughness^2 to [0.089^2, 1]. Then a channel whose albedo or incoming throughput is
// exactly zero, the other factor being finite and far from overflow, comes out as +-0; three such channels give
// p = +-0 < EPSILON. Materials with an albedo texture or a normal map take the general path.
PT_DEV bool scatter_goes_dark(const SceneView& sc, const HitInfo& h, v3 ray_dir, v3 ray_color) {
    const DMaterial* m = &sc.materials[h.mat];
    if (m->albedo_texture != PT_NONE || m->normal_map_texture != PT_NONE) return false;
    const float big = 1.0e30f;
    const float ax = m->albedo[0], ay = m->albedo[1], az = m->albedo[2];
    const bool zx = (ax == 0.0f || ray_color.x == 0.0f) && __builtin_fabsf(ax) <= big && __builtin_fabsf(ray_color.x) <= big;
    const bool zy = (ay == 0.0f || ray_color.y == 0.0f) && __builtin_fabsf(ay) <= big && __builtin_fabsf(ray_color.y) <= big;
    const bool zz = (az == 0.0f || ray_color.z == 0.0f) && __builtin_fabsf(az) <= big && __builtin_fabsf(ray_color.z) <= big;
    if (!(zx && zy && zz)) return false;
    const float NdotV = pt_max(dot3(h.n, neg3(ray_dir)), 0.0001f);
    return NdotV <= 2.0f;   // (false for NaN) |1 - N.V| <= 1, so both Schlick factors are finite
}

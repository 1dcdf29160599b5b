// pt_build.h — kernels that turn the uploaded wire arrays into the traversal layout:
// triangle records, instance records with world boxes, and the LBVH (Morton codes,
// LSD radix sort, Karras hierarchy, bottom-up box fit) used for both the per-mesh BLAS
// and the per-frame TLAS. The reference has no acceleration structure ("BVH" is a TODO in
// its README.md:3-7); it loops over every object and triangle (src/query.wgsl:55-125).
#pragma once

#include "pt_scene.h"

// wire structs as the device sees them (include/pt_amd.h)
struct WVertex {
    float position[3]; uint32_t _p0;
    float normal[3];   uint32_t _p1;
    float uv[2];       uint32_t _p2[2];
};
struct WMesh {
    float aabb_min[3]; uint32_t _p0;
    float aabb_max[3];
    uint32_t ihead, vhead, tri_count, _p1[2];
};
struct WObject {
    float l2w[16];
    float w2l[16];
    uint32_t mat, mesh, _p[2];
};
static_assert(sizeof(WVertex) == 48 && sizeof(WMesh) == 48 && sizeof(WObject) == 144, "wire structs");

// One thread per triangle of every mesh: geometry record, attribute record, and its box.
__global__ void k_build_tris(const WMesh* __restrict__ meshes, const uint32_t* __restrict__ tri_mesh,
                             const uint32_t* __restrict__ mesh_tri_first, const uint32_t* __restrict__ indices,
                             const WVertex* __restrict__ vertices, uint32_t n_tris, DTri* __restrict__ tris,
                             DTriAttr* __restrict__ attrs, float4* __restrict__ box_lo, float4* __restrict__ box_hi) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_tris) return;
    uint32_t m = tri_mesh[g];
    WMesh me = meshes[m];
    uint32_t i3 = (g - mesh_tri_first[m]) * 3u;
    const WVertex* va = &vertices[me.vhead + indices[me.ihead + i3]];
    const WVertex* vb = &vertices[me.vhead + indices[me.ihead + i3 + 1u]];
    const WVertex* vc = &vertices[me.vhead + indices[me.ihead + i3 + 2u]];
    v3 a = V3(va->position[0], va->position[1], va->position[2]);
    v3 b = V3(vb->position[0], vb->position[1], vb->position[2]);
    v3 c = V3(vc->position[0], vc->position[1], vc->position[2]);
    // reference src/query.wgsl:95-97
    v3 e_ab = sub3(b, a);
    v3 e_ac = sub3(c, a);
    v3 n = cross3(e_ab, e_ac);
    DTri T;
    T.q0 = make_float4(a.x, a.y, a.z, e_ab.x);
    T.q1 = make_float4(e_ab.y, e_ab.z, e_ac.x, e_ac.y);
    T.q2 = make_float4(e_ac.z, n.x, n.y, n.z);
    tris[g] = T;
    DTriAttr A;
    A.q0 = make_float4(va->normal[0], va->normal[1], va->normal[2], vb->normal[0]);
    A.q1 = make_float4(vb->normal[1], vb->normal[2], vc->normal[0], vc->normal[1]);
    A.q2 = make_float4(vc->normal[2], va->uv[0], va->uv[1], vb->uv[0]);
    A.q3 = make_float4(vb->uv[1], vc->uv[0], vc->uv[1], 0.0f);
    attrs[g] = A;
    box_lo[g] = make_float4(fminf(a.x, fminf(b.x, c.x)), fminf(a.y, fminf(b.y, c.y)), fminf(a.z, fminf(b.z, c.z)), 0.0f);
    box_hi[g] = make_float4(fmaxf(a.x, fmaxf(b.x, c.x)), fmaxf(a.y, fmaxf(b.y, c.y)), fmaxf(a.z, fmaxf(b.z, c.z)), 0.0f);
}

// One thread per object: instance record + padded world box of the mesh AABB.
// The TLAS box only culls; the exact per-object gate runs in local space afterwards, with a ray
// moved by world_to_local, which is a separately rounded inverse. The pad therefore covers
// (a) rounding of the corner transform and of the slab arithmetic, 2^-16 relative to the box's
// largest coordinate, and (b) how far local_to_world * world_to_local is from the identity.
__global__ void k_prep_objects(const WObject* __restrict__ wobj, const uint32_t* __restrict__ tri_base,
                               const DMesh* __restrict__ meshes, uint32_t n, DObject* __restrict__ objects,
                               Affine* __restrict__ obj_l2w, float4* __restrict__ box_lo, float4* __restrict__ box_hi) {
    uint32_t o = blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= n) return;
    const WObject* w = &wobj[o];
    DObject d;
    d.w2l.r0 = make_float4(w->w2l[0], w->w2l[4], w->w2l[8], w->w2l[12]);
    d.w2l.r1 = make_float4(w->w2l[1], w->w2l[5], w->w2l[9], w->w2l[13]);
    d.w2l.r2 = make_float4(w->w2l[2], w->w2l[6], w->w2l[10], w->w2l[14]);
    Affine l2w;
    l2w.r0 = make_float4(w->l2w[0], w->l2w[4], w->l2w[8], w->l2w[12]);
    l2w.r1 = make_float4(w->l2w[1], w->l2w[5], w->l2w[9], w->l2w[13]);
    l2w.r2 = make_float4(w->l2w[2], w->l2w[6], w->l2w[10], w->l2w[14]);
    d.mat = w->mat;
    d.mesh = w->mesh;
    d.tri_base = tri_base[o];
    d.tri_first = meshes[w->mesh].tri_first;
    objects[o] = d;
    obj_l2w[o] = l2w;

    DMesh me = meshes[w->mesh];
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (int k = 0; k < 8; ++k) {
        v3 c = V3((k & 1) ? me.bmax[0] : me.bmin[0], (k & 2) ? me.bmax[1] : me.bmin[1], (k & 4) ? me.bmax[2] : me.bmin[2]);
        v3 p = xform_point(l2w, c);
        lo[0] = fminf(lo[0], p.x); lo[1] = fminf(lo[1], p.y); lo[2] = fminf(lo[2], p.z);
        hi[0] = fmaxf(hi[0], p.x); hi[1] = fmaxf(hi[1], p.y); hi[2] = fmaxf(hi[2], p.z);
    }
    // residual of l2w * w2l against the identity (3x4 affine parts)
    const float4 L[3] = {l2w.r0, l2w.r1, l2w.r2};
    const float4 Wm[3] = {d.w2l.r0, d.w2l.r1, d.w2l.r2};
    float resid = 0.0f;
    for (int i = 0; i < 3; ++i) {
        float li[3] = {L[i].x, L[i].y, L[i].z};
        for (int j = 0; j < 4; ++j) {
            float wj[3] = {j == 0 ? Wm[0].x : j == 1 ? Wm[0].y : j == 2 ? Wm[0].z : Wm[0].w,
                           j == 0 ? Wm[1].x : j == 1 ? Wm[1].y : j == 2 ? Wm[1].z : Wm[1].w,
                           j == 0 ? Wm[2].x : j == 1 ? Wm[2].y : j == 2 ? Wm[2].z : Wm[2].w};
            float s = li[0] * wj[0] + li[1] * wj[1] + li[2] * wj[2] + (j == 3 ? L[i].w : 0.0f);
            float want = (i == j) ? 1.0f : 0.0f;
            // the translation column is measured relative to the translation's size
            float scale = (j == 3) ? fmaxf(1.0f, fabsf(L[i].w)) : 1.0f;
            resid = fmaxf(resid, fabsf(s - want) / scale);
        }
    }
    float max_abs = 0.0f, ext = 0.0f;
    for (int k = 0; k < 3; ++k) {
        max_abs = fmaxf(max_abs, fmaxf(fabsf(lo[k]), fabsf(hi[k])));
        ext = fmaxf(ext, hi[k] - lo[k]);
    }
    float pad = (0x1p-16f + 8.0f * resid) * (max_abs + ext) + 1.0e-30f;
    if (!(pad == pad) || !(max_abs < 3.0e38f)) {  // non-finite box: make it unbounded so that it is never culled
        lo[0] = lo[1] = lo[2] = -3.0e38f;
        hi[0] = hi[1] = hi[2] = 3.0e38f;
        pad = 0.0f;
    }
    box_lo[o] = make_float4(lo[0] - pad, lo[1] - pad, lo[2] - pad, 0.0f);
    box_hi[o] = make_float4(hi[0] + pad, hi[1] + pad, hi[2] + pad, 0.0f);
}

// ---- bounds of a set of boxes (for Morton normalisation) ---------------------------------
PT_DEV uint32_t float_to_ordered(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ inline float ordered_to_float(uint32_t o) {
    uint32_t u = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
#ifdef __HIP_DEVICE_COMPILE__
    return __uint_as_float(u);
#else
    float f;
    memcpy(&f, &u, 4);
    return f;
#endif
}
// bounds[0..2] = min of centroids (ordered encoding), bounds[3..5] = max. Initialise to 0xFFFFFFFF / 0.
__global__ void k_centroid_bounds(const float4* __restrict__ box_lo, const float4* __restrict__ box_hi, uint32_t first,
                                  uint32_t n, uint32_t* __restrict__ bounds) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    float c[3] = {0, 0, 0};
    bool ok = i < n;
    if (ok) {
        float4 lo = box_lo[first + i], hi = box_hi[first + i];
        c[0] = 0.5f * lo.x + 0.5f * hi.x;
        c[1] = 0.5f * lo.y + 0.5f * hi.y;
        c[2] = 0.5f * lo.z + 0.5f * hi.z;
    }
    for (int k = 0; k < 3; ++k) {
        uint32_t mn = ok ? float_to_ordered(c[k]) : 0xFFFFFFFFu;
        uint32_t mx = ok ? float_to_ordered(c[k]) : 0u;
        for (int off = 32; off > 0; off >>= 1) {
            mn = min(mn, (uint32_t)__shfl_xor(mn, off, 64));
            mx = max(mx, (uint32_t)__shfl_xor(mx, off, 64));
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&bounds[k], mn);
            atomicMax(&bounds[3 + k], mx);
        }
    }
}

PT_DEV uint32_t expand_bits10(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__global__ void k_morton(const float4* __restrict__ box_lo, const float4* __restrict__ box_hi, uint32_t first, uint32_t n,
                         const uint32_t* __restrict__ bounds, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float4 lo = box_lo[first + i], hi = box_hi[first + i];
    float c[3] = {0.5f * lo.x + 0.5f * hi.x, 0.5f * lo.y + 0.5f * hi.y, 0.5f * lo.z + 0.5f * hi.z};
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        float mn = ordered_to_float(bounds[k]), mx = ordered_to_float(bounds[3 + k]);
        float e = mx - mn;
        float f = (e > 0.0f) ? (c[k] - mn) / e : 0.0f;
        f = fminf(fmaxf(f * 1024.0f, 0.0f), 1023.0f);
        q[k] = (uint32_t)f;
    }
    keys[i] = (expand_bits10(q[0]) << 2) | (expand_bits10(q[1]) << 1) | expand_bits10(q[2]);
    vals[i] = i;
}

// ---- LSD radix sort, 8 bits per pass, stable. One wave per block, RS_KEYS keys per block. ----
#define RS_ITEMS 16
#define RS_KEYS (64 * RS_ITEMS)

template <int BITS = 8>
__global__ __launch_bounds__(64) void k_rs_hist(const uint32_t* __restrict__ keys, uint32_t n, const uint32_t* __restrict__ n_ptr,
                                                uint32_t shift, uint32_t n_blocks, uint32_t* __restrict__ hist) {
    constexpr int DIGITS = 1 << BITS;
    __shared__ uint32_t cnt[DIGITS];
    if (n_ptr) n = *n_ptr;  // key count produced on the device (the grid is sized for the largest possible count)
    for (int d = threadIdx.x; d < DIGITS; d += 64) cnt[d] = 0;
    __syncthreads();
    uint32_t base = blockIdx.x * RS_KEYS;
    for (int j = 0; j < RS_ITEMS; ++j) {
        uint32_t i = base + j * 64 + threadIdx.x;
        if (i < n) atomicAdd(&cnt[(keys[i] >> shift) & (DIGITS - 1u)], 1u);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < DIGITS; d += 64) hist[d * n_blocks + blockIdx.x] = cnt[d];
}

__global__ __launch_bounds__(1024) void k_rs_scan(uint32_t* __restrict__ hist, uint32_t total) {
    __shared__ uint32_t part[1024];
    uint32_t per = (total + 1023u) / 1024u;
    uint32_t b = threadIdx.x * per, e = min(b + per, total);
    uint32_t s = 0;
    for (uint32_t i = b; i < e; ++i) s += hist[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 1024; off <<= 1) {
        uint32_t v = (threadIdx.x >= off) ? part[threadIdx.x - off] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;
    for (uint32_t i = b; i < e; ++i) {
        uint32_t v = hist[i];
        hist[i] = run;
        run += v;
    }
}

template <int BITS = 8>
__global__ __launch_bounds__(64) void k_rs_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                   uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t n,
                                                   const uint32_t* __restrict__ n_ptr, uint32_t shift, uint32_t n_blocks,
                                                   const uint32_t* __restrict__ hist) {
    constexpr int DIGITS = 1 << BITS;
    __shared__ uint32_t offs[DIGITS];
    if (n_ptr) n = *n_ptr;
    for (int d = threadIdx.x; d < DIGITS; d += 64) offs[d] = hist[d * n_blocks + blockIdx.x];
    __syncthreads();
    const uint32_t lane = threadIdx.x;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t base = blockIdx.x * RS_KEYS;
    for (int j = 0; j < RS_ITEMS; ++j) {
        uint32_t i = base + j * 64 + lane;
        bool ok = i < n;
        uint32_t key = ok ? keys_in[i] : 0u;
        uint32_t val = ok ? (vals_in ? vals_in[i] : i) : 0u;   // no value array: the values are the positions
        uint32_t digit = (key >> shift) & (DIGITS - 1u);
        unsigned long long peers = __ballot(ok);
        for (int bit = 0; bit < BITS; ++bit) {
            unsigned long long bmask = __ballot((digit >> bit) & 1u);
            peers &= ((digit >> bit) & 1u) ? bmask : ~bmask;
        }
        uint32_t rank = __popcll(peers & lt_mask);
        uint32_t pos = 0;
        if (ok) pos = offs[digit] + rank;
        __syncthreads();
        if (ok && rank == 0) offs[digit] += __popcll(peers);
        __syncthreads();
        if (ok) {
            if (keys_out) keys_out[pos] = key;
            vals_out[pos] = val;
        }
    }
}

// ---- Karras 2012: internal node i of n-1, over n sorted keys -----------------------------
PT_DEV int lbvh_delta(const uint32_t* __restrict__ keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    uint32_t a = keys[i], b = keys[j];
    if (a == b) return 32 + __clz((uint32_t)i ^ (uint32_t)j);
    return __clz(a ^ b);
}
// ranges (may be null): for internal node i the sorted positions [x, y] of the keys below it.
__global__ void k_karras(const uint32_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ parent_internal,
                         uint32_t* __restrict__ parent_leaf, uint2* __restrict__ ranges) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int ni = (int)n;
    if (i >= ni - 1) return;
    int d = (lbvh_delta(keys, ni, i, i + 1) - lbvh_delta(keys, ni, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = lbvh_delta(keys, ni, i, i - d);
    int lmax = 2;
    while (lbvh_delta(keys, ni, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (lbvh_delta(keys, ni, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = lbvh_delta(keys, ni, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) >> 1;
        if (lbvh_delta(keys, ni, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    int gamma = i + s * d + min(d, 0);
    int lo = min(i, j), hi = max(i, j);
    if (ranges) ranges[i] = make_uint2((uint32_t)lo, (uint32_t)hi);
    // the left child covers [lo, gamma], the right one [gamma+1, hi]; a one-key range is a leaf.
    // parent link: bit 31 = "I am the right child"
    if (lo == gamma) parent_leaf[gamma] = (uint32_t)i; else parent_internal[gamma] = (uint32_t)i;
    if (hi == gamma + 1) parent_leaf[gamma + 1] = (uint32_t)i | 0x80000000u; else parent_internal[gamma + 1] = (uint32_t)i | 0x80000000u;
}

// Bottom-up fit. One thread per leaf climbs towards the root; at each internal node the first
// arrival stores its child's box and stops, the second one merges and continues. A child box is
// handed between workgroups (possibly on different XCDs, whose L2s are not coherent) with an
// agent-scope release before the counter increment and an agent-scope acquire after it.
// out_info[0] = longest root-to-leaf path in internal nodes.
// leaf_is_pos: a leaf is referenced by its position in the sorted order (BLAS: the triangle records are stored in that
// order) instead of by the primitive's own number (TLAS: the object).
__global__ void k_fit(const float4* __restrict__ box_lo, const float4* __restrict__ box_hi, uint32_t first,
                      const uint32_t* __restrict__ sorted_prim, uint32_t n, const uint32_t* __restrict__ parent_internal,
                      const uint32_t* __restrict__ parent_leaf, uint32_t* __restrict__ arrive, BvhNode* __restrict__ nodes,
                      uint32_t node_base, uint32_t* __restrict__ out_info, uint32_t leaf_is_pos) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t prim = sorted_prim[k];
    float4 lo = box_lo[first + prim], hi = box_hi[first + prim];
    uint32_t ref = PT_LEAF_BIT | (leaf_is_pos ? k : prim);
    uint32_t depth = 0;
    uint32_t link = parent_leaf[k];
    while (true) {
        uint32_t p = link & 0x7FFFFFFFu;
        bool right = (link & 0x80000000u) != 0;
        BvhNode* N = &nodes[p];
        float4 slot_lo = make_float4(lo.x, lo.y, lo.z, __uint_as_float(ref));
        float4 slot_hi = make_float4(hi.x, hi.y, hi.z, __uint_as_float(depth));
        if (right) { N->lo_r = slot_lo; N->hi_r = slot_hi; } else { N->lo_l = slot_lo; N->hi_l = slot_hi; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t old = __hip_atomic_fetch_add(&arrive[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == 0u) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const volatile float* other_lo = reinterpret_cast<const volatile float*>(right ? &N->lo_l : &N->lo_r);
        const volatile float* other_hi = reinterpret_cast<const volatile float*>(right ? &N->hi_l : &N->hi_r);
        float sx = other_lo[0], sy = other_lo[1], sz = other_lo[2];
        float tx = other_hi[0], ty = other_hi[1], tz = other_hi[2];
        uint32_t sdepth = __float_as_uint(other_hi[3]);
        lo = make_float4(fminf(lo.x, sx), fminf(lo.y, sy), fminf(lo.z, sz), 0.0f);
        hi = make_float4(fmaxf(hi.x, tx), fmaxf(hi.y, ty), fmaxf(hi.z, tz), 0.0f);
        depth = max(depth, sdepth) + 1u;
        ref = node_base + p;
        if (p == 0u) {
            out_info[0] = depth;
            return;
        }
        link = parent_internal[p];
    }
}

// Checks every node of a built tree: each stored child box must contain what is below it.
// errors[0] counts violations. (Debug / test aid.)
// sorted_prim (may be null): leaf references are positions in this order instead of primitive numbers.
__global__ void k_check_bvh(const BvhNode* __restrict__ nodes, uint32_t node_base, uint32_t n_internal,
                            const float4* __restrict__ box_lo, const float4* __restrict__ box_hi, uint32_t first,
                            const uint32_t* __restrict__ sorted_prim, uint32_t* __restrict__ errors) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_internal) return;
    const BvhNode N = nodes[i];
    for (int side = 0; side < 2; ++side) {
        float4 lo = side ? N.lo_r : N.lo_l, hi = side ? N.hi_r : N.hi_l;
        uint32_t ref = __float_as_uint(lo.w);
        float4 clo, chi;
        if (ref & PT_LEAF_BIT) {
            const uint32_t prim = sorted_prim ? sorted_prim[ref & ~PT_LEAF_BIT] : (ref & ~PT_LEAF_BIT);
            clo = box_lo[first + prim];
            chi = box_hi[first + prim];
        } else {
            uint32_t c = ref - node_base;
            if (c >= n_internal) { atomicAdd(&errors[0], 1u); continue; }
            const BvhNode C = nodes[c];
            clo = make_float4(fminf(C.lo_l.x, C.lo_r.x), fminf(C.lo_l.y, C.lo_r.y), fminf(C.lo_l.z, C.lo_r.z), 0.0f);
            chi = make_float4(fmaxf(C.hi_l.x, C.hi_r.x), fmaxf(C.hi_l.y, C.hi_r.y), fmaxf(C.hi_l.z, C.hi_r.z), 0.0f);
        }
        bool ok = lo.x <= clo.x && lo.y <= clo.y && lo.z <= clo.z && hi.x >= chi.x && hi.y >= chi.y && hi.z >= chi.z;
        if (!ok) atomicAdd(&errors[0], 1u);
    }
}

// ---- triangle records of a mesh with a BLAS, in the order of the tree's leaves ---------------------------------------
__global__ void k_permute_tris(const DTri* __restrict__ tris_in, const uint32_t* __restrict__ sorted_prim, uint32_t n,
                               DTri* __restrict__ tris_out, uint32_t* __restrict__ tri_ids) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t prim = sorted_prim[k];
    tris_out[k] = tris_in[prim];
    tri_ids[k] = prim;
}
__global__ void k_iota(uint32_t* __restrict__ out, uint32_t n) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = k;
}

// ---- 4-wide tree from the binary one --------------------------------------------------------------------------------
// One workgroup walks the binary tree breadth first. A wide node starts from a binary node's two children and, while it
// has a free slot, replaces the child with the largest surface area that can still be opened by that child's two
// children (a greedy surface-area collapse). In a BLAS a subtree of at most `max_leaf` triangles is not opened: it
// becomes one leaf (the triangles below a binary node are consecutive in the sorted order, `ranges`). The nodes of one
// level get consecutive numbers after those of the level above, a node's own children consecutive ones.
//   work[]  : per wide node, x = the binary node it starts from, y = stack entries a walk holds when it arrives there
//   info[0] = number of wide nodes, [1] = levels, [2] = reference of the root, [3] = stack entries a walk can need
// Conservative boxes: child boxes are grown by 2^-20 of the node's largest coordinate before they are put on the grid
// (the ray side evaluates origin + q * step as two fused operations, not one), and every stored corner is decoded with
// fmaf(q, step, origin) and moved outward until it encloses the real one.
struct CollapseSlot {
    uint32_t ref;
    float lo[3], hi[3];
};
PT_DEV float slot_area(const CollapseSlot& c) {
    const float dx = c.hi[0] - c.lo[0], dy = c.hi[1] - c.lo[1], dz = c.hi[2] - c.lo[2];
    return dx * dy + dy * dz + dz * dx;
}
// Puts the boxes of a wide node's n children on the node's grid: origin, step exponents and the 8-bit corners.
PT_DEV void wnode_set_grid(WNode& N, const CollapseSlot* sl, int n) {
    float big = 0.0f;
    for (int j = 0; j < n; ++j)
        for (int a = 0; a < 3; ++a) big = fmaxf(big, fmaxf(fabsf(sl[j].lo[a]), fabsf(sl[j].hi[a])));
    const float pad = fminf(big, 1.7e38f) * 0x1p-20f;
    uint32_t exps = 0;
    for (int a = 0; a < 3; ++a) {
        float mn = 3.0e38f, mx = -3.0e38f;
        for (int j = 0; j < n; ++j) { mn = fminf(mn, sl[j].lo[a] - pad); mx = fmaxf(mx, sl[j].hi[a] + pad); }
        mn = fmaxf(mn, -1.7e38f);   // ("unbounded" boxes of objects with non-finite transforms: keep the extent finite)
        mx = fminf(mx, 1.7e38f);
        const float o = mn;
        const float t = (mx - o) * (1.0f / 253.0f);
        int e = -100;
        if (t > 0.0f) { (void)frexpf(t, &e); }   // t = m * 2^e, m in [0.5, 1): 2^e >= t
        e = min(max(e, -100), 126);
        const float step = ldexpf(1.0f, e), inv_step = ldexpf(1.0f, -e);
        uint32_t wlo = 0, whi = 0;
        for (int j = 0; j < 4; ++j) {
            int ql = 255, qh = 0;   // empty slot: inverted
            if (j < n) {
                const float lo = fmaxf(sl[j].lo[a] - pad, -1.7e38f), hi = fminf(sl[j].hi[a] + pad, 1.7e38f);
                ql = (int)fminf(fmaxf(floorf((lo - o) * inv_step), 0.0f), 255.0f);
                while (ql > 0 && fmaf((float)ql, step, o) > lo) --ql;
                qh = (int)fminf(fmaxf(ceilf((hi - o) * inv_step), 0.0f), 255.0f);
                while (qh < 255 && fmaf((float)qh, step, o) < hi) ++qh;
            }
            wlo |= (uint32_t)ql << (8 * j);
            whi |= (uint32_t)qh << (8 * j);
        }
        N.origin[a] = o;
        N.qlo[a] = wlo;
        N.qhi[a] = whi;
        exps |= (uint32_t)(e + 127) << (8 * a);
    }
    N.exps = exps;
}
// One wide node: work item `item` of node number w. Children that become nodes get consecutive numbers from *next;
// *need collects the largest number of stack entries held.
PT_DEV void collapse_one(uint32_t w, const uint2 item, const BvhNode* __restrict__ bnodes, uint32_t bnode_base,
                         const uint2* __restrict__ ranges, uint32_t max_leaf, WNode* __restrict__ wnodes, uint32_t wnode_base,
                         uint2* __restrict__ work, uint32_t* next, uint32_t* need) {
    auto count_below = [&](uint32_t ref) -> uint32_t {   // primitives below a binary child reference
        if (ref & PT_LEAF_BIT) return 1u;
        if (!ranges) return 0xFFFFFFFFu;
        const uint2 r = ranges[ref - bnode_base];
        return r.y - r.x + 1u;
    };
    CollapseSlot sl[4];
    int n = 2;
    {
        const BvhNode B = bnodes[item.x];
        sl[0].ref = __float_as_uint(B.lo_l.w); sl[0].lo[0] = B.lo_l.x; sl[0].lo[1] = B.lo_l.y; sl[0].lo[2] = B.lo_l.z;
        sl[0].hi[0] = B.hi_l.x; sl[0].hi[1] = B.hi_l.y; sl[0].hi[2] = B.hi_l.z;
        sl[1].ref = __float_as_uint(B.lo_r.w); sl[1].lo[0] = B.lo_r.x; sl[1].lo[1] = B.lo_r.y; sl[1].lo[2] = B.lo_r.z;
        sl[1].hi[0] = B.hi_r.x; sl[1].hi[1] = B.hi_r.y; sl[1].hi[2] = B.hi_r.z;
    }
    while (n < 4) {
        int best = -1;
        float best_area = -1.0f;
        for (int j = 0; j < n; ++j) {
            if (sl[j].ref & PT_LEAF_BIT) continue;
            if (ranges && count_below(sl[j].ref) <= max_leaf) continue;
            const float a = slot_area(sl[j]);
            if (!(a <= best_area)) { best_area = a; best = j; }   // (a NaN area is opened too)
        }
        if (best < 0) break;
        const BvhNode B = bnodes[sl[best].ref - bnode_base];
        sl[best].ref = __float_as_uint(B.lo_l.w); sl[best].lo[0] = B.lo_l.x; sl[best].lo[1] = B.lo_l.y; sl[best].lo[2] = B.lo_l.z;
        sl[best].hi[0] = B.hi_l.x; sl[best].hi[1] = B.hi_l.y; sl[best].hi[2] = B.hi_l.z;
        sl[n].ref = __float_as_uint(B.lo_r.w); sl[n].lo[0] = B.lo_r.x; sl[n].lo[1] = B.lo_r.y; sl[n].lo[2] = B.lo_r.z;
        sl[n].hi[0] = B.hi_r.x; sl[n].hi[1] = B.hi_r.y; sl[n].hi[2] = B.hi_r.z;
        ++n;
    }
    // children: leaves keep / get their leaf reference, the others become wide nodes of the next level
    uint32_t n_inner = 0;
    for (int j = 0; j < n; ++j)
        if (!(sl[j].ref & PT_LEAF_BIT) && !(ranges && count_below(sl[j].ref) <= max_leaf)) ++n_inner;
    const uint32_t base = n_inner ? atomicAdd(next, n_inner) : 0u;
    const uint32_t held = item.y + (uint32_t)n - 1u;   // entries on the stack while a walk is below this node
    atomicMax(need, held);
    WNode N;
    uint32_t r = 0;
    for (int j = 0; j < 4; ++j) {
        if (j >= n) { N.child[j] = PT_NONE; continue; }
        const uint32_t ref = sl[j].ref;
        if (ref & PT_LEAF_BIT) {
            N.child[j] = ref;   // TLAS: the object; BLAS: one triangle at this position (count - 1 = 0)
        } else if (ranges && count_below(ref) <= max_leaf) {
            const uint2 rg = ranges[ref - bnode_base];
            N.child[j] = PT_LEAF_BIT | ((rg.y - rg.x) << PT_WLEAF_COUNT_SHIFT) | rg.x;
        } else {
            work[base + r] = make_uint2(ref - bnode_base, held);
            N.child[j] = wnode_base + base + r;
            ++r;
        }
    }
    wnode_set_grid(N, sl, n);
    N._pad[0] = item.x;   // (build bookkeeping: the binary node this one started from)
    N._pad[1] = held;
    wnodes[w] = N;
}
// Small trees: one workgroup does all levels.
__global__ __launch_bounds__(1024) void k_collapse(const BvhNode* __restrict__ bnodes, uint32_t bnode_base, uint32_t n_prims,
                                                   const uint2* __restrict__ ranges, uint32_t max_leaf,
                                                   WNode* __restrict__ wnodes, uint32_t wnode_base, uint2* __restrict__ work,
                                                   uint32_t* __restrict__ info) {
    __shared__ uint32_t s_begin, s_end, s_next, s_need, s_levels;
    const uint32_t tid = threadIdx.x;
    if (n_prims < 2u || (ranges && n_prims <= max_leaf)) {   // no node at all: the root is a leaf
        if (tid == 0) {
            info[0] = 0; info[1] = 0; info[3] = 0;
            info[2] = ranges ? (PT_LEAF_BIT | ((n_prims - 1u) << PT_WLEAF_COUNT_SHIFT)) : (PT_LEAF_BIT | 0u);
        }
        return;
    }
    if (tid == 0) {
        work[0] = make_uint2(0u, 0u);
        s_begin = 0; s_end = 1; s_next = 1; s_need = 0; s_levels = 0;
    }
    __syncthreads();
    for (;;) {
        const uint32_t begin = s_begin, end = s_end;
        if (begin == end) break;
        for (uint32_t w = begin + tid; w < end; w += blockDim.x)
            collapse_one(w, work[w], bnodes, bnode_base, ranges, max_leaf, wnodes, wnode_base, work, &s_next, &s_need);
        __syncthreads();
        if (tid == 0) { s_begin = end; s_end = s_next; s_levels++; }
        __syncthreads();
    }
    if (tid == 0) {
        info[0] = s_next;
        info[1] = s_levels;
        info[2] = wnode_base;
        info[3] = s_need;
    }
}
// Large trees: one launch per level over many workgroups. state: [0] next free node number, [1] stack need, [2] workgroups
// that have finished the level, [3] -, [4 + 2 L], [5 + 2 L] first and end node of level L. The workgroup that finishes a
// level last sets up the next level's range and brings info[] up to date, so that after the last launch info[] is what
// k_collapse would have left (launches beyond the tree's last level find an empty range and do nothing).
__global__ void k_collapse_begin(uint2* __restrict__ work, uint32_t* __restrict__ state) {
    work[0] = make_uint2(0u, 0u);
    state[0] = 1u;
    state[4] = 0u;
    state[5] = 1u;
}
__global__ __launch_bounds__(256) void k_collapse_level(const BvhNode* __restrict__ bnodes, uint32_t bnode_base, const uint2* __restrict__ ranges,
                                                        uint32_t max_leaf, WNode* __restrict__ wnodes, uint32_t wnode_base,
                                                        uint2* __restrict__ work, uint32_t* __restrict__ state, uint32_t level,
                                                        uint32_t* __restrict__ info) {
    const uint32_t begin = state[4 + 2 * level], end = state[5 + 2 * level];   // written before this launch began
    if (begin == end) return;
    for (uint32_t w = begin + blockIdx.x * blockDim.x + threadIdx.x; w < end; w += gridDim.x * blockDim.x)
        collapse_one(w, work[w], bnodes, bnode_base, ranges, max_leaf, wnodes, wnode_base, work, &state[0], &state[1]);
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const uint32_t done = __hip_atomic_fetch_add(&state[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1u) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const uint32_t next = __hip_atomic_load(&state[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t need = __hip_atomic_load(&state[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            state[4 + 2 * (level + 1u)] = end;
            state[5 + 2 * (level + 1u)] = next;
            __hip_atomic_store(&state[2], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            info[0] = next;
            info[1] = level + 1u;
            info[2] = wnode_base;
            info[3] = need;
        }
    }
}

// Refit of a TLAS's wide tree after k_fit has fitted the binary tree it was made from again: the topology (children,
// numbering, stack need) stays, every node takes its children's new boxes -- from the binary node a child node started
// from, or from the object's box for a leaf -- and puts them on a new grid. One thread per node, no order between them.
__global__ void k_requantise(WNode* __restrict__ wnodes, uint32_t wnode_base, uint32_t n_nodes, const BvhNode* __restrict__ bnodes,
                             const float4* __restrict__ box_lo, const float4* __restrict__ box_hi) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    WNode N = wnodes[i];
    CollapseSlot sl[4];
    int n = 0;
    for (int j = 0; j < 4; ++j) {   // (children fill the slots from 0 up)
        const uint32_t ref = N.child[j];
        if (ref == PT_NONE) break;
        sl[j].ref = ref;
        if (ref & PT_LEAF_BIT) {
            const float4 a = box_lo[ref & ~PT_LEAF_BIT], b = box_hi[ref & ~PT_LEAF_BIT];
            sl[j].lo[0] = a.x; sl[j].lo[1] = a.y; sl[j].lo[2] = a.z;
            sl[j].hi[0] = b.x; sl[j].hi[1] = b.y; sl[j].hi[2] = b.z;
        } else {
            const BvhNode B = bnodes[wnodes[ref - wnode_base]._pad[0]];
            sl[j].lo[0] = fminf(B.lo_l.x, B.lo_r.x); sl[j].lo[1] = fminf(B.lo_l.y, B.lo_r.y); sl[j].lo[2] = fminf(B.lo_l.z, B.lo_r.z);
            sl[j].hi[0] = fmaxf(B.hi_l.x, B.hi_r.x); sl[j].hi[1] = fmaxf(B.hi_l.y, B.hi_r.y); sl[j].hi[2] = fmaxf(B.hi_l.z, B.hi_r.z);
        }
        n = j + 1;
    }
    wnode_set_grid(N, sl, n);
    WNode* W = &wnodes[i];   // (only the box words: another thread may be reading this node's bookkeeping word)
    for (int a = 0; a < 3; ++a) { W->origin[a] = N.origin[a]; W->qlo[a] = N.qlo[a]; W->qhi[a] = N.qhi[a]; }
    W->exps = N.exps;
}

// Checks a wide tree: every child box, decoded the way the comment of WNode says, must enclose what lies below it (the
// primitives' boxes for a leaf; for a node, the box of the binary subtree it was made from, which k_check_bvh in turn
// checks against the primitives), and the leaves must hold every primitive
// exactly once (counted into errors[1], compared with the number of primitives by the caller). errors[0] counts
// violations. prim_of_pos (BLAS): the primitive stored at a sorted position; null (TLAS): leaves name primitives directly.
PT_DEV void wnode_child_box(const WNode& N, int j, float lo[3], float hi[3]) {
    for (int a = 0; a < 3; ++a) {
        const float step = __uint_as_float(((N.exps >> (8 * a)) & 0xFFu) << 23);
        lo[a] = fmaf((float)((N.qlo[a] >> (8 * j)) & 0xFFu), step, N.origin[a]);
        hi[a] = fmaf((float)((N.qhi[a] >> (8 * j)) & 0xFFu), step, N.origin[a]);
    }
}
__global__ void k_check_wide(const WNode* __restrict__ wnodes, uint32_t wnode_base, uint32_t n_nodes, const BvhNode* __restrict__ bnodes,
                             const float4* __restrict__ box_lo, const float4* __restrict__ box_hi, uint32_t first,
                             const uint32_t* __restrict__ prim_of_pos, uint32_t* __restrict__ errors) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_nodes) return;
    const WNode N = wnodes[i];
    uint32_t leaves = 0, bad = 0;
    for (int j = 0; j < 4; ++j) {
        const uint32_t ref = N.child[j];
        if (ref == PT_NONE) continue;
        float lo[3], hi[3];
        wnode_child_box(N, j, lo, hi);
        if (ref & PT_LEAF_BIT) {
            const uint32_t pos = prim_of_pos ? (ref & PT_WLEAF_POS_MASK) : (ref & ~PT_LEAF_BIT);
            const uint32_t cnt = prim_of_pos ? (((ref & ~PT_LEAF_BIT) >> PT_WLEAF_COUNT_SHIFT) + 1u) : 1u;
            for (uint32_t k = 0; k < cnt; ++k) {
                const uint32_t prim = prim_of_pos ? prim_of_pos[pos + k] : pos;
                const float4 a = box_lo[first + prim], b = box_hi[first + prim];
                if (!(lo[0] <= fmaxf(a.x, -1.7e38f) && lo[1] <= fmaxf(a.y, -1.7e38f) && lo[2] <= fmaxf(a.z, -1.7e38f) &&
                      hi[0] >= fminf(b.x, 1.7e38f) && hi[1] >= fminf(b.y, 1.7e38f) && hi[2] >= fminf(b.z, 1.7e38f))) ++bad;
            }
            leaves += cnt;
        } else {
            const uint32_t c = ref - wnode_base;
            if (c >= n_nodes || c <= i) { ++bad; continue; }   // (breadth-first numbering: children come later)
            const BvhNode B = bnodes[wnodes[c]._pad[0]];
            const float clo[3] = {fminf(B.lo_l.x, B.lo_r.x), fminf(B.lo_l.y, B.lo_r.y), fminf(B.lo_l.z, B.lo_r.z)};
            const float chi[3] = {fmaxf(B.hi_l.x, B.hi_r.x), fmaxf(B.hi_l.y, B.hi_r.y), fmaxf(B.hi_l.z, B.hi_r.z)};
            // (boxes beyond +-1.7e38 are stored clipped to that: objects whose transform is not finite)
            if (!(lo[0] <= fmaxf(clo[0], -1.7e38f) && lo[1] <= fmaxf(clo[1], -1.7e38f) && lo[2] <= fmaxf(clo[2], -1.7e38f) &&
                  hi[0] >= fminf(chi[0], 1.7e38f) && hi[1] >= fminf(chi[1], 1.7e38f) && hi[2] >= fminf(chi[2], 1.7e38f))) ++bad;
        }
    }
    if (bad) atomicAdd(&errors[0], bad);
    if (leaves) atomicAdd(&errors[1], leaves);
}

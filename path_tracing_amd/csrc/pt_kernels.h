// pt_kernels.h — the kernels of a frame.
//
//   k_primary  one lane per pixel: seed, camera ray, closest hit (the query stage, reference src/query.wgsl:55-147,
//              for the ray of src/raytrace.wgsl:212-217). The primary ray of a pixel is the same for all of its
//              samples (src/raytrace.wgsl:223 restarts from the same origin/direction, no jitter), so it is traced
//              once. Pixels whose primary hit ends every path at once (miss, or black-albedo emitter) are finished
//              here; every other pixel is appended, with its hit record, to the path list.
//   k_paths    persistent waves over the path list: the sample x bounce loop of src/raytrace.wgsl:221-273 with the
//              reference's per-pixel order of RNG draws and f32 accumulations, one pixel per lane at a time.
//              The work queue holds (sweep, chunk) items, sweep-major: a chunk is 64 consecutive list entries, a
//              sweep is a slice of the samples (half of them, then a quarter, ...). A pixel's state (RNG, radiance
//              sum, ray count) is parked in its list entry between sweeps; the order of operations per pixel is
//              unchanged. The slices get shorter towards the end of the launch, so the waves run out of work within
//              one short slice of each other instead of one whole pixel.
//   k_query    hit_all for a batch of rays (tests, tools).
//   k_accumulate  running mean over frames.
#pragma once

#include <hip/hip_fp16.h>

#include <type_traits>

#include "pt_trace.h"

#ifndef PT_PATHS_MIN_WAVES
#define PT_PATHS_MIN_WAVES 4   // waves per SIMD k_paths is compiled for (caps its VGPR allocation)
#endif
#ifndef PT_PATHS_MIN_WAVES_LIST
#define PT_PATHS_MIN_WAVES_LIST 4   // ... its object-list variants (PT_MODE_LIST, PT_MODE_LIST_LB)
#endif
// ... and their ROOMY builds: twice the registers per lane and half the waves per SIMD, for frames that are bound by a
// few long chains of dependent rays and cannot fill more waves anyway (no register of the round lives in scratch memory,
// and the compiler has room to overlap independent work within a lane)
#define PT_PATHS_MIN_WAVES_LIST_ROOMY 2
#ifndef PT_PATHS_MIN_WAVES_TLAS
#define PT_PATHS_MIN_WAVES_TLAS 4   // ... its PT_MODE_TLAS variant
#endif

// One entry of the path list: what k_primary found for a pixel that needs paths. 64 B, constant during k_paths.
struct PathRec {
    float4 a;  // bits(pix), bits(material of the primary hit), hit uv.x, hit uv.y
    float4 b;  // hit p.xyz, hit n.x
    float4 c;  // hit n.y, hit n.z, primary dir.x, primary dir.y
    float4 d;  // primary dir.z, bits(rng seed x), bits(seed y), bits(seed z)
};
static_assert(sizeof(PathRec) == 64, "PathRec");

// State of a pixel parked between two sweeps, one per list entry: rng x y z, rays so far, radiance sum x y z, 0.
// Written by one workgroup and read by another (possibly on another XCD, whose L2 is not coherent with the
// writer's) inside one launch, so it only moves through loads and stores that bypass the non-coherent cache
// levels; the chunk's flag is published after the stores have been acknowledged.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct PathCarry {
    u32x4 lo, hi;
};
static_assert(sizeof(PathCarry) == 32, "PathCarry");
// 16-byte accesses that go past the non-coherent cache levels (sc0 sc1: system scope), like device-scope atomic
// loads and stores of a dword do; the load waits for its data.
PT_DEV void carry_store(PathCarry* p, u32x4 lo, u32x4 hi) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc0 sc1"
                 :: "v"(p), "v"(lo), "v"(hi) : "memory");
}
PT_DEV void carry_load(const PathCarry* p, u32x4& lo, u32x4& hi) {
    asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc0 sc1\n\ts_waitcnt vmcnt(0)"
                 : "=&v"(lo), "=&v"(hi) : "v"(p) : "memory");
}

struct PathParams {
    float origin[3];
    float sky[3];
    uint32_t bounces, samples;
    uint32_t chunk;       // path numbers a wave takes from the work counter at a time (multiple of 64)
    uint32_t refill_min;  // a wave hands out new pixels only when at least this many of its lanes are idle (1..64)
    uint32_t walk_quota;    // PT_MODE_TLAS: lanes that have to finish their walk before a round's trace step ends (1..64)
    uint32_t n_sweeps;      // the samples of a pixel are run in this many consecutive slices (1..PT_MAX_SWEEPS)
    uint32_t sweep_end[8];  // slice j runs samples [sweep_end[j-1], sweep_end[j]); sweep_end[n_sweeps-1] == samples
    uint32_t split_chunks;  // 1: the most expensive chunks are handed out in parts (ChunkOrderInfo::n_split)
    uint32_t tlas_top_n;    // PT_MODE_TLAS: nodes of the TLAS (its top levels) every workgroup copies into LDS
    uint32_t carry_acquire;    // 1: an agent-scope acquire before a chunk's parked state is read (see the sweep hand-over in k_paths)
    uint32_t* stack_overflow;  // PT_MODE_TLAS: traversal stack entries beyond the `stack_depth` kept in LDS, [entry][lane of the grid]
};
#define PT_MAX_SWEEPS 8

// Where each scene array sits in the LDS copy (byte offsets from the start of the staged region, multiples of 16)
// and how many bytes it has. Small scenes only: see lds_scene_layout() on the host side.
struct LdsScene {
    uint32_t off_obj_slab, off_obj_lo, off_obj_hi, off_objects, off_l2w, off_meshes, off_tris, off_tri_ids, off_attrs, off_blas, off_materials, off_textures;
    uint32_t bytes_obj_box, bytes_objects, bytes_l2w, bytes_meshes, bytes_tris, bytes_tri_ids, bytes_attrs, bytes_blas, bytes_materials, bytes_textures;
    uint32_t total_bytes;
};

PT_DEV void lds_copy16(uint32_t* dst_words, const void* src, uint32_t bytes) {
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst_words);
    for (uint32_t i = threadIdx.x; i < bytes / 16u; i += blockDim.x) d4[i] = s4[i];
}

// Copy every scene record array into LDS and point the view at the copies. All later reads of objects, boxes, meshes,
// triangles, BLAS nodes, materials and texture descriptors are then LDS reads (texels stay in global memory).
PT_DEV void stage_scene_in_lds(SceneView& sc, const LdsScene& ls, uint32_t* lds) {
    lds_copy16(lds + ls.off_obj_lo / 4, sc.obj_lo, ls.bytes_obj_box);
    lds_copy16(lds + ls.off_obj_hi / 4, sc.obj_hi, ls.bytes_obj_box);
    lds_copy16(lds + ls.off_objects / 4, sc.objects, ls.bytes_objects);
    lds_copy16(lds + ls.off_l2w / 4, sc.obj_l2w, ls.bytes_l2w);
    lds_copy16(lds + ls.off_meshes / 4, sc.meshes, ls.bytes_meshes);
    lds_copy16(lds + ls.off_tris / 4, sc.tris, ls.bytes_tris);
    for (uint32_t i = threadIdx.x; i < ls.bytes_tri_ids / 4u; i += blockDim.x) lds[ls.off_tri_ids / 4 + i] = sc.tri_ids[i];
    lds_copy16(lds + ls.off_attrs / 4, sc.tri_attrs, ls.bytes_attrs);
    lds_copy16(lds + ls.off_blas / 4, sc.blas_nodes, ls.bytes_blas);
    lds_copy16(lds + ls.off_materials / 4, sc.materials, ls.bytes_materials);
    lds_copy16(lds + ls.off_textures / 4, sc.textures, ls.bytes_textures);
    {   // world boxes once more, in the (lo, hi, lo, 0)-per-axis form read by slab_test_sel
        float* slab = reinterpret_cast<float*>(lds + ls.off_obj_slab / 4);
        const uint32_t n_obj = ls.bytes_obj_box / 16u;
        for (uint32_t i = threadIdx.x; i < n_obj * 3u; i += blockDim.x) {
            const uint32_t o = i / 3u, axis = i - o * 3u;
            const float lo = reinterpret_cast<const float*>(&sc.obj_lo[o])[axis], hi = reinterpret_cast<const float*>(&sc.obj_hi[o])[axis];
            float* d = slab + o * 12u + axis * 4u;
            d[0] = lo; d[1] = hi; d[2] = lo; d[3] = 0.0f;
        }
    }
    __syncthreads();
    sc.obj_slab = reinterpret_cast<const float*>(lds + ls.off_obj_slab / 4);
    sc.obj_lo = reinterpret_cast<const float4*>(lds + ls.off_obj_lo / 4);
    sc.obj_hi = reinterpret_cast<const float4*>(lds + ls.off_obj_hi / 4);
    sc.objects = reinterpret_cast<const DObject*>(lds + ls.off_objects / 4);
    sc.obj_l2w = reinterpret_cast<const Affine*>(lds + ls.off_l2w / 4);
    sc.meshes = reinterpret_cast<const DMesh*>(lds + ls.off_meshes / 4);
    sc.tris = reinterpret_cast<const DTri*>(lds + ls.off_tris / 4);
    sc.tri_ids = lds + ls.off_tri_ids / 4;
    sc.tri_attrs = reinterpret_cast<const DTriAttr*>(lds + ls.off_attrs / 4);
    sc.blas_nodes = reinterpret_cast<const WNode*>(lds + ls.off_blas / 4);
    sc.materials = reinterpret_cast<const DMaterial*>(lds + ls.off_materials / 4);
    sc.textures = reinterpret_cast<const DTexture*>(lds + ls.off_textures / 4);
}

// One mesh's BLAS (and, when they fit too, its triangle records) to be copied into LDS by every k_paths workgroup.
struct LdsBlas {
    uint32_t mesh, node_base, n_nodes, tri_first, n_tris;   // n_tris == 0: triangles stay in global memory
};
__host__ __device__ constexpr uint32_t lds_blas_bytes(uint32_t n_nodes, uint32_t n_tris) {
    return n_nodes * (uint32_t)sizeof(WNode) + n_tris * (uint32_t)(sizeof(DTri) + sizeof(uint32_t));
}
PT_DEV void stage_blas_in_lds(SceneView& sc, const LdsBlas& lb, uint32_t* lds) {
    lds_copy16(lds, sc.blas_nodes + lb.node_base, lb.n_nodes * (uint32_t)sizeof(WNode));
    uint32_t* tris = lds + lb.n_nodes * (uint32_t)(sizeof(WNode) / 4);
    uint32_t* ids = tris + lb.n_tris * (uint32_t)(sizeof(DTri) / 4);
    if (lb.n_tris) {
        lds_copy16(tris, sc.tris + lb.tri_first, lb.n_tris * (uint32_t)sizeof(DTri));
        for (uint32_t i = threadIdx.x; i < lb.n_tris; i += blockDim.x) ids[i] = sc.tri_ids[lb.tri_first + i];
    }
    __syncthreads();
    sc.lb_nodes = reinterpret_cast<const WNode*>(lds);
    sc.lb_tris = lb.n_tris ? reinterpret_cast<const DTri*>(tris) : nullptr;
    sc.lb_tri_ids = lb.n_tris ? ids : nullptr;
    sc.lb_mesh = lb.mesh;
    sc.lb_node_base = lb.node_base;
}
// The top levels of the TLAS (its first n nodes: the tree is numbered breadth first) copied into LDS.
PT_DEV void stage_tlas_top_in_lds(SceneView& sc, uint32_t n, uint32_t* lds) {
    if (n == 0u) return;
    lds_copy16(lds, sc.tlas_nodes, n * (uint32_t)sizeof(WNode));
    __syncthreads();
    sc.tlas_top = reinterpret_cast<const WNode*>(lds);
    sc.tlas_top_n = n;
}

PT_DEV uint32_t wave_sum_u32(uint32_t v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// True when a path that reaches this hit stops there without scattering: a miss, or a surface whose albedo is
// black (src/raytrace.wgsl:245-248). `c` is what the path adds: fma(ray_color = 1, emissive or sky, color = 0).
PT_DEV bool primary_is_terminal(const SceneView& sc, const HitInfo& h, v3 sky, v3& c) {
    c = sky;
    if (h.mat != PT_NONE) {
        const DMaterial m = sc.materials[h.mat];
        const v3 alb = V3(m.albedo[0], m.albedo[1], m.albedo[2]);
        if (!(dot3(alb, alb) < PT_EPSILON)) return false;
        c = V3(m.emissive[0], m.emissive[1], m.emissive[2]);
        if (m.emissive_texture != PT_NONE) c = sample_texture(sc, m.emissive_texture, h.u, h.v);
    }
    c = V3(pt_fma(1.0f, c.x, 0.0f), pt_fma(1.0f, c.y, 0.0f), pt_fma(1.0f, c.z, 0.0f));
    return true;
}

// ---- primary pass: one wave per 8x8 pixel tile, WAVES tiles per workgroup ---------------------------------------
// (16 where the traversal stacks leave room: the append costs one atomic per workgroup on a single address)
template <int MODE, bool STATS, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_primary(PT_SCENE_PARAMS, LdsScene ls, FrameParams fp, uint32_t stack_depth, float4* __restrict__ out,
                                                 uint32_t* __restrict__ aux_obj, uint32_t* __restrict__ aux_flat,
                                                 float* __restrict__ aux_t, uint32_t* __restrict__ aux_rays,
                                                 PathRec* __restrict__ recs, uint32_t* __restrict__ cost, uint32_t use_hint,
                                                 DCounters* __restrict__ counters) {
    PT_SCENE_VIEW_FROM_PARAMS()
    // (16-byte aligned: the workgroup's static LDS words come first, and the staged records are read 16 bytes at a time)
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mem[];  // [WAVES][stack_depth][64 lanes], then the staged scene (PT_MODE_LDS)
    if (MODE == PT_MODE_LDS) stage_scene_in_lds(sc, ls, s_mem + WAVES * stack_depth * 64);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // Which tile this wave takes. Whole frames with 16 waves: the workgroup covers a 4x4 group of tiles (32x32
    // pixels), so that its entries of the path list, and with them the chunks k_paths works through, come from one
    // compact patch of the image with similar path lengths. Otherwise: tiles in row-major order.
    uint32_t tx, ty;
    bool tile_ok;  // wave-uniform; a wave without a tile stays for the block-wide append below
    if (WAVES == 16 && fp.part_count == 1u) {
        const uint32_t groups_x = (fp.tiles_x + 3u) / 4u;
        const uint32_t gy = blockIdx.x / groups_x, gx = blockIdx.x - gy * groups_x;
        tx = gx * 4u + (wave & 3u);
        ty = gy * 4u + (wave >> 2);
        tile_ok = tx < fp.tiles_x && ty < fp.tiles_y;
    } else {
        const uint32_t local_tile = blockIdx.x * WAVES + wave;
        tile_ok = local_tile < fp.n_local_tiles;
        const uint32_t tile = local_tile * fp.part_count + fp.part_index;
        ty = tile / fp.tiles_x;
        tx = tile - ty * fp.tiles_x;
    }
    LdsStack stk;
    stk.base = &s_mem[wave * stack_depth * 64 + lane];

    const uint32_t px = tx * 8 + (lane & 7), py = ty * 8 + (lane >> 3);
    const bool valid = tile_ok && px < fp.width && py < fp.height;
    const uint32_t pix = py * fp.width + px;

    // fullscreen triangle at the pixel centre; src/raytrace.wgsl:210-217
    const float W = (float)fp.width, H = (float)fp.height;
    const float uvx = pt_div((float)px + 0.5f, W), uvy = pt_div((float)py + 0.5f, H);
    const float tp1 = fp.time + 1.0f;
    const float sx = (uvx * W) * tp1, sy = (uvy * H) * tp1;
    const uint32_t seed_x = pt_f2u(sx), seed_y = pt_f2u(sy), seed_z = seed_x ^ seed_y;
    const float dx = uvx * 2.0f - 1.0f;
    const float dy = (uvy * 2.0f - 1.0f) * -1.0f;
    const v3 origin = V3(fp.wfv[12], fp.wfv[13], fp.wfv[14]);
    v3 temp = V3(pt_fma(fp.vfc[4], dy, fp.vfc[0] * dx) + fp.vfc[8] + fp.vfc[12],
                 pt_fma(fp.vfc[5], dy, fp.vfc[1] * dx) + fp.vfc[9] + fp.vfc[13],
                 pt_fma(fp.vfc[6], dy, fp.vfc[2] * dx) + fp.vfc[10] + fp.vfc[14]);
    temp = normalize3(temp);
    const v3 dir0 = V3(pt_fma(fp.wfv[8], temp.z, pt_fma(fp.wfv[4], temp.y, fp.wfv[0] * temp.x)),
                       pt_fma(fp.wfv[9], temp.z, pt_fma(fp.wfv[5], temp.y, fp.wfv[1] * temp.x)),
                       pt_fma(fp.wfv[10], temp.z, pt_fma(fp.wfv[6], temp.y, fp.wfv[2] * temp.x)));

    const uint32_t samples = fp.samples;
    const float ns = (float)samples;
    if (!(fp.bounces > 0 && samples > 0)) {
        // no ray is ever traced: `samples` additions of zero, then the division (0/0 when samples == 0)
        if (valid) {
            out[pix] = make_float4(pt_div(0.0f, ns), pt_div(0.0f, ns), pt_div(0.0f, ns), 1.0f);
            aux_obj[pix] = PT_NONE; aux_flat[pix] = PT_NONE; aux_t[pix] = PT_T_MAX; aux_rays[pix] = 0u;
        }
        return;
    }

    LaneCounters cnt = {};
    Best b;
    closest_hit<MODE, STATS>(sc, origin, dir0, valid, b, stk, cnt);
    bool to_list = false;
    uint32_t hint = 0;  // rays beyond one per sample that this pixel took in the previous frame (0 when unknown)
    HitInfo h;
    h.mat = PT_NONE; h.p = V3(0, 0, 0); h.n = V3(0, 0, 0); h.u = 0.0f; h.v = 0.0f;
    if (valid) {
        if (STATS && b.obj != PT_NONE) cnt.hits++;
        h = finalize_hit<MODE != PT_MODE_LDS>(sc, origin, dir0, b);
        aux_obj[pix] = b.obj;
        aux_flat[pix] = (b.obj != PT_NONE) ? sc.objects[b.obj].tri_base + (b.tri - (MODE != PT_MODE_LDS ? sc.objects[b.obj].tri_first : 0u)) : PT_NONE;
        aux_t[pix] = b.t;
        v3 c;
        if (primary_is_terminal(sc, h, V3(fp.sky[0], fp.sky[1], fp.sky[2]), c)) {
            v3 pixel_color = V3(0, 0, 0);
            for (uint32_t k = 0; k < samples; ++k) pixel_color = add3(pixel_color, c);
            out[pix] = make_float4(pt_div(pixel_color.x, ns), pt_div(pixel_color.y, ns), pt_div(pixel_color.z, ns), 1.0f);
            aux_rays[pix] = samples;  // the reference traces the primary ray once per sample
        } else {
            to_list = true;
            if (use_hint) {
                // aux_rays still holds the previous frame's count for this pixel: rays + samples - 1 (k_paths writes it)
                const uint32_t prev = aux_rays[pix];
                hint = prev > 2u * samples ? prev - 2u * samples : 0u;
            }
        }
    }
    // append the workgroup's surviving pixels to the path list: one atomic per workgroup (a single counter takes
    // only ~90 atomics per microsecond, so one per wave would cost more than the tracing)
    __shared__ uint32_t s_count[WAVES], s_base;
    const unsigned long long m = __ballot(to_list);
    if (lane == 0) s_count[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t n = 0;
        for (int w = 0; w < WAVES; ++w) n += s_count[w];
        s_base = n ? atomicAdd(&counters->n_paths, n) : 0u;
    }
    __syncthreads();
    {
        uint32_t base = s_base;
        for (uint32_t w = 0; w < wave; ++w) base += s_count[w];
        if (to_list) {
            PathRec r;
            r.a = make_float4(__uint_as_float(pix), __uint_as_float(h.mat), h.u, h.v);
            r.b = make_float4(h.p.x, h.p.y, h.p.z, h.n.x);
            r.c = make_float4(h.n.y, h.n.z, dir0.x, dir0.y);
            r.d = make_float4(dir0.z, __uint_as_float(seed_x), __uint_as_float(seed_y), __uint_as_float(seed_z));
            const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            recs[pos] = r;
            cost[pos] = hint;
        }
    }
    // (the primary rays, one per pixel of the partition, are added to the ray count on the host)
    if (STATS) {
        const uint32_t a = wave_sum_u32(cnt.xforms), bb = wave_sum_u32(cnt.box_tests), c = wave_sum_u32(cnt.tri_tests),
                       e = wave_sum_u32(cnt.hits);
        if (lane == 0) {
            atomicAdd(&counters->xforms, (unsigned long long)a);
            atomicAdd(&counters->box_tests, (unsigned long long)bb);
            atomicAdd(&counters->tri_tests, (unsigned long long)c);
            atomicAdd(&counters->hits, (unsigned long long)e);
        }
    }
}

// ---- order of the chunks ---------------------------------------------------------------------------------------
// A chunk is PT_CHUNK consecutive list entries. k_chunk_keys gives every chunk a cost bin from the per-pixel hints
// (mean rays beyond one per sample, 1/8 ray per sample per bin, 32 bins); k_chunk_order (one workgroup) sorts the
// chunk numbers by descending bin with a counting sort and reports how many chunks count as expensive.
#define PT_CHUNK 64u
#define PT_COST_BINS 32u
#ifndef PT_HEAVY_BIN
#define PT_HEAVY_BIN 2u   // chunks from this bin up (0.25 extra rays per sample) are run in one piece, first
#endif
// The most expensive chunks of all (whole cost bins from the top, at most 1/PT_SPLIT_FRACTION of the chunks while other
// frames run on the device, 1/PT_SPLIT_FRACTION_ALONE when the frame has the device to itself and idle lanes cost
// nothing) are handed out in PT_SPLIT_PARTS parts of 16 entries, one part to a wave: their pixels are the longest chains of dependent rays of
// the frame, and a round of a wave that traces for 16 lanes is much shorter than one for 64 (it lasts as long as its
// slowest lane). The lanes left idle are affordable: those chains are what the rest of the chip ends up waiting for.
#ifndef PT_SPLIT_PARTS
#define PT_SPLIT_PARTS 4u
#endif
#ifndef PT_QUAD_TRACE
#define PT_QUAD_TRACE 1   // 1: the rays of such a portion are traced by four lanes each (closest_hit_list_quad)
#endif
#ifndef PT_SPLIT_FRACTION
#define PT_SPLIT_FRACTION 16u
#endif
#ifndef PT_SPLIT_FRACTION_ALONE
#define PT_SPLIT_FRACTION_ALONE 4u
#endif
// n_groups: how many workgroups of k_paths the frame can keep busy. A chunk's bin is its expected length (rounds of one
// wave); the frame cannot end before its longest chunk does, so  margin x (sum of the lengths) / (longest length)  waves
// finish it in about that time when the longest are started first, and more waves only hold LDS and wave slots that the
// kernels of other frames in flight (other contexts, other streams) could use. Workgroups beyond n_groups leave at once.
// The margin is tighter when such frames exist: g_frames_running counts the hint-ordered frames of this process whose
// k_paths has not finished on this device (k_chunk_order signs a frame in, the last wave of its k_paths signs it off).
// Short frames overlap only now and then even when the caller keeps several in flight, so the tighter margin stays in
// use for PT_SHARED_MEMORY_TICKS of the constant 100 MHz clock after the last frame that did see another one.
#define PT_SHARED_MEMORY_TICKS 2000000ull   // 20 ms
__device__ uint32_t g_frames_running;
__device__ unsigned long long g_last_shared_tick;
struct ChunkOrderInfo {
    uint32_t n_chunks, n_heavy, n_split, n_groups;
    uint32_t one_sweep;   // the frame shares the device: every chunk runs all its samples in one item (see k_chunk_order)
    uint32_t pad[3];
};
__global__ __launch_bounds__(256) void k_chunk_keys(const uint32_t* __restrict__ cost, const DCounters* __restrict__ counters,
                                                    uint32_t samples, uint32_t* __restrict__ chunk_bin) {
    const uint32_t total = counters->n_paths;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t c = blockIdx.x * 4u + wave;
    if (c * PT_CHUNK >= total) return;
    const uint32_t i = c * PT_CHUNK + lane;
    uint32_t v = (i < total) ? min(cost[i], 1u << 24) : 0u;   // (clamped: 64 of them are summed in 32 bits)
    uint32_t n = (i < total) ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) { v += __shfl_xor(v, off, 64); n += __shfl_xor(n, off, 64); }
    if (lane == 0) {
        const unsigned long long bin = ((unsigned long long)v * 8ull) / ((unsigned long long)n * samples);
        chunk_bin[c] = bin < PT_COST_BINS ? (uint32_t)bin : PT_COST_BINS - 1u;
    }
}
__global__ __launch_bounds__(1024) void k_chunk_order(const uint32_t* __restrict__ chunk_bin, DCounters* __restrict__ counters,
                                                      uint32_t* __restrict__ order, ChunkOrderInfo* __restrict__ info, uint32_t split_fraction_alone, uint32_t split_fraction_shared,
                                                      uint32_t margin_alone_8ths, uint32_t margin_shared_8ths, uint32_t min_groups,
                                                      uint32_t launch_waves, uint32_t split_load_8ths, uint32_t sweeps_rule) {
    __shared__ uint32_t cnt[PT_COST_BINS], start[PT_COST_BINS];
    const uint32_t total = counters->n_paths;
    const uint32_t n_chunks = (total + PT_CHUNK - 1u) / PT_CHUNK;
    if (threadIdx.x < PT_COST_BINS) cnt[threadIdx.x] = 0;
    __syncthreads();
    // One workgroup walks the bins of all chunks twice. Eight loads are in flight per thread (the passes are a chain of
    // memory round trips otherwise), and most chunks of a frame share a few bins: one atomic per wave and bin, not per chunk.
    const uint32_t lane = threadIdx.x & 63u;
    constexpr uint32_t BATCH = 8u;
    for (uint32_t c0 = 0; c0 < n_chunks; c0 += BATCH * 1024u) {
        uint32_t bins[BATCH];
#pragma unroll
        for (uint32_t k = 0; k < BATCH; ++k) {
            const uint32_t c = c0 + k * 1024u + threadIdx.x;
            bins[k] = c < n_chunks ? chunk_bin[c] : PT_NONE;
        }
#pragma unroll
        for (uint32_t k = 0; k < BATCH; ++k) {
            const uint32_t b = bins[k];
            unsigned long long todo = __ballot(b != PT_NONE);
            while (todo != 0ull) {
                const uint32_t leader = (uint32_t)__ffsll((long long)todo) - 1u;
                const uint32_t lb = __shfl(b, (int)leader, 64);
                const unsigned long long same = __ballot(b == lb);
                if (lane == leader) atomicAdd(&cnt[lb], (uint32_t)__popcll(same));
                todo &= ~same;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // other frames on the device? (see g_frames_running)
        const uint32_t others = atomicAdd(&g_frames_running, 1u);
        const unsigned long long now = __builtin_amdgcn_s_memrealtime();
        bool shared = others != 0u;
        if (shared) atomicMax(&g_last_shared_tick, now);
        else {
            const unsigned long long last = atomicMax(&g_last_shared_tick, 0ull);
            shared = last != 0ull && now - last < PT_SHARED_MEMORY_TICKS;
        }
        const uint32_t split_fraction = shared ? split_fraction_shared : split_fraction_alone;
        const uint32_t group_margin_8ths = shared ? margin_shared_8ths : margin_alone_8ths;
        // (and no more parts than one and a half times the waves of the launch: beyond that they only queue up)
        const uint32_t split_limit = min(n_chunks / split_fraction, launch_waves * 3u / (2u * PT_SPLIT_PARTS));
        uint32_t run = 0, heavy = 0, split = 0;
        unsigned long long work = 0, longest = 0;   // in 1/8 rays per sample: a chunk of bin b takes about 8 + b
        bool top = true;
        for (int b = (int)PT_COST_BINS - 1; b >= 0; --b) {   // most expensive bin first
            start[b] = run;
            run += cnt[b];
            work += (unsigned long long)cnt[b] * (8u + (uint32_t)b);
            if (cnt[b] && !longest) longest = 8u + (uint32_t)b;
            if ((uint32_t)b >= PT_HEAVY_BIN) {
                heavy = run;
                if (top && run <= split_limit) split = run; else if (cnt[b]) top = false;
            }
        }
        // Parts leave three lanes in four idle: worth it while the launch has waves to spare, i.e. while the frame is bound
        // by its longest chains; a frame with several times more work than the launch has waves is bound by throughput.
        const unsigned long long waves_needed = longest ? work / longest : 0ull;
        if (split_load_8ths && waves_needed * 8ull > (unsigned long long)launch_waves * split_load_8ths) split = 0;
        // Sample sweeps shorten the tail of a frame that has the device to itself. Beside other frames there is no tail to
        // shorten (their kernels fill the gaps; measured: the same time per frame with and without on every workload), and
        // a wave that polls for a chunk's previous sweep would do so while the hardware shares its compute units between
        // queues: frames that share the device run without sweeps, and so without any wait inside the kernel.
        // sweeps_rule 0: that; 1: sweeps whatever runs beside the frame; 2: the host has already given the frame one sweep.
        info->one_sweep = (sweeps_rule == 2u || (sweeps_rule == 0u && shared)) ? 1u : 0u;
        if (info->one_sweep) { heavy = 0; split = 0; }
        info->n_chunks = n_chunks;
        info->n_heavy = heavy;
        info->n_split = split;
        unsigned long long groups = 0xFFFFFFFFull;
        if (group_margin_8ths && longest) groups = (work * group_margin_8ths / (8ull * longest) + 3ull) / 4ull;   // four waves each
        info->n_groups = groups < min_groups ? min_groups : (groups > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)groups);
        counters->n_groups = info->n_groups;
        counters->waves_needed = (uint32_t)waves_needed;
        counters->n_split = split;
        counters->others_running = others | (shared ? 0x100u : 0u);
    }
    __syncthreads();
    for (uint32_t c0 = 0; c0 < n_chunks; c0 += BATCH * 1024u) {
        uint32_t bins[BATCH];
#pragma unroll
        for (uint32_t k = 0; k < BATCH; ++k) {
            const uint32_t c = c0 + k * 1024u + threadIdx.x;
            bins[k] = c < n_chunks ? chunk_bin[c] : PT_NONE;
        }
#pragma unroll
        for (uint32_t k = 0; k < BATCH; ++k) {
            const uint32_t b = bins[k];
            unsigned long long todo = __ballot(b != PT_NONE);
            while (todo != 0ull) {
                const uint32_t leader = (uint32_t)__ffsll((long long)todo) - 1u;
                const uint32_t lb = __shfl(b, (int)leader, 64);
                const unsigned long long same = __ballot(b == lb);
                uint32_t base = 0;
                if (lane == leader) base = atomicAdd(&start[lb], (uint32_t)__popcll(same));
                base = __shfl(base, (int)leader, 64);
                if (b == lb) order[base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull))] = c0 + k * 1024u + threadIdx.x;
                todo &= ~same;
            }
        }
    }
}

// ---- path pass -----------------------------------------------------------------------------------------------
// Persistent waves with per-lane refill. A lane owns one pixel at a time; across pixels nothing is shared. Every
// round of the loop
//   A. hands a new list entry to each idle lane (the wave owns a chunk of consecutive entries, i.e. pixels of a
//      few neighbouring tiles, and takes the next item from the work queue when it runs out),
//   B. shades: every lane that holds a hit shades it; a lane whose path ends moves on to the pixel's next sample
//      and shades the cached primary hit right away, until it holds a new ray or its samples are done,
//   C. traces one ray for every lane that has one.
// So the closest-hit code always runs with (nearly) all lanes, however uneven the path lengths are.
//
// Registers are the scarce resource (128 per lane at four waves per SIMD), so everything that is only touched when
// a path ends or a chunk changes lives in LDS: per lane the cached primary hit, the pixel's radiance sum, the
// samples left in its sweep, its list entry and pixel number; per wave the scheduling state.
#define PT_ST_IDLE 0u
#define PT_ST_SHADE 1u
#define PT_ST_TRACE 2u
#define PT_ST_RESTART 3u   // a new sample whose first hit, the cached primary hit, is shaded in the wave's next round
#ifndef PT_WAVE_PRIO_PASSES
#define PT_WAVE_PRIO_PASSES 2u  // object-test passes in a round from which a wave raises its issue priority
#endif
// per-lane LDS words
#define PT_SLOT_HIT 0        // primary hit record: object, p.xyz, n.xyz, uv (9 words)
#define PT_SLOT_DIR 9        // primary ray direction (3)
#define PT_SLOT_SUM 12       // radiance summed over the pixel's samples so far (3)
#define PT_SLOT_LEFT 15      // samples left in this sweep
#define PT_SLOT_ENTRY 16     // list entry; bit 31: this sweep ends the pixel
#define PT_SLOT_PIX 17       // pixel number
#define PT_LANE_SLOTS 18
// PT_MODE_TLAS keeps no copy of the primary hit in LDS (a restarting lane reads its list entry again); the twelve words
// hold what the lane does not need while it walks the tree instead: throughput, radiance, RNG state, bounce and ray
// counts. Left in registers they were spilled to scratch memory every round, 20 words out and 31 back per lane, on the
// same vector-memory path the walk's node fetches are bound by.
#ifndef PT_TLAS_PARK
#define PT_TLAS_PARK 2
#endif
// The same words hold the other half of the lane's state while it refills and shades: where its unfinished walk stands
// (node, stack height) and the closest hit so far. The two never have to be in registers at the same time.
#define PT_SLOT_PARK 0       // (PT_MODE_TLAS) during the walk: ray_color xyz, color xyz, rng xyz, bounce, rays of the pixel, rays of the lane
                             //                otherwise: walk node, stack height, closest hit t, object, triangle, u, v
// PT_MODE_LDS only (small scenes leave the room): the part of the scatter evaluation of the primary hit that is the
// same for every sample of the pixel (BrdfFrame: tangent frame, reflected ray, N.V, albedo), 13 more words per lane
#define PT_SLOT_FRAME 18
#define PT_LANE_SLOTS_LDS 31
__host__ __device__ constexpr uint32_t lane_slots(int mode) { return mode == PT_MODE_LDS ? PT_LANE_SLOTS_LDS : PT_LANE_SLOTS; }
// per-wave LDS words (scheduling state)
#define PT_SCHED_NEXT 0      // list entries of the current item not handed to lanes yet: [next, end)
#define PT_SCHED_END 1
#define PT_SCHED_ITEM 2      // queue item taken but not started (its chunk's previous sweep is still running elsewhere), or PT_NONE
#define PT_SCHED_REPORT 3    // chunk whose sweep this wave has to report as complete, or PT_NONE
#define PT_SCHED_SWEEP 4     // sweep of the current item
#define PT_SCHED_BEGIN 5     // its first sample
#define PT_SCHED_LEN 6       // its number of samples; bit 31: it ends the pixel
#define PT_SCHED_WORDS 8

template <int MODE, bool STATS, bool ROOMY = false>
__global__ __launch_bounds__(256, MODE == PT_MODE_TLAS ? PT_PATHS_MIN_WAVES_TLAS : MODE == PT_MODE_LDS ? PT_PATHS_MIN_WAVES : ROOMY ? PT_PATHS_MIN_WAVES_LIST_ROOMY : PT_PATHS_MIN_WAVES_LIST) void k_paths(PT_SCENE_PARAMS, LdsScene ls, LdsBlas lb, PathParams pp, uint32_t stack_depth,
                                                                    const PathRec* __restrict__ recs, PathCarry* __restrict__ carry,
                                                                    uint32_t* __restrict__ sweeps_done, const uint32_t* __restrict__ chunk_order,
                                                                    const ChunkOrderInfo* __restrict__ order_info,
                                                                    float4* __restrict__ out, uint32_t* __restrict__ aux_rays,
                                                                    DCounters* __restrict__ counters, uint4* __restrict__ wave_log) {
    if (order_info && blockIdx.x >= order_info->n_groups) {   // more workgroups than the frame has work for
        if (threadIdx.x == 0 && atomicAdd(&counters->waves_done, 4u) + 4u == 4u * gridDim.x) atomicSub(&g_frames_running, 1u);
        return;
    }
    PT_SCENE_VIEW_FROM_PARAMS()
    // [4 waves][stack_depth][64 lanes] stacks, [4 waves][lane_slots(MODE)][64 lanes], [4 waves][PT_SCHED_WORDS], staged scene
    constexpr uint32_t SLOTS = lane_slots(MODE);
    constexpr bool TLAS_PARK = MODE == PT_MODE_TLAS && PT_TLAS_PARK != 0;
    constexpr bool TLAS_PARK_WALK = MODE == PT_MODE_TLAS && PT_TLAS_PARK >= 2;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mem[];
    const uint32_t lds_fixed = 4 * (stack_depth + SLOTS) * 64 + 4 * PT_SCHED_WORDS;
    if (MODE == PT_MODE_LDS) stage_scene_in_lds(sc, ls, s_mem + lds_fixed);
    if (MODE == PT_MODE_LIST_LB) stage_blas_in_lds(sc, lb, s_mem + lds_fixed);
    if (MODE == PT_MODE_TLAS) stage_tlas_top_in_lds(sc, pp.tlas_top_n, s_mem + lds_fixed);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // (the TLAS walk keeps only the first stack_depth entries of its stack in LDS)
    typename std::conditional<MODE == PT_MODE_TLAS, SpillStack, LdsStack>::type stk;
    stk.base = &s_mem[wave * stack_depth * 64 + lane];
    if constexpr (MODE == PT_MODE_TLAS) {
        stk.overflow = pp.stack_overflow + (blockIdx.x * 256u + threadIdx.x);
        stk.lds_levels = (int)stack_depth;
        stk.stride = gridDim.x * 256u;
    }
    uint32_t* slot_u = &s_mem[4 * stack_depth * 64 + wave * SLOTS * 64 + lane];  // slot[k * 64]
    float* slot = reinterpret_cast<float*>(slot_u);
    uint32_t* sched = &s_mem[4 * (stack_depth + SLOTS) * 64 + wave * PT_SCHED_WORDS];
    if (lane == 0) {
        sched[PT_SCHED_NEXT] = 0; sched[PT_SCHED_END] = 0;
        sched[PT_SCHED_ITEM] = PT_NONE; sched[PT_SCHED_REPORT] = PT_NONE;
    }

    const v3 origin = V3(pp.origin[0], pp.origin[1], pp.origin[2]);
    const uint32_t bounces = pp.bounces;
    // (read once: the line that holds n_paths also holds the queue head, which every wave keeps hitting with atomics)
    const uint32_t total = counters->n_paths;  // written by k_primary, which ran before on the same stream
    const uint32_t n_sweeps = (order_info && order_info->one_sweep) ? 1u : pp.n_sweeps;
    const uint32_t n_heavy = (order_info && n_sweeps > 1u) ? order_info->n_heavy : 0u;
    // (not for scenes staged in LDS: their rounds are bound by instruction issue and cost the same for 16 lanes as for 64)
    const uint32_t n_split = (MODE != PT_MODE_LDS && order_info && n_sweeps > 1u && pp.split_chunks) ? order_info->n_split : 0u;   // <= n_heavy

    LaneCounters cnt = {};
    uint32_t lane_rays = 0;
    unsigned long long t_begin = 0, t_begin_real = 0;
    if (STATS) {
        t_begin = __builtin_amdgcn_s_memtime();
        t_begin_real = __builtin_amdgcn_s_memrealtime();  // constant 100 MHz, the same on every XCD
    }

    uint32_t st = PT_ST_IDLE, bounce = 0, nrays = 0;
    Rng rng = {0, 0, 0};
    v3 rpos = origin, rdir = V3(0, 0, 1);
    v3 ray_color = V3(1, 1, 1), color = V3(0, 0, 0);
    HitInfo h;
    h.mat = PT_NONE; h.p = V3(0, 0, 0); h.n = V3(0, 0, 0); h.u = 0.0f; h.v = 0.0f;
    // PT_MODE_TLAS: a lane's TLAS walk survives the rounds of the loop below (see wide_walk_resume)
    WalkState ws = {PT_WALK_DONE, 0};
    Best wb;
    wb.t = PT_T_MAX; wb.obj = PT_NONE; wb.tri = PT_NONE; wb.u = 0.0f; wb.v = 0.0f;
    auto park_walk = [&]() {
        float* pk = slot + PT_SLOT_PARK * 64;
        uint32_t* pku = slot_u + PT_SLOT_PARK * 64;
        pku[0 * 64] = ws.node; pku[1 * 64] = (uint32_t)ws.sp;
        pk[2 * 64] = wb.t; pku[3 * 64] = wb.obj; pku[4 * 64] = wb.tri; pk[5 * 64] = wb.u; pk[6 * 64] = wb.v;
    };
    if (TLAS_PARK_WALK) park_walk();
    bool quad_wave = false;  // wave-uniform: the wave holds a portion of a split chunk, its pixels in lanes 0, 4, 8 ... (quads)
    bool exhausted = false;  // wave-uniform: the work queue has run out
    bool waiting = false;    // wave-uniform: holding an item whose chunk is not ready yet
    uint32_t first_entry = 0xFFFFFFFFu, n_chunks = 0;  // STATS only
    unsigned long long item_t0 = 0, item_r0 = 0, worst_cycles = 0, worst_rounds = 0;  // STATS only: the wave's longest item
    uint32_t worst_item = 0, cur_item = 0;
    unsigned long long cyc_refill = 0, cyc_trace = 0, cyc_shade = 0, n_rounds = 0, lane_rounds = 0, n_passes = 0;  // STATS only
    uint32_t quad_rounds = 0;   // STATS only: trace steps this wave did with four lanes to a ray

    for (;;) {
        unsigned long long t0 = 0, t1 = 0, t2 = 0;
        if (STATS) t0 = __builtin_amdgcn_s_memtime();
        // ---- A. refill idle lanes ---------------------------------------------------------------------------
        {
            bool need = st == PT_ST_IDLE;
            unsigned long long m = __ballot(need);
            if ((uint32_t)__popcll(m) >= pp.refill_min && !exhausted) {  // else: wait until enough lanes are free (newcomers share a tile)
                // Queue: first every chunk once, most expensive first (chunk_order): the n_heavy expensive ones with
                // all their samples, so that the longest chains of dependent rays start at once and run through; the
                // others with their first sweep; then sweep after sweep over the others.
                const uint32_t n_chunks_total = (total + pp.chunk - 1u) / pp.chunk;
                const uint32_t n_light = n_chunks_total - n_heavy;
                const uint32_t n_items = n_chunks_total + (n_sweeps - 1u) * n_light + (PT_SPLIT_PARTS - 1u) * n_split;
                bool solo = false;   // the item just taken is a part of a split chunk: this wave takes nothing else with it
                uint32_t chunk_next = __builtin_amdgcn_readfirstlane(sched[PT_SCHED_NEXT]);
                uint32_t chunk_end = __builtin_amdgcn_readfirstlane(sched[PT_SCHED_END]);
                uint32_t report = __builtin_amdgcn_readfirstlane(sched[PT_SCHED_REPORT]);
                uint32_t sw_begin = __builtin_amdgcn_readfirstlane(sched[PT_SCHED_BEGIN]);
                uint32_t sw_len = __builtin_amdgcn_readfirstlane(sched[PT_SCHED_LEN]);
                if (report != PT_NONE && chunk_next == chunk_end && m == ~0ull) {
                    // every pixel of the chunk is parked: publish the sweep once the parking stores have been acknowledged
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const uint32_t sweep = __builtin_amdgcn_readfirstlane(sched[PT_SCHED_SWEEP]);
                    if (lane == 0) __hip_atomic_store(&sweeps_done[report], sweep + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    report = PT_NONE;
                }
                waiting = false;
                while (m != 0ull) {
                    if (chunk_next == chunk_end) {
                        uint32_t item = __builtin_amdgcn_readfirstlane(sched[PT_SCHED_ITEM]);
                        if (item == PT_NONE) {
                            uint32_t q = 0;
                            if (lane == 0) q = atomicAdd(&counters->next_path, 1u);
                            q = __builtin_amdgcn_readfirstlane(q);
                            if (q >= n_items) {
                                exhausted = true;
                                break;
                            }
                            item = q;
                        }
                        // the queue starts with the parts of the split chunks, then goes on as described above
                        uint32_t j = 0, cpos = item, part = PT_NONE;   // sweep, position in the chunk order
                        if (item < PT_SPLIT_PARTS * n_split) {
                            cpos = item / PT_SPLIT_PARTS;
                            part = item - cpos * PT_SPLIT_PARTS;
                        } else {
                            cpos = item - (PT_SPLIT_PARTS - 1u) * n_split;
                        }
                        if (cpos >= n_chunks_total) {
                            const uint32_t r = cpos - n_chunks_total;
                            j = 1u + r / n_light;
                            cpos = n_heavy + (r - (j - 1u) * n_light);
                        }
                        const uint32_t c = chunk_order ? chunk_order[cpos] : cpos;
                        if (j > 0u) {
                            uint32_t v = 0;
                            if (lane == 0) v = __hip_atomic_load(&sweeps_done[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            v = __builtin_amdgcn_readfirstlane(v);
                            if (v < j) {  // not yet: keep the item, run the lanes that have work, ask again next round
                                if (lane == 0) sched[PT_SCHED_ITEM] = item;
                                waiting = true;
                                break;
                            }
                            // Consumer side of the hand-over: one relaxed poll (above), one agent-scope acquire, then the
                            // loads. The parked state is stored and loaded with accesses that go past the non-coherent
                            // cache levels anyway (sc0 sc1), so the acquire is a second line of defence; it can be left out
                            // (PT_AMD_CARRY_ACQUIRE=0) to measure what it costs.
                            if (pp.carry_acquire) {
                                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            }
                            asm volatile("" ::: "memory");
                        }
                        if (lane == 0) {
                            sched[PT_SCHED_ITEM] = PT_NONE;
                            sched[PT_SCHED_SWEEP] = j;
                        }
                        sw_begin = j ? pp.sweep_end[j - 1u] : 0u;
                        const uint32_t sw_end = (cpos < n_heavy || n_sweeps == 1u) ? pp.samples : pp.sweep_end[j];
                        sw_len = (sw_end - sw_begin) | (sw_end >= pp.samples ? 0x80000000u : 0u);
                        chunk_next = c * pp.chunk;
                        chunk_end = (total - chunk_next < pp.chunk) ? total : chunk_next + pp.chunk;
                        solo = part != PT_NONE;   // (a part can be empty, the list's last chunk being short: the wave then goes on to another item)
                        if (solo) {
                            const uint32_t len = pp.chunk / PT_SPLIT_PARTS;
                            chunk_next = min(chunk_next + part * len, chunk_end);
                            chunk_end = min(chunk_next + len, chunk_end);
                        }
                        if (n_sweeps > 1u && cpos >= n_heavy) report = c;
                        if (STATS) {
                            if (n_chunks == 0) first_entry = chunk_next;
                            const unsigned long long now = __builtin_amdgcn_s_memtime();
                            if (n_chunks != 0 && now - item_t0 > worst_cycles) { worst_cycles = now - item_t0; worst_rounds = n_rounds - item_r0; worst_item = cur_item; }
                            item_t0 = now; item_r0 = n_rounds; cur_item = item;
                            n_chunks++;
                        }
                    }
                    const uint32_t avail = chunk_end - chunk_next;
                    const uint32_t want = (uint32_t)__popcll(m);
                    const uint32_t take = want < avail ? want : avail;
                    // (a portion of a split chunk goes to every fourth lane: the lanes between help with its rays, see
                    // closest_hit_list_quad)
                    // (only when every lane of the wave is idle, which whole-chunk refill guarantees: with a smaller refill
                    // threshold the quad leaders may still be busy, and the entries then go to the idle lanes one by one)
                    const bool quads = PT_QUAD_TRACE && solo && MODE != PT_MODE_LDS && MODE != PT_MODE_TLAS && m == ~0ull;
                    const uint32_t rank = quads ? ((lane & 3u) ? 64u : (lane >> 2)) : (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (take != 0u) quad_wave = quads;
                    if (need && rank < take) {
                        need = false;
                        const uint32_t entry = chunk_next + rank;
                        const PathRec* r = &recs[entry];
                        const float4 ra = r->a, rb = r->b, rc = r->c, rd = r->d;
                        h.mat = __float_as_uint(ra.y);
                        h.u = ra.z; h.v = ra.w;
                        h.p = V3(rb.x, rb.y, rb.z);
                        h.n = V3(rb.w, rc.x, rc.y);
                        rdir = V3(rc.z, rc.w, rd.x);
                        rng.x = __float_as_uint(rd.y); rng.y = __float_as_uint(rd.z); rng.z = __float_as_uint(rd.w);
                        if (!TLAS_PARK) {
                            slot[(PT_SLOT_HIT + 0) * 64] = ra.y;
                            slot[(PT_SLOT_HIT + 1) * 64] = h.p.x; slot[(PT_SLOT_HIT + 2) * 64] = h.p.y; slot[(PT_SLOT_HIT + 3) * 64] = h.p.z;
                            slot[(PT_SLOT_HIT + 4) * 64] = h.n.x; slot[(PT_SLOT_HIT + 5) * 64] = h.n.y; slot[(PT_SLOT_HIT + 6) * 64] = h.n.z;
                            slot[(PT_SLOT_HIT + 7) * 64] = h.u; slot[(PT_SLOT_HIT + 8) * 64] = h.v;
                            slot[(PT_SLOT_DIR + 0) * 64] = rdir.x; slot[(PT_SLOT_DIR + 1) * 64] = rdir.y; slot[(PT_SLOT_DIR + 2) * 64] = rdir.z;
                        }
                        nrays = 1;  // the primary ray, traced by k_primary
                        v3 sum = V3(0, 0, 0);
                        if (sw_begin != 0u) {  // resume
                            u32x4 lo, hi;
                            carry_load(&carry[entry], lo, hi);
                            rng.x = lo.x; rng.y = lo.y; rng.z = lo.z;
                            nrays = lo.w;
                            sum = V3(__uint_as_float(hi.x), __uint_as_float(hi.y), __uint_as_float(hi.z));
                        }
                        slot[(PT_SLOT_SUM + 0) * 64] = sum.x; slot[(PT_SLOT_SUM + 1) * 64] = sum.y; slot[(PT_SLOT_SUM + 2) * 64] = sum.z;
                        slot_u[PT_SLOT_LEFT * 64] = sw_len & 0x7FFFFFFFu;
                        slot_u[PT_SLOT_ENTRY * 64] = entry | (sw_len & 0x80000000u);
                        slot_u[PT_SLOT_PIX * 64] = __float_as_uint(ra.x);
                        if (MODE == PT_MODE_LDS) {
                            // the sample-independent part of scattering off the primary hit, once per pixel and sweep
                            const DMaterial pm = sc.materials[h.mat];
                            v3 N = h.n;
                            if (pm.normal_map_texture != PT_NONE) N = mul3(N, sample_texture(sc, pm.normal_map_texture, h.u, h.v));
                            const BrdfFrame f = brdf_frame(sc, rdir, N, h.u, h.v, pm);
                            float* fs = slot + PT_SLOT_FRAME * 64;
                            fs[0 * 64] = f.tt.x; fs[1 * 64] = f.tt.y; fs[2 * 64] = f.tt.z;
                            fs[3 * 64] = f.bb.x; fs[4 * 64] = f.bb.y; fs[5 * 64] = f.bb.z;
                            fs[6 * 64] = f.refl_om.x; fs[7 * 64] = f.refl_om.y; fs[8 * 64] = f.refl_om.z;
                            fs[9 * 64] = f.NdotV;
                            fs[10 * 64] = f.albedo.x; fs[11 * 64] = f.albedo.y; fs[12 * 64] = f.albedo.z;
                        }
                        bounce = 0;
                        ray_color = V3(1, 1, 1);
                        color = V3(0, 0, 0);
                        rpos = origin;
                        st = PT_ST_SHADE;
                    }
                    chunk_next += take;
                    m = __ballot(need);
                    if (report != PT_NONE && chunk_next == chunk_end) break;  // sweeps: one chunk per wave at a time
                    if (solo && take != 0u) break;
                }
                if (lane == 0) {
                    sched[PT_SCHED_NEXT] = chunk_next; sched[PT_SCHED_END] = chunk_end;
                    sched[PT_SCHED_REPORT] = report;
                    sched[PT_SCHED_BEGIN] = sw_begin; sched[PT_SCHED_LEN] = sw_len;
                }
            }
        }
        if (!__any(st != PT_ST_IDLE)) {
            if (exhausted) break;
            if (waiting) __builtin_amdgcn_s_sleep(16);
            continue;
        }

        // ---- B. shade (src/raytrace.wgsl:229-270) and advance (sample, bounce) ----------------------------------
        if (STATS) t1 = __builtin_amdgcn_s_memtime();
        // A path that stops moves on to the pixel's next sample, whose first hit is the cached primary hit.
        // Returns false when the lane has run the last sample of its sweep.
        // the primary hit of the lane's pixel (and, with_dir, the primary ray's direction)
        auto primary_hit = [&](bool with_dir) {
            if (TLAS_PARK) {
                const PathRec* r = &recs[slot_u[PT_SLOT_ENTRY * 64] & 0x7FFFFFFFu];
                const float4 ra = r->a, rb = r->b, rc = r->c;
                h.mat = __float_as_uint(ra.y);
                h.u = ra.z; h.v = ra.w;
                h.p = V3(rb.x, rb.y, rb.z);
                h.n = V3(rb.w, rc.x, rc.y);
                if (with_dir) rdir = V3(rc.z, rc.w, r->d.x);
            } else {
                if (with_dir) rdir = V3(slot[(PT_SLOT_DIR + 0) * 64], slot[(PT_SLOT_DIR + 1) * 64], slot[(PT_SLOT_DIR + 2) * 64]);
                h.mat = slot_u[(PT_SLOT_HIT + 0) * 64];
                h.p = V3(slot[(PT_SLOT_HIT + 1) * 64], slot[(PT_SLOT_HIT + 2) * 64], slot[(PT_SLOT_HIT + 3) * 64]);
                h.n = V3(slot[(PT_SLOT_HIT + 4) * 64], slot[(PT_SLOT_HIT + 5) * 64], slot[(PT_SLOT_HIT + 6) * 64]);
                h.u = slot[(PT_SLOT_HIT + 7) * 64];
                h.v = slot[(PT_SLOT_HIT + 8) * 64];
            }
        };
        auto next_sample = [&]() -> bool {
            const v3 sum = add3(V3(slot[(PT_SLOT_SUM + 0) * 64], slot[(PT_SLOT_SUM + 1) * 64], slot[(PT_SLOT_SUM + 2) * 64]), color);
            const uint32_t left = slot_u[PT_SLOT_LEFT * 64] - 1u;
            if (left == 0u) {
                const uint32_t e = slot_u[PT_SLOT_ENTRY * 64];
                if (e & 0x80000000u) {
                    const uint32_t pix = slot_u[PT_SLOT_PIX * 64];
                    const float ns = (float)pp.samples;
                    out[pix] = make_float4(pt_div(sum.x, ns), pt_div(sum.y, ns), pt_div(sum.z, ns), 1.0f);
                    // queries the reference would have run: it re-traces the primary ray for every sample
                    aux_rays[pix] = nrays + pp.samples - 1u;
                } else {
                    // park the pixel until its next sweep
                    u32x4 lo, hi;
                    lo.x = rng.x; lo.y = rng.y; lo.z = rng.z; lo.w = nrays;
                    hi.x = __float_as_uint(sum.x); hi.y = __float_as_uint(sum.y); hi.z = __float_as_uint(sum.z); hi.w = 0u;
                    carry_store(&carry[e], lo, hi);
                }
                st = PT_ST_IDLE;
                return false;
            }
            slot[(PT_SLOT_SUM + 0) * 64] = sum.x; slot[(PT_SLOT_SUM + 1) * 64] = sum.y; slot[(PT_SLOT_SUM + 2) * 64] = sum.z;
            slot_u[PT_SLOT_LEFT * 64] = left;
            bounce = 0;
            rpos = origin;
            ray_color = V3(1, 1, 1);
            color = V3(0, 0, 0);
            primary_hit(true);
            return true;
        };
        // what the surface emits, and whether the path stops on it (black albedo)
        auto add_emitted = [&]() -> bool {
            const DMaterial* m = &sc.materials[h.mat];
            v3 emissive = V3(m->emissive[0], m->emissive[1], m->emissive[2]);
            const uint32_t et = m->emissive_texture;
            if (et != PT_NONE) emissive = sample_texture(sc, et, h.u, h.v);
            color = V3(pt_fma(ray_color.x, emissive.x, color.x), pt_fma(ray_color.y, emissive.y, color.y),
                       pt_fma(ray_color.z, emissive.z, color.z));
            const v3 alb = V3(m->albedo[0], m->albedo[1], m->albedo[2]);
            return dot3(alb, alb) < PT_EPSILON;
        };
        // One pass per round. A lane whose path ends AFTER scattering (throughput below the cutoff that could not be
        // foreseen, or the bounce limit in the general case) has begun its next sample too late for this round's
        // scatter step: it waits one round (PT_ST_RESTART, 3 in 1000 rounds on the example scenes) rather than
        // making the whole wave loop here.
        bool pending = st == PT_ST_SHADE || st == PT_ST_RESTART;
        if (st == PT_ST_RESTART) {
            primary_hit(false);
            st = PT_ST_SHADE;
        }
        if (__any(pending)) {
            PT_COUNT(cnt, 0);
            // 1. (cheap) a path that ends on this hit, a miss or a black surface, restarts from the primary hit ...
            bool restarted = false;
            if (pending) {
                bool stop;
                PT_COUNT(cnt, 1);
                if (h.mat != PT_NONE) {
                    PT_COUNT(cnt, 2);
                    stop = add_emitted();
                    // Scatters whose only surviving effect is the one random draw of cosine_sample: the hit of a
                    // path's last segment (the reference still scatters off it, src/raytrace.wgsl:256-260, but the new
                    // ray is never traced and the throughput never used again), and a scatter that is known to leave
                    // a throughput of exactly zero (src/raytrace.wgsl:266-269 then ends the path). Only the RNG is
                    // stepped and the path ends here, in the cheap step, instead of costing the wave a scatter pass.
                    if (!stop && (bounce + 1u >= bounces || scatter_goes_dark(sc, h, rdir, ray_color))) {
                        pcg3d(rng);
                        stop = true;
                    }
                } else {
                    color = V3(pt_fma(ray_color.x, pp.sky[0], color.x), pt_fma(ray_color.y, pp.sky[1], color.y),
                               pt_fma(ray_color.z, pp.sky[2], color.z));
                    stop = true;
                }
                if (stop) {
                    PT_COUNT(cnt, 3);
                    pending = next_sample();
                    restarted = pending;
                }
            }
            // 2. ... whose surface never ends a path (k_primary finished those pixels)
            if (restarted) { PT_COUNT(cnt, 4); (void)add_emitted(); }
            // 3. (expensive) so every lane that still has work scatters in the same pass
            if (pending) {
                PT_COUNT(cnt, 5);
                const DMaterial m = sc.materials[h.mat];
                v3 N = h.n;
                if (m.normal_map_texture != PT_NONE) N = mul3(N, sample_texture(sc, m.normal_map_texture, h.u, h.v));
                v3 ndir, bcol;
                if (STATS) cnt.shades++;
                if (MODE == PT_MODE_LDS) {
                    // bounce 0 scatters off the primary hit: its frame is in LDS; waves whose lanes all restart a
                    // sample together (the common case) skip the frame computation altogether
                    BrdfFrame f;
                    if (bounce == 0u) {
                        const float* fs = slot + PT_SLOT_FRAME * 64;
                        f.tt = V3(fs[0 * 64], fs[1 * 64], fs[2 * 64]);
                        f.bb = V3(fs[3 * 64], fs[4 * 64], fs[5 * 64]);
                        f.refl_om = V3(fs[6 * 64], fs[7 * 64], fs[8 * 64]);
                        f.NdotV = fs[9 * 64];
                        f.albedo = V3(fs[10 * 64], fs[11 * 64], fs[12 * 64]);
                    } else {
                        f = brdf_frame(sc, rdir, N, h.u, h.v, m);
                    }
                    brdf_sample(rng, rdir, N, m.roughness, f, ndir, bcol);
                } else {
                    calculate_brdf(sc, rng, rdir, N, h.u, h.v, m, ndir, bcol);
                }
                rdir = ndir;
                rpos = madd3(h.p, rdir, 0.001f);
                ray_color = mul3(ray_color, bcol);
                const float p = pt_max(ray_color.x, pt_max(ray_color.y, ray_color.z));
                bounce++;
                if (p < PT_EPSILON || bounce >= bounces) {
                    // dark path or bounce limit: next sample; its primary hit is shaded in the next round
                    PT_COUNT(cnt, 6);
                    if (next_sample()) st = PT_ST_RESTART;
                    pending = false;
                } else {
                    st = PT_ST_TRACE;
                    pending = false;
                    if (MODE == PT_MODE_TLAS) {   // a new walk starts at the root
                        ws.node = (sc.tlas_root == PT_NONE) ? PT_WALK_DONE : sc.tlas_root;
                        ws.sp = 0;
                        wb.t = PT_T_MAX; wb.obj = PT_NONE; wb.tri = PT_NONE; wb.u = 0.0f; wb.v = 0.0f;
                        if (TLAS_PARK_WALK) park_walk();
                    }
                }
            }
        }

        // ---- C. trace ---------------------------------------------------------------------------------------
        if (STATS) {
            t2 = __builtin_amdgcn_s_memtime();
            n_rounds++;
            lane_rounds += (unsigned long long)__popcll(__ballot(st == PT_ST_TRACE));
        }
        // No lane holds a hit that still has to be shaded at this point: say so, or the nine registers of the old
        // record stay occupied all through the closest-hit code.
        h.mat = PT_NONE; h.p = V3(0, 0, 0); h.n = V3(0, 0, 0); h.u = 0.0f; h.v = 0.0f;
        if (MODE == PT_MODE_TLAS) {
            // Lanes walk the TLAS until pp.walk_quota of them are through; those get their hit record and go back to
            // shading in the next round while the others' walks continue where they stopped. With incoherent rays
            // the walks of a wave differ in length several times over; waiting for the longest one every round left
            // four lanes in five idle.
            if (__any(st == PT_ST_TRACE)) {
                const bool tracing = st == PT_ST_TRACE;
                if (TLAS_PARK_WALK) {
                    const float* pk = slot + PT_SLOT_PARK * 64;
                    const uint32_t* pku = slot_u + PT_SLOT_PARK * 64;
                    ws.node = pku[0 * 64]; ws.sp = (int)pku[1 * 64];
                    wb.t = pk[2 * 64]; wb.obj = pku[3 * 64]; wb.tri = pku[4 * 64]; wb.u = pk[5 * 64]; wb.v = pk[6 * 64];
                }
                if (TLAS_PARK) {
                    float* pk = slot + PT_SLOT_PARK * 64;
                    uint32_t* pku = slot_u + PT_SLOT_PARK * 64;
                    pk[0 * 64] = ray_color.x; pk[1 * 64] = ray_color.y; pk[2 * 64] = ray_color.z;
                    pk[3 * 64] = color.x; pk[4 * 64] = color.y; pk[5 * 64] = color.z;
                    pku[6 * 64] = rng.x; pku[7 * 64] = rng.y; pku[8 * 64] = rng.z;
                    pku[9 * 64] = bounce; pku[10 * 64] = nrays; pku[11 * 64] = lane_rays;
                }
                const WideRay wr = make_wide_ray(world_slab_ray(rpos, rdir));
                wide_walk_resume<STATS, PT_BURST_TLAS>([&](uint32_t r, uint4& q0, uint4& q1, uint4& q2, uint4& q3) { tlas_fetch(sc, r, q0, q1, q2, q3); }, wr, wb.t, stk, ws, pp.walk_quota, cnt,
                                                       [&](uint32_t obj, int sp) { object_test<STATS, false>(sc, obj, rpos, rdir, wb, stk, sp, cnt); });
                if (TLAS_PARK) {
                    const float* pk = slot + PT_SLOT_PARK * 64;
                    const uint32_t* pku = slot_u + PT_SLOT_PARK * 64;
                    ray_color = V3(pk[0 * 64], pk[1 * 64], pk[2 * 64]);
                    color = V3(pk[3 * 64], pk[4 * 64], pk[5 * 64]);
                    rng.x = pku[6 * 64]; rng.y = pku[7 * 64]; rng.z = pku[8 * 64];
                    bounce = pku[9 * 64]; nrays = pku[10 * 64]; lane_rays = pku[11 * 64];
                }
                if (tracing && ws.node == PT_WALK_DONE) {
                    nrays++;
                    lane_rays++;
                    if (STATS && wb.obj != PT_NONE) cnt.hits++;
                    h = finalize_hit<MODE != PT_MODE_LDS>(sc, rpos, rdir, wb);
                    st = PT_ST_SHADE;
                }
                if (TLAS_PARK_WALK) park_walk();
            }
        } else if (__any(st == PT_ST_TRACE)) {
            const bool tracing = st == PT_ST_TRACE;
            Best b;
            cnt.passes = 0;
            if (PT_QUAD_TRACE && MODE != PT_MODE_LDS && quad_wave) {
                // 16 rays in this wave, in the first lane of every quad: the quads trace them together
                const v3 qo = V3(quad_perm_f32<PT_QUAD_BCAST0>(rpos.x), quad_perm_f32<PT_QUAD_BCAST0>(rpos.y), quad_perm_f32<PT_QUAD_BCAST0>(rpos.z));
                const v3 qd = V3(quad_perm_f32<PT_QUAD_BCAST0>(rdir.x), quad_perm_f32<PT_QUAD_BCAST0>(rdir.y), quad_perm_f32<PT_QUAD_BCAST0>(rdir.z));
                const bool qa = quad_perm_u32<PT_QUAD_BCAST0>(tracing ? 1u : 0u) != 0u;
                b.t = PT_T_MAX; b.obj = PT_NONE; b.tri = PT_NONE; b.u = 0.0f; b.v = 0.0f;
                if (STATS) quad_rounds++;
                closest_hit_list_quad<STATS, MODE == PT_MODE_LIST_LB>(sc, qo, qd, qa, b, stk, cnt);
            } else
            closest_hit<MODE, STATS>(sc, rpos, rdir, tracing, b, stk, cnt);
            if (STATS) n_passes += cnt.passes;
            // waves whose rays are expensive (many candidate objects) are the critical path of the launch: let them issue first
            if (cnt.passes >= PT_WAVE_PRIO_PASSES) __builtin_amdgcn_s_setprio(3);
            else __builtin_amdgcn_s_setprio(0);
            PT_COUNT(cnt, 7);
#ifdef PT_EXEC_COUNTS
            unsigned long long tf = __builtin_amdgcn_s_memtime();
#endif
            if (tracing) {
                PT_COUNT(cnt, 12);
                nrays++;
                lane_rays++;
                if (STATS && b.obj != PT_NONE) cnt.hits++;
                h = finalize_hit<MODE != PT_MODE_LDS>(sc, rpos, rdir, b);
                st = PT_ST_SHADE;
            }
            PT_COUNT_CYCLES(cnt, 15, tf);   // list scenes: cycles of the hit records
        }
        if (STATS) {
            const unsigned long long t3 = __builtin_amdgcn_s_memtime();
            cyc_refill += t1 - t0;
            cyc_shade += t2 - t1;
            cyc_trace += t3 - t2;
        }
    }

    const uint32_t r = wave_sum_u32(lane_rays);
    if (lane == 0 && r) atomicAdd(&counters->rays, (unsigned long long)r);
    // (chunk_order and order_info come together; the former is in use all through the loop anyway)
    if (chunk_order && lane == 0 && atomicAdd(&counters->waves_done, 1u) + 1u == 4u * gridDim.x) atomicSub(&g_frames_running, 1u);
#ifdef PT_EXEC_COUNTS
    if (STATS) {
        for (int k = 0; k < 16; ++k) {
            const uint32_t v = wave_sum_u32(cnt.exec[k]);
            if (lane == 0) atomicAdd(&counters->life_hist[16 + k], v);   // debug build: the upper half of the histogram
        }
    }
#endif
    if (STATS) {
        const uint32_t a = wave_sum_u32(cnt.xforms), b = wave_sum_u32(cnt.box_tests), c = wave_sum_u32(cnt.tri_tests),
                       d = wave_sum_u32(cnt.shades), e = wave_sum_u32(cnt.hits);
        if (lane == 0) {
            const unsigned long long life = __builtin_amdgcn_s_memtime() - t_begin;
            atomicAdd(&counters->wave_life_sum, life);
            atomicMax(&counters->wave_life_max, life);
#ifndef PT_EXEC_COUNTS
            { unsigned long long bkt = life >> 19; if (bkt > 31) bkt = 31; atomicAdd(&counters->life_hist[bkt], 1u); }
#endif
            if (n_rounds) atomicAdd(&counters->waves, 1ull);
            if (wave_log) {
                // per-wave record: start and end on the 100 MHz clock, lifetime in 16-cycle units, rounds; lane rounds,
                // first list position, items taken << 8 | XCC_ID, HW_ID; the wave's longest item: 16-cycle units, rounds, index
                const uint32_t hw_id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
                const uint32_t xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
                uint4* rec = wave_log + 4u * (blockIdx.x * 4u + wave);
                rec[0] = make_uint4((uint32_t)t_begin_real, (uint32_t)__builtin_amdgcn_s_memrealtime(), (uint32_t)(life >> 4), (uint32_t)n_rounds);
                {
                    const unsigned long long now = __builtin_amdgcn_s_memtime();
                    if (n_chunks != 0 && now - item_t0 > worst_cycles) { worst_cycles = now - item_t0; worst_rounds = n_rounds - item_r0; worst_item = cur_item; }
                }
                rec[1] = make_uint4((uint32_t)lane_rounds, first_entry, (n_chunks << 8) | (xcc_id & 15u), hw_id);
                rec[2] = make_uint4((uint32_t)(worst_cycles >> 4), (uint32_t)worst_rounds, worst_item, 0u);
                rec[3] = make_uint4(quad_rounds, 0u, 0u, 0u);
            }
            atomicAdd(&counters->cyc_refill, cyc_refill);
            atomicAdd(&counters->cyc_trace, cyc_trace);
            atomicAdd(&counters->cyc_shade, cyc_shade);
            atomicAdd(&counters->rounds, n_rounds);
            atomicAdd(&counters->passes, n_passes);
            atomicAdd(&counters->lane_rounds, lane_rounds);
            atomicAdd(&counters->hits, (unsigned long long)e);
            atomicAdd(&counters->xforms, (unsigned long long)a);
            atomicAdd(&counters->box_tests, (unsigned long long)b);
            atomicAdd(&counters->tri_tests, (unsigned long long)c);
            atomicAdd(&counters->shades, (unsigned long long)d);
        }
    }
}

// ---- hit_all for a batch of rays (the query stage on its own) ---------------------------------------------------
struct DRay {
    float pos[3];
    float dir[3];
};
struct DHit {
    uint32_t object, triangle, flat_id;
    float t;
    float p[3], n[3], uv[2];
};

template <int MODE>
__global__ __launch_bounds__(256) void k_query(PT_SCENE_PARAMS, LdsScene ls, uint32_t stack_depth, const DRay* __restrict__ rays,
                                               uint32_t n_rays, DHit* __restrict__ hits) {
    PT_SCENE_VIEW_FROM_PARAMS()
    extern __shared__ __attribute__((aligned(16))) uint32_t s_stack[];  // [4 waves][stack_depth][64 lanes], then the staged scene (PT_MODE_LDS)
    if (MODE == PT_MODE_LDS) stage_scene_in_lds(sc, ls, s_stack + 4 * stack_depth * 64);
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    LdsStack stk;
    stk.base = &s_stack[wave * stack_depth * 64 + lane];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < n_rays;
    v3 ro = V3(0, 0, 0), rd = V3(0, 0, 1);
    if (valid) {
        ro = V3(rays[i].pos[0], rays[i].pos[1], rays[i].pos[2]);
        rd = V3(rays[i].dir[0], rays[i].dir[1], rays[i].dir[2]);
    }
    LaneCounters cnt = {};
    Best b;
    closest_hit<MODE, false>(sc, ro, rd, valid, b, stk, cnt);
    if (!valid) return;
    HitInfo h = finalize_hit<MODE != PT_MODE_LDS>(sc, ro, rd, b);
    DHit o;
    o.object = b.obj;
    o.t = b.t;
    if (b.obj != PT_NONE) {
        o.triangle = b.tri - (MODE != PT_MODE_LDS ? sc.objects[b.obj].tri_first : 0u);
        o.flat_id = sc.objects[b.obj].tri_base + o.triangle;
    } else {
        o.triangle = PT_NONE;
        o.flat_id = PT_NONE;
    }
    o.p[0] = h.p.x; o.p[1] = h.p.y; o.p[2] = h.p.z;
    o.n[0] = h.n.x; o.n[1] = h.n.y; o.n[2] = h.n.z;
    o.uv[0] = h.u; o.uv[1] = h.v;
    hits[i] = o;
}

// RGBA f32 -> RGBA binary16, round to nearest even (the reference's render target is Rgba16Float, src/shader.rs:321)
__global__ void k_pack_f16(const float4* __restrict__ img, uint2* __restrict__ out, uint32_t n_pixels) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    const float4 v = img[i];
    const uint32_t x = __half_as_ushort(__float2half_rn(v.x)), y = __half_as_ushort(__float2half_rn(v.y));
    const uint32_t z = __half_as_ushort(__float2half_rn(v.z)), w = __half_as_ushort(__float2half_rn(v.w));
    out[i] = make_uint2(x | (y << 16), z | (w << 16));
}

// acc += frame; out = acc / n_frames  (radiance accumulation across frames; RGB only, A stays 1)
__global__ void k_accumulate(float4* __restrict__ acc, float4* __restrict__ img, uint32_t n_pixels, float n_frames) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    float4 a = acc[i], f = img[i];
    a.x += f.x; a.y += f.y; a.z += f.z; a.w = 1.0f;
    acc[i] = a;
    img[i] = make_float4(pt_div(a.x, n_frames), pt_div(a.y, n_frames), pt_div(a.z, n_frames), 1.0f);
}

// pt_amd.hip — context, scene upload and the C ABI of include/pt_amd.h.
// Single translation unit for gfx950: hipcc --offload-arch=gfx950 -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/pt_amd.h"
#include "pt_build.h"
#include "pt_kernels.h"

static_assert(sizeof(DMaterial) == sizeof(PtMaterial) && sizeof(DTexture) == sizeof(PtTexture), "wire == device");
static_assert(sizeof(WObject) == sizeof(PtObject) && sizeof(WMesh) == sizeof(PtMesh) && sizeof(WVertex) == sizeof(PtVertex), "wire");
static_assert(sizeof(DRay) == sizeof(PtRay) && sizeof(DHit) == sizeof(PtHit), "query structs");

namespace {

constexpr uint32_t kBlasMinTris = 16;   // PT_ACCEL_AUTO: meshes with more triangles than this get a BLAS
constexpr uint32_t kTlasMinObjects = 12; // PT_ACCEL_AUTO: scenes with more objects than this get a TLAS
constexpr uint32_t kMaxStackDepth = 96;
constexpr uint32_t kBlasNeedMaxSah = 48;    // stack entries a surface-area BLAS may need before the mesh is built in Morton order instead
constexpr uint32_t kTlasSettledUploads = 8;        // uploads with one object count after which a scene counts as settled (see rebuild_objects)
constexpr uint32_t kSahHostMaxTris = 1u << 17;     // meshes up to this many triangles get their BLAS from the host-side builder
constexpr uint32_t kCollapseOneGroupMax = 4096;   // primitives up to which one workgroup makes the whole wide tree
constexpr uint32_t kTlasMaxRefits = 15;  // object uploads that may refit the TLAS before it is built afresh  // LDS: 4 waves * depth * 256 B per workgroup

thread_local std::string g_create_error;

// ---- work memory of pt_render_device_noalloc -----------------------------------------------------------------------
// Everything a frame needs besides the scene, its image and the context: path list, parked pixel state, the four aux
// planes, per-entry cost, per-chunk sweep flags and the work counters, for frames of at most PT_NOALLOC_MAX_PIXELS
// pixels (sizes rounded up to whole 8x8 tiles). One array per device, part of the code object, held until the process
// ends; host code only ever uses the address hipGetSymbolAddress returns for the current device (g_scratch_ptr).
constexpr size_t kNoallocPix = PT_NOALLOC_MAX_PIXELS;
constexpr size_t kScrRecs = 0;                                         // PathRec[pix]
constexpr size_t kScrCarry = kScrRecs + sizeof(PathRec) * kNoallocPix;    // PathCarry[pix]
constexpr size_t kScrAux = kScrCarry + sizeof(PathCarry) * kNoallocPix;   // 4 planes of 4 B
constexpr size_t kScrCost = kScrAux + 16 * kNoallocPix;                   // u32[pix]
constexpr size_t kScrSweeps = kScrCost + 4 * kNoallocPix;                 // u32[pix / 64]
constexpr size_t kScrCounters = kScrSweeps + 4 * (kNoallocPix / 64);      // DCounters
constexpr size_t kScratchBytes = kScrCounters + ((sizeof(DCounters) + 255) & ~(size_t)255);
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[kMaxDevices] = {nullptr};   // per device: the address of its copy of `scratch`

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = std::max<size_t>(bytes, 256);
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want; else p = nullptr;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

}  // namespace

// The no-allocation work array itself (layout: kScr* above). In host code this name is only a handle for
// hipGetSymbolAddress; kernels and memsets get the per-device address kept in g_scratch_ptr.
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

// What a frame's per-pixel costs depend on (besides the seeds): compared bytewise.
struct FrameKey {
    uint64_t scene_version;
    uint32_t width, height, samples, bounces, part_index, part_count;
    float wfv[16], vfc[16];
};

// What the host reads back about a frame once it has finished. Two of them per lane, so that the caller can enqueue frame
// n+1 while frame n is still running (the launch overhead then hides behind the previous frame's kernels); the
// device-side work buffers of the lane are shared by the two, which stream order makes safe.
struct FrameSlot {
    hipEvent_t ev0 = nullptr, evm = nullptr, ev1 = nullptr;
    DevBuf counters;
    DCounters* h_counters = nullptr;      // page-locked copies, written behind the frame on its own stream: reading a frame's
    DCounters* h_counters_k1 = nullptr;   // record never goes through the null stream, whose hardware queue a frame stream may share
    bool pending = false;   // enqueued, not read back yet
    bool detail = false;    // rendered with detailed statistics
    uint32_t obj_set = 0;   // the object set (instances + TLAS) the frame reads
    uint32_t accel_used = 0; // PtStats::accel_used of the frame
    uint64_t paths = 0, primary = 0, primary_traced = 0;
};

constexpr uint32_t kMaxFrameSlots = PT_MAX_FRAME_SLOTS;     // frames one context keeps in flight (pt_render_submit)
constexpr uint32_t kMaxLanes = 1u + kMaxFrameSlots;
constexpr uint32_t kMaxObjectSets = kMaxFrameSlots + 2u;     // versions of the instance data + TLAS kept at most

// Everything one frame in flight owns besides the scene: its stream, its work memory (path list, parked pixel state, aux
// planes, chunk order, image) and the schedule hints its previous frame left. Lane 0 serves the synchronous entry points
// (pt_render, pt_render_device), lane 1 + k the slot k of pt_render_submit.
struct Lane {
    bool ready = false;               // events, counters and stream exist
    hipStream_t stream = nullptr;     // lane 0: the context's stream; else a stream of its own
    FrameSlot slot[2];                // host-visible records of the two most recent frames
    uint64_t frame_seq = 0;           // frames enqueued so far; frame n uses slot[n & 1]
    uint32_t last_schedule = 0;       // how the last frame's k_paths was set up (pt_debug_slot_state, out[3])
    hipStream_t last_frame_stream = nullptr;
    DevBuf d_stack_overflow, d_out, d_out16, d_aux_obj, d_aux_flat, d_aux_t, d_aux_rays, d_paths, d_sweeps_done, d_carry, d_cost, d_chunk_bin,
        d_chunk_order, d_wave_log;
    uint32_t wave_log_records = 0;
    uint64_t last_n_paths = ~0ull;    // entries of the path list of the last frame read back
    bool hint_valid = false;          // the aux ray-count plane holds a complete frame described by hint_key
    FrameKey hint_key;
    uint32_t aux_w = 0, aux_h = 0;
    bool submitted = false;           // pt_render_submit without its pt_render_wait yet
    hipEvent_t ev_done = nullptr;     // slot lanes: behind the submitted frame and the copy of its image (pt_render_stream_wait)
    hipEvent_t ev_dep = nullptr;      // slot lanes: what the slot's next work waits for (pt_render_slot_wait_for)
    PtStats stats{};
    DevBuf* bufs[14] = {&d_stack_overflow, &d_out, &d_out16, &d_aux_obj, &d_aux_flat, &d_aux_t, &d_aux_rays, &d_paths, &d_sweeps_done, &d_carry,
                        &d_cost, &d_chunk_bin, &d_chunk_order, &d_wave_log};
    Lane() = default;
    Lane(const Lane&) = delete;
    Lane& operator=(const Lane&) = delete;
};

// One version of what pt_upload_objects derives from the object array: instance records, world boxes and the TLAS the rays
// walk. A frame reads the version that was current when it was enqueued; an upload writes another one, so that frames in
// flight are not disturbed (the reference rewrites its object buffer every frame, src/extract.rs:297-346).
struct ObjectSet {
    DevBuf w_objects, d_objects, d_obj_l2w, d_obj_lo, d_obj_hi, d_tri_base, d_tlas_nodes;
    uint32_t n_objects = 0;
    uint32_t tlas_depth = 0, tlas_root = PT_NONE;   // depth: levels of the 4-wide tree
    uint32_t tlas_need = 0;                         // traversal stack entries the TLAS walk can need
    uint32_t tlas_n_wide = 0;                       // nodes of the TLAS
    uint32_t obj_max_mesh = 0, obj_max_mat = 0;
    bool use_tlas = false;
    bool refitted = false;                          // made by a refit of the previous version's tree
    bool built = false;                             // derived from h_objects and the current meshes
    DevBuf* bufs[7] = {&w_objects, &d_objects, &d_obj_l2w, &d_obj_lo, &d_obj_hi, &d_tri_base, &d_tlas_nodes};
    ObjectSet() = default;
    ObjectSet(const ObjectSet&) = delete;
    ObjectSet& operator=(const ObjectSet&) = delete;
};

struct pt_ctx {
    PtConfig cfg{};
    int device = 0;
    hipStream_t stream = nullptr;   // uploads, builds, queries, and the frames of the synchronous entry points
    hipStream_t debug_stream = nullptr;   // pt_debug_slot_state: made at its first call, so that its copies wait for no other stream
    hipEvent_t evb0 = nullptr, evb1 = nullptr;
    Lane lanes[kMaxLanes];
    uint32_t cur_lane = 0;        // the lane pt_get_stats / pt_read_aux / pt_debug_wave_log describe: that of the last frame completed
    ObjectSet sets[kMaxObjectSets];
    uint32_t n_sets = 1;          // versions in use so far
    uint32_t cur_set = 0;         // the version new frames read
    // a frame of pt_render_device_noalloc leaves no FrameSlot; this event stands for the last one
    hipEvent_t ev_noalloc = nullptr;
    bool noalloc_pending = false;
    uint32_t noalloc_set = 0;
    // radiance accumulation is one buffer per context: frames that add to it do so in the order they were enqueued
    hipEvent_t ev_acc = nullptr;
    bool acc_pending = false;
    hipStream_t acc_stream = nullptr;
    std::string err;
    char info[256] = {0};

    // uploaded wire arrays (device copies)
    DevBuf w_meshes, w_indices, w_vertices;
    // traversal layout
    DevBuf d_meshes, d_tris, d_tri_ids, d_attrs, d_tri_lo, d_tri_hi, d_blas_nodes, d_blas_bnodes, d_blas_order;
    DevBuf d_tlas_bnodes;
    DevBuf d_materials, d_textures, d_texdata;
    // scratch for uploads / builds
    DevBuf t_tri_mesh, t_mesh_first, t_keys0, t_keys1, t_vals0, t_vals1, t_hist, t_par_int, t_par_leaf, t_arrive,
        t_bounds, t_info, t_errors, t_ranges, t_work, t_tris_tmp, t_cstate;
    // the topology of the TLAS a refit keeps (upload stream only), query buffers, accumulation
    DevBuf d_acc, d_rays, d_hits, d_tlas_order, d_tlas_par_int, d_tlas_par_leaf;

    std::vector<DMesh> h_meshes;
    std::vector<uint32_t> h_blas_nodes_base;  // per mesh: first node of its BLAS (or PT_NONE)
    std::vector<uint32_t> h_blas_n_wide;      // per mesh: nodes of its (4-wide) BLAS
    uint32_t n_meshes = 0, n_tris = 0, n_materials = 0, n_textures = 0, n_texdata = 0;
    uint32_t n_blas_nodes = 0;
    uint32_t max_blas_depth = 0;   // levels of the deepest 4-wide BLAS
    uint32_t max_blas_need = 0;    // traversal stack entries a BLAS walk can need
    uint32_t tlas_top_max = PT_NONE;             // nodes of its top levels k_paths stages in LDS at most (PT_NONE: what fits beside four workgroups per compute unit)
    uint32_t lds_stack_max = 16;                 // PT_MODE_TLAS, k_paths: traversal stack entries per lane kept in LDS
    bool have_meshes = false, have_materials = false, have_objects = false;
    std::vector<PtObject> h_objects;  // kept so that a mesh re-upload can rebuild the instance data

    unsigned char* scratch = nullptr;   // this device's no-allocation work array (pt_render_device_noalloc), or null
    uint32_t n_cus = 256;
    uint32_t chunk_pixels = 64;   // see PathParams::chunk
    uint32_t refill_min = 0;      // see PathParams::refill_min; 0 = choose per scene
    uint32_t max_sweeps = 3;      // see PathParams::n_sweeps
    uint32_t last_waves_needed = 0;   // the last frame that was ordered by cost hints: its work over its longest chunk, in waves (0: none yet)
    uint32_t deep_pipeline_slots = 4;   // frames of other slots in flight from which object-list scenes are served for throughput (PT_AMD_DEEP_PIPELINE; 0: always, 99: never)
    uint32_t alone_streak = 0;    // consecutive submits on slots that found no frame of another slot still on the device
    bool sweeps_when_shared = false;   // PT_AMD_SWEEPS_SHARED=1: sample sweeps also for frames that run beside other frames (for measurements)
    std::vector<void*> host_allocs;  // page-locked host buffers handed out by pt_host_alloc
    bool queue_ahead = true;      // let the caller enqueue a frame while the previous one is still running
    int list_regs = 0;             // object-list variants of k_paths: 0 chosen per frame, 1 always the build for four waves per SIMD, 2 always the roomy one
    uint32_t walk_quota = 16;     // see PathParams::walk_quota
    uint32_t blas_max_leaf = 2;   // triangles a BLAS leaf holds at most (1..PT_WIDE_MAX_LEAF); 2 measured best on the 968-triangle mesh of scene.glb
    bool carry_acquire = true;    // see PathParams::carry_acquire (costs about 1 % of a frame; PT_AMD_CARRY_ACQUIRE=0 leaves it out)
    bool split_chunks = true;     // see PathParams::split_chunks: the most expensive chunks of list-mode scenes in four parts (shorter single frames, and more frames per second since the workgroups of such scenes leave wave slots free)
    bool lds_blas = true;         // object-list scenes: stage the biggest BLAS that fits in LDS (k_paths)
    bool lds_blas_nodes_only = false;   // ... its nodes only, never the triangle records (PT_AMD_LDS_BLAS_NODES_ONLY)
    bool lds_scene = true;        // small scenes are staged in LDS whole (PT_AMD_NO_LDS_SCENE switches it off)
    bool k1_narrow = false;       // k_primary with four waves per workgroup even where sixteen fit (PT_AMD_K1_NARROW)
    int blocks_per_cu = 0;        // workgroups of k_paths per compute unit at most (PT_AMD_BLOCKS_PER_CU; 0: what fits)
    bool debug = false;           // PT_AMD_DEBUG: a line per frame on stderr
    bool slot_priority = false;   // see lane_init
    uint32_t lds_per_cu = 160 * 1024;
    bool allow_refit = true;      // pt_upload_objects may refit the TLAS instead of rebuilding it
    uint32_t tlas_built_n = 0;    // object count the kept TLAS topology was built for (0: none kept)
    bool tlas_upgrade_pending = false;   // ... made by the device builder because the count kept changing: due for one surface-area build
    uint32_t tlas_last_n = 0;         // object count of the previous upload
    uint32_t tlas_stable_uploads = 0; // uploads in a row with that count (the count before it lasted tlas_stable_before)
    uint32_t tlas_stable_before = 0;
    uint32_t tlas_refits = 0;     // refits since that build
    bool cost_hints = true;       // order the path list by the previous frame's per-pixel ray counts when they apply
    uint32_t split_fraction[2] = {PT_SPLIT_FRACTION_ALONE, PT_SPLIT_FRACTION};   // at most 1/this of the chunks are handed out in parts {frame alone on the device, other frames running}
    uint32_t split_load_8ths = 16;                 // chunks are handed out in parts only while work / longest chunk <= this/8 x the waves of the launch; 0: always
    uint32_t group_margin_8ths[2] = {24, 8};       // see ChunkOrderInfo::n_groups: waves = this/8 x work / longest chunk, {frame alone on the device, other frames running}; 0: always the whole grid
    bool blas_sah = true;         // BLAS of moderate meshes by the surface-area heuristic on the host (else Morton order)
    bool tlas_sah = true;         // the TLAS of a newly arrived object set likewise (PT_AMD_TLAS_BUILDER=lbvh: always Morton order)
    bool strict_hints = false;    // see render_enqueue
    uint64_t scene_version = 0;   // bumped by every upload
    uint32_t acc_w = 0, acc_h = 0, acc_frames = 0;
    bool stats_detail = false;
    double build_ms = 0.0;        // device time of the last TLAS build or refit
};

namespace {

int fail(pt_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}

#define PT_HIP(ctx, call)                                                                                   \
    do {                                                                                                    \
        hipError_t e__ = (call);                                                                            \
        if (e__ != hipSuccess) {                                                                            \
            int code__ = (e__ == hipErrorOutOfMemory) ? PT_ERR_OUT_OF_MEMORY : PT_ERR_DEVICE;               \
            return fail(ctx, code__, std::string(#call) + ": " + hipGetErrorString(e__));                   \
        }                                                                                                   \
    } while (0)

inline uint32_t div_up(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
inline bool finite3(const float* v) { return std::isfinite(v[0]) && std::isfinite(v[1]) && std::isfinite(v[2]); }

// ---- LBVH over n boxes box_lo/hi[first .. first+n) -> nodes[0 .. n-1) ----------------------
// Child references to internal nodes are node_base + index; depth goes to info_slot (device).
// leaf_is_pos: leaves are referenced by sorted position (BLAS) instead of by primitive number (TLAS); then the ranges of
// sorted positions below every node are kept in t_ranges for the collapse.
int build_lbvh(pt_ctx* ctx, const float4* box_lo, const float4* box_hi, uint32_t first, uint32_t n, BvhNode* nodes,
               uint32_t node_base, uint32_t* info_slot, bool leaf_is_pos) {
    hipStream_t s = ctx->stream;
    const uint32_t n_blocks = div_up(n, RS_KEYS);
    PT_HIP(ctx, ctx->t_keys0.ensure(sizeof(uint32_t) * n));
    PT_HIP(ctx, ctx->t_keys1.ensure(sizeof(uint32_t) * n));
    PT_HIP(ctx, ctx->t_vals0.ensure(sizeof(uint32_t) * n));
    PT_HIP(ctx, ctx->t_vals1.ensure(sizeof(uint32_t) * n));
    PT_HIP(ctx, ctx->t_hist.ensure(sizeof(uint32_t) * 256 * n_blocks));
    PT_HIP(ctx, ctx->t_par_int.ensure(sizeof(uint32_t) * n));
    PT_HIP(ctx, ctx->t_par_leaf.ensure(sizeof(uint32_t) * n));
    PT_HIP(ctx, ctx->t_arrive.ensure(sizeof(uint32_t) * n));
    PT_HIP(ctx, ctx->t_bounds.ensure(sizeof(uint32_t) * 8));
    if (leaf_is_pos) PT_HIP(ctx, ctx->t_ranges.ensure(sizeof(uint2) * n));

    uint32_t* bounds = ctx->t_bounds.as<uint32_t>();
    PT_HIP(ctx, hipMemsetAsync(bounds, 0xFF, 12, s));
    PT_HIP(ctx, hipMemsetAsync(bounds + 3, 0x00, 12, s));
    k_centroid_bounds<<<div_up(n, 256), 256, 0, s>>>(box_lo, box_hi, first, n, bounds);
    uint32_t *k0 = ctx->t_keys0.as<uint32_t>(), *k1 = ctx->t_keys1.as<uint32_t>();
    uint32_t *v0 = ctx->t_vals0.as<uint32_t>(), *v1 = ctx->t_vals1.as<uint32_t>();
    k_morton<<<div_up(n, 256), 256, 0, s>>>(box_lo, box_hi, first, n, bounds, k0, v0);
    uint32_t* hist = ctx->t_hist.as<uint32_t>();
    for (uint32_t shift = 0; shift < 32; shift += 8) {
        k_rs_hist<8><<<n_blocks, 64, 0, s>>>(k0, n, nullptr, shift, n_blocks, hist);
        k_rs_scan<<<1, 1024, 0, s>>>(hist, 256 * n_blocks);
        k_rs_scatter<8><<<n_blocks, 64, 0, s>>>(k0, v0, k1, v1, n, nullptr, shift, n_blocks, hist);
        std::swap(k0, k1);
        std::swap(v0, v1);
    }
    // after 4 passes the sorted data is back in t_keys0 / t_vals0
    uint32_t* par_int = ctx->t_par_int.as<uint32_t>();
    uint32_t* par_leaf = ctx->t_par_leaf.as<uint32_t>();
    uint32_t* arrive = ctx->t_arrive.as<uint32_t>();
    PT_HIP(ctx, hipMemsetAsync(arrive, 0, sizeof(uint32_t) * n, s));
    k_karras<<<div_up(n - 1, 256), 256, 0, s>>>(k0, n, par_int, par_leaf, leaf_is_pos ? ctx->t_ranges.as<uint2>() : nullptr);
    k_fit<<<div_up(n, 256), 256, 0, s>>>(box_lo, box_hi, first, v0, n, par_int, par_leaf, arrive, nodes, node_base, info_slot, leaf_is_pos ? 1u : 0u);
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}

// ---- binary BVH over n boxes by the surface-area heuristic, made on the host ---------------------------------------------
// For the BLAS of meshes of moderate size: a mesh is uploaded once and walked by every ray of every frame, so the better
// tree is worth the milliseconds (16 bins per axis; where no plane separates the centroids the range is cut in the
// middle). Produces exactly what build_lbvh() leaves behind for a BLAS: the nodes (root = node 0, leaves referenced by
// sorted position), the sorted order in t_vals0, the range of positions below every node in t_ranges, the depth in
// info_slot[0]. The boxes are unions of the very boxes the device computed for the triangles.
// With leaf_is_pos == false (a TLAS: leaves name the primitive itself) it also leaves the parent links a refit climbs
// (t_par_int, t_par_leaf, in k_karras's form) and the per-node ranges are not kept.
int build_sah_host(pt_ctx* ctx, const float4* box_lo, const float4* box_hi, uint32_t first, uint32_t n, BvhNode* nodes,
                   uint32_t node_base, uint32_t* info_slot, bool leaf_is_pos = true) {
    hipStream_t s = ctx->stream;
    std::vector<float4> lo(n), hi(n);
    PT_HIP(ctx, hipMemcpyAsync(lo.data(), box_lo + first, sizeof(float4) * n, hipMemcpyDeviceToHost, s));
    PT_HIP(ctx, hipMemcpyAsync(hi.data(), box_hi + first, sizeof(float4) * n, hipMemcpyDeviceToHost, s));
    PT_HIP(ctx, hipStreamSynchronize(s));
    struct Box { float lo[3], hi[3]; };
    auto empty = [] { Box b; for (int a = 0; a < 3; ++a) { b.lo[a] = 3.0e38f; b.hi[a] = -3.0e38f; } return b; };
    auto grow = [&](Box& b, uint32_t p) {
        const float l[3] = {lo[p].x, lo[p].y, lo[p].z}, h[3] = {hi[p].x, hi[p].y, hi[p].z};
        for (int a = 0; a < 3; ++a) { b.lo[a] = std::min(b.lo[a], l[a]); b.hi[a] = std::max(b.hi[a], h[a]); }
    };
    auto merge = [](Box& b, const Box& o) { for (int a = 0; a < 3; ++a) { b.lo[a] = std::min(b.lo[a], o.lo[a]); b.hi[a] = std::max(b.hi[a], o.hi[a]); } };
    auto area = [](const Box& b) {
        const double dx = std::max(0.0f, b.hi[0] - b.lo[0]), dy = std::max(0.0f, b.hi[1] - b.lo[1]), dz = std::max(0.0f, b.hi[2] - b.lo[2]);
        return dx * dy + dy * dz + dz * dx;
    };
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    std::vector<float> cen(3 * (size_t)n);
    for (uint32_t i = 0; i < n; ++i) {
        cen[3 * (size_t)i] = 0.5f * lo[i].x + 0.5f * hi[i].x;
        cen[3 * (size_t)i + 1] = 0.5f * lo[i].y + 0.5f * hi[i].y;
        cen[3 * (size_t)i + 2] = 0.5f * lo[i].z + 0.5f * hi[i].z;
    }
    std::vector<BvhNode> out(n - 1);
    std::vector<uint2> ranges(n - 1);
    std::vector<uint32_t> par_int, par_leaf;
    if (!leaf_is_pos) { par_int.assign(n, 0u); par_leaf.assign(n, 0u); }
    struct Task { uint32_t node, lo, hi; };
    std::vector<Task> stack;
    stack.push_back({0u, 0u, n - 1u});
    uint32_t next_node = 1;
    constexpr int kBins = 16;
    while (!stack.empty()) {
        const Task t = stack.back();
        stack.pop_back();
        ranges[t.node] = make_uint2(t.lo, t.hi);
        float cmin[3] = {3.0e38f, 3.0e38f, 3.0e38f}, cmax[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
        for (uint32_t k = t.lo; k <= t.hi; ++k)
            for (int a = 0; a < 3; ++a) { const float c = cen[3 * (size_t)order[k] + a]; cmin[a] = std::min(cmin[a], c); cmax[a] = std::max(cmax[a], c); }
        int best_axis = -1, best_plane = 0;
        double best_cost = 1.0e300;
        float best_scale = 0.0f;
        for (int a = 0; a < 3; ++a) {
            const float ext = cmax[a] - cmin[a];
            if (!(ext > 0.0f)) continue;
            const float scale = (float)kBins / ext;
            Box bb[kBins];
            uint32_t cnt[kBins] = {};
            for (int b = 0; b < kBins; ++b) bb[b] = empty();
            for (uint32_t k = t.lo; k <= t.hi; ++k) {
                const uint32_t p = order[k];
                const int b = std::min(kBins - 1, std::max(0, (int)((cen[3 * (size_t)p + a] - cmin[a]) * scale)));
                cnt[b]++;
                grow(bb[b], p);
            }
            double right_area[kBins];
            uint32_t right_cnt[kBins];
            Box acc = empty();
            uint32_t c = 0;
            for (int b = kBins - 1; b >= 1; --b) { merge(acc, bb[b]); c += cnt[b]; right_area[b] = area(acc); right_cnt[b] = c; }
            acc = empty();
            c = 0;
            for (int b = 0; b < kBins - 1; ++b) {   // plane between bin b and b + 1
                merge(acc, bb[b]);
                c += cnt[b];
                if (c == 0 || right_cnt[b + 1] == 0) continue;
                const double cost = area(acc) * c + right_area[b + 1] * right_cnt[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_plane = b; best_scale = scale; }
            }
        }
        uint32_t mid;   // last position of the left part
        if (best_axis >= 0) {
            const int a = best_axis;
            uint32_t i = t.lo, j = t.hi;
            auto left = [&](uint32_t p) { return std::min(kBins - 1, std::max(0, (int)((cen[3 * (size_t)p + a] - cmin[a]) * best_scale))) <= best_plane; };
            while (true) {
                while (i <= j && left(order[i])) ++i;
                while (i < j && !left(order[j])) --j;
                if (i >= j) break;
                std::swap(order[i], order[j]);
                ++i; --j;
            }
            mid = i - 1u;   // (both parts are non-empty: the plane was chosen with counts > 0 on both sides)
        } else {
            mid = t.lo + (t.hi - t.lo) / 2u;
        }
        BvhNode& N = out[t.node];
        for (int side = 0; side < 2; ++side) {
            const uint32_t a = side ? mid + 1u : t.lo, b = side ? t.hi : mid;
            Box bx = empty();
            for (uint32_t k = a; k <= b; ++k) grow(bx, order[k]);
            uint32_t ref;
            const uint32_t link = t.node | (side ? 0x80000000u : 0u);   // parent link: bit 31 = "I am the right child"
            if (a == b) {
                ref = PT_LEAF_BIT | (leaf_is_pos ? a : order[a]);
                if (!leaf_is_pos) par_leaf[a] = link;
            } else {
                const uint32_t child = next_node++;
                ref = node_base + child;
                if (!leaf_is_pos) par_int[child] = link;
                stack.push_back({child, a, b});
            }
            uint32_t bits = ref;
            float w;
            std::memcpy(&w, &bits, sizeof w);
            if (side == 0) { N.lo_l = make_float4(bx.lo[0], bx.lo[1], bx.lo[2], w); N.hi_l = make_float4(bx.hi[0], bx.hi[1], bx.hi[2], 0.0f); }
            else { N.lo_r = make_float4(bx.lo[0], bx.lo[1], bx.lo[2], w); N.hi_r = make_float4(bx.hi[0], bx.hi[1], bx.hi[2], 0.0f); }
        }
    }
    // subtree depths, as k_fit records them (a node's children have larger numbers than the node)
    std::vector<uint32_t> depth(n - 1, 0);
    auto ref_of = [](float w) { uint32_t b; std::memcpy(&b, &w, sizeof b); return b; };
    auto as_float = [](uint32_t b) { float w; std::memcpy(&w, &b, sizeof w); return w; };
    for (uint32_t i = n - 1; i-- > 0;) {
        const uint32_t rl = ref_of(out[i].lo_l.w), rr = ref_of(out[i].lo_r.w);
        const uint32_t dl = (rl & PT_LEAF_BIT) ? 0u : depth[rl - node_base], dr = (rr & PT_LEAF_BIT) ? 0u : depth[rr - node_base];
        out[i].hi_l.w = as_float(dl);
        out[i].hi_r.w = as_float(dr);
        depth[i] = std::max(dl, dr) + 1u;
    }
    PT_HIP(ctx, ctx->t_vals0.ensure(sizeof(uint32_t) * n));
    PT_HIP(ctx, ctx->t_ranges.ensure(sizeof(uint2) * n));
    PT_HIP(ctx, hipMemcpyAsync(ctx->t_vals0.p, order.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, s));
    PT_HIP(ctx, hipMemcpyAsync(ctx->t_ranges.p, ranges.data(), sizeof(uint2) * (n - 1), hipMemcpyHostToDevice, s));
    if (!leaf_is_pos) {
        PT_HIP(ctx, ctx->t_par_int.ensure(sizeof(uint32_t) * n));
        PT_HIP(ctx, ctx->t_par_leaf.ensure(sizeof(uint32_t) * n));
        PT_HIP(ctx, hipMemcpyAsync(ctx->t_par_int.p, par_int.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, s));
        PT_HIP(ctx, hipMemcpyAsync(ctx->t_par_leaf.p, par_leaf.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, s));
    }
    PT_HIP(ctx, hipMemcpyAsync(nodes, out.data(), sizeof(BvhNode) * (n - 1), hipMemcpyHostToDevice, s));
    PT_HIP(ctx, hipMemcpyAsync(info_slot, &depth[0], sizeof(uint32_t), hipMemcpyHostToDevice, s));
    PT_HIP(ctx, hipStreamSynchronize(s));   // (the host vectors go out of scope)
    return PT_OK;
}

// The 4-wide tree the rays walk, from a binary tree of n - 1 nodes over n primitives (k_collapse). info (device):
// {nodes, levels, root reference, stack entries a walk can need}.
// binary_depth (device): the word k_fit left the binary tree's depth in; a wide tree has at most that many levels.
int collapse_wide(pt_ctx* ctx, const BvhNode* bnodes, uint32_t bnode_base, uint32_t n, bool blas, WNode* wnodes, uint32_t wnode_base,
                  uint32_t* info, const uint32_t* binary_depth) {
    hipStream_t s = ctx->stream;
    PT_HIP(ctx, ctx->t_work.ensure(sizeof(uint2) * (size_t)std::max<uint32_t>(n, 1)));
    const uint2* ranges = blas ? ctx->t_ranges.as<uint2>() : nullptr;
    if (n <= kCollapseOneGroupMax) {
        k_collapse<<<1, 1024, 0, s>>>(bnodes, bnode_base, n, ranges, ctx->blas_max_leaf, wnodes, wnode_base, ctx->t_work.as<uint2>(), info);
        PT_HIP(ctx, hipGetLastError());
        return PT_OK;
    }
    // one launch per level; the number of levels is known only on the device, so as many launches as the binary tree is deep
    uint32_t depth = 0;
    PT_HIP(ctx, hipMemcpyAsync(&depth, binary_depth, sizeof depth, hipMemcpyDeviceToHost, s));
    PT_HIP(ctx, hipStreamSynchronize(s));
    const uint32_t launches = depth + 1u;
    const size_t state_bytes = sizeof(uint32_t) * (4 + 2 * ((size_t)launches + 2));
    PT_HIP(ctx, ctx->t_cstate.ensure(state_bytes));
    PT_HIP(ctx, hipMemsetAsync(ctx->t_cstate.p, 0, state_bytes, s));
    k_collapse_begin<<<1, 1, 0, s>>>(ctx->t_work.as<uint2>(), ctx->t_cstate.as<uint32_t>());
    const uint32_t groups = std::min<uint32_t>(2u * ctx->n_cus, div_up(n, 256));
    for (uint32_t level = 0; level < launches; ++level)
        k_collapse_level<<<groups, 256, 0, s>>>(bnodes, bnode_base, ranges, ctx->blas_max_leaf, wnodes, wnode_base, ctx->t_work.as<uint2>(),
                                                ctx->t_cstate.as<uint32_t>(), level, info);
    PT_HIP(ctx, hipGetLastError());
    return PT_OK;
}

int rebuild_objects(pt_ctx* ctx, uint32_t target);
int acquire_object_set(pt_ctx* ctx, uint32_t* target);
int finish_all(pt_ctx* ctx);
hipError_t lane_init(pt_ctx* ctx, Lane& L, bool own_stream);

}  // namespace

// ============================================================================================
extern "C" {

const char* pt_last_error(const pt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int pt_create(const PtConfig* cfg, pt_ctx** out_ctx) {
    try {
        if (!cfg || !out_ctx) return fail(nullptr, PT_ERR_INVALID_ARGUMENT, "pt_create: null argument");
        *out_ctx = nullptr;
        if (cfg->abi_version != PT_ABI_VERSION) return fail(nullptr, PT_ERR_INVALID_ARGUMENT, "pt_create: abi_version mismatch");
        if (cfg->accel > PT_ACCEL_BVH || cfg->tie_rule > PT_TIE_LOWEST) return fail(nullptr, PT_ERR_INVALID_ARGUMENT, "pt_create: bad accel/tie_rule");
        int n_dev = 0;
        if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(nullptr, PT_ERR_NO_DEVICE, "pt_create: no HIP device");
        if (cfg->device < 0 || cfg->device >= n_dev) return fail(nullptr, PT_ERR_NO_DEVICE, "pt_create: device ordinal out of range");
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return fail(nullptr, PT_ERR_DEVICE, "pt_create: hipGetDeviceProperties failed");
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            return fail(nullptr, PT_ERR_NO_DEVICE, std::string("pt_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
        pt_ctx* ctx = new pt_ctx();
        ctx->cfg = *cfg;
        ctx->device = cfg->device;
        ctx->n_cus = (uint32_t)std::max(prop.multiProcessorCount, 1);
        if (const char* e = std::getenv("PT_AMD_REFILL_MIN")) { int v = std::atoi(e); if (v >= 1 && v <= 64) ctx->refill_min = (uint32_t)v; }
        if (std::getenv("PT_AMD_NO_COST_HINTS")) ctx->cost_hints = false;
        if (std::getenv("PT_AMD_NO_REFIT")) ctx->allow_refit = false;
        if (const char* e = std::getenv("PT_AMD_SPLIT_CHUNKS")) ctx->split_chunks = std::atoi(e) != 0;
        if (const char* e = std::getenv("PT_AMD_GROUP_MARGIN")) {   // "alone,shared" in eighths, or one number for both
            int a = 0, b = 0;
            const int n = std::sscanf(e, "%d,%d", &a, &b);
            if (n == 1) b = a;
            if (n >= 1 && a >= 0 && a <= 4096 && b >= 0 && b <= 4096) { ctx->group_margin_8ths[0] = (uint32_t)a; ctx->group_margin_8ths[1] = (uint32_t)b; }
        }
        if (const char* e = std::getenv("PT_AMD_LIST_REGS")) ctx->list_regs = !std::strcmp(e, "roomy") ? 2 : !std::strcmp(e, "tight") ? 1 : 0;
        if (const char* e = std::getenv("PT_AMD_SPLIT_LOAD")) { int v = std::atoi(e); if (v >= 0 && v <= 65536) ctx->split_load_8ths = (uint32_t)v; }
        if (const char* e = std::getenv("PT_AMD_SPLIT_FRACTION")) {   // "alone,shared", or one number for both
            int a = 0, b = 0;
            const int n = std::sscanf(e, "%d,%d", &a, &b);
            if (n == 1) b = a;
            if (n >= 1 && a >= 1 && a <= 4096 && b >= 1 && b <= 4096) { ctx->split_fraction[0] = (uint32_t)a; ctx->split_fraction[1] = (uint32_t)b; }
        }
        if (const char* e = std::getenv("PT_AMD_CARRY_ACQUIRE")) ctx->carry_acquire = std::atoi(e) != 0;
        if (const char* e = std::getenv("PT_AMD_BLAS_BUILDER")) ctx->blas_sah = std::strcmp(e, "lbvh") != 0;
        if (const char* e = std::getenv("PT_AMD_TLAS_BUILDER")) ctx->tlas_sah = std::strcmp(e, "lbvh") != 0;
        if (const char* e = std::getenv("PT_AMD_STRICT_HINTS")) ctx->strict_hints = std::atoi(e) != 0;
        if (const char* e = std::getenv("PT_AMD_WALK_QUOTA")) { int v = std::atoi(e); if (v >= 1 && v <= 64) ctx->walk_quota = (uint32_t)v; }
        if (const char* e = std::getenv("PT_AMD_BLAS_LEAF")) { int v = std::atoi(e); if (v >= 1 && v <= (int)PT_WIDE_MAX_LEAF) ctx->blas_max_leaf = (uint32_t)v; }
        if (const char* e = std::getenv("PT_AMD_LDS_STACK")) { int v = std::atoi(e); if (v >= 4 && v <= (int)kMaxStackDepth) ctx->lds_stack_max = (uint32_t)v; }
        if (const char* e = std::getenv("PT_AMD_TLAS_TOP")) { int v = std::atoi(e); if (v >= 0 && v <= 2048) ctx->tlas_top_max = (uint32_t)v; }
        if (std::getenv("PT_AMD_NO_LDS_BLAS")) ctx->lds_blas = false;
        if (std::getenv("PT_AMD_LDS_BLAS_NODES_ONLY")) ctx->lds_blas_nodes_only = true;
        if (std::getenv("PT_AMD_NO_LDS_SCENE")) ctx->lds_scene = false;
        if (std::getenv("PT_AMD_K1_NARROW")) ctx->k1_narrow = true;
        if (const char* e = std::getenv("PT_AMD_BLOCKS_PER_CU")) { int v = std::atoi(e); if (v >= 1 && v <= 64) ctx->blocks_per_cu = v; }
        if (std::getenv("PT_AMD_DEBUG")) ctx->debug = true;
        if (const char* e = std::getenv("PT_AMD_SLOT_PRIORITY")) ctx->slot_priority = std::atoi(e) != 0;
        if (prop.maxSharedMemoryPerMultiProcessor > 0) ctx->lds_per_cu = (uint32_t)prop.maxSharedMemoryPerMultiProcessor;
        if (std::getenv("PT_AMD_NO_QUEUE_AHEAD")) ctx->queue_ahead = false;
        if (const char* e = std::getenv("PT_AMD_DEEP_PIPELINE")) { int v = std::atoi(e); if (v >= 0 && v <= 99) ctx->deep_pipeline_slots = (uint32_t)v; }
        if (const char* e = std::getenv("PT_AMD_SWEEPS_SHARED")) ctx->sweeps_when_shared = std::atoi(e) != 0;
        if (const char* e = std::getenv("PT_AMD_SWEEPS")) { int v = std::atoi(e); if (v >= 1 && v <= PT_MAX_SWEEPS) ctx->max_sweeps = (uint32_t)v; }
        if (const char* e = std::getenv("PT_AMD_CHUNK")) { int v = std::atoi(e); if (v >= 64 && v <= 65536) ctx->chunk_pixels = (uint32_t)(v / 64) * 64; else if (v == 8 || v == 16 || v == 32) ctx->chunk_pixels = (uint32_t)v; }
        std::snprintf(ctx->info, sizeof ctx->info, "pt_amd abi %d | %s %s | %d CUs | %d MHz | %.0f GiB", PT_ABI_VERSION, prop.name,
                      prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000, prop.totalGlobalMem / 1073741824.0);
        hipError_t e = hipSetDevice(ctx->device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = lane_init(ctx, ctx->lanes[0], false);
        if (e == hipSuccess) e = hipEventCreate(&ctx->evb0);
        if (e == hipSuccess) e = hipEventCreate(&ctx->evb1);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_noalloc, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_acc, hipEventDisableTiming);
        if (e == hipSuccess && ctx->device < kMaxDevices) {
            // this device's copy of the no-allocation work array (the device is current here)
            std::lock_guard<std::mutex> lock(g_scratch_mutex);
            if (!g_scratch_ptr[ctx->device]) {
                void* ptr = nullptr;
                e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
                if (e == hipSuccess) g_scratch_ptr[ctx->device] = static_cast<unsigned char*>(ptr);
            }
            ctx->scratch = g_scratch_ptr[ctx->device];
        }
        if (e != hipSuccess) {
            std::string m = std::string("pt_create: ") + hipGetErrorString(e);
            pt_destroy(ctx);
            return fail(nullptr, PT_ERR_DEVICE, m);
        }
        *out_ctx = ctx;
        return PT_OK;
    } catch (const std::bad_alloc&) {
        return fail(nullptr, PT_ERR_OUT_OF_MEMORY, "pt_create: host allocation failed");
    } catch (...) {
        return fail(nullptr, PT_ERR_DEVICE, "pt_create: unexpected exception");
    }
}

int pt_destroy(pt_ctx* ctx) {
    if (!ctx) return PT_OK;
    (void)hipSetDevice(ctx->device);
    // frames may still be running on a caller's stream or on the lanes' own: wait for them before any buffer they use is released
    for (Lane& L : ctx->lanes) {
        for (FrameSlot& sl : L.slot)
            if (sl.ev1 && sl.pending) (void)hipEventSynchronize(sl.ev1);
        if (L.stream) (void)hipStreamSynchronize(L.stream);
    }
    if (ctx->ev_noalloc && ctx->noalloc_pending) (void)hipEventSynchronize(ctx->ev_noalloc);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    DevBuf* bufs[] = {&ctx->w_meshes, &ctx->w_indices, &ctx->w_vertices, &ctx->d_meshes, &ctx->d_tris,
                      &ctx->d_attrs, &ctx->d_tri_lo, &ctx->d_tri_hi, &ctx->d_blas_nodes, &ctx->d_materials, &ctx->d_textures,
                      &ctx->d_texdata, &ctx->t_tri_mesh, &ctx->t_mesh_first, &ctx->t_keys0, &ctx->t_keys1, &ctx->t_vals0,
                      &ctx->t_vals1, &ctx->t_hist, &ctx->t_par_int, &ctx->t_par_leaf, &ctx->t_arrive, &ctx->t_bounds,
                      &ctx->t_info, &ctx->t_errors, &ctx->d_acc, &ctx->d_rays, &ctx->d_hits, &ctx->d_tlas_order, &ctx->d_tlas_par_int,
                      &ctx->d_tlas_par_leaf, &ctx->d_tri_ids, &ctx->d_blas_bnodes, &ctx->d_blas_order, &ctx->d_tlas_bnodes, &ctx->t_ranges,
                      &ctx->t_work, &ctx->t_tris_tmp, &ctx->t_cstate};
    for (DevBuf* b : bufs) b->release();
    for (ObjectSet& os : ctx->sets)
        for (DevBuf* b : os.bufs) b->release();
    for (void* p : ctx->host_allocs) (void)hipHostFree(p);
    for (Lane& L : ctx->lanes) {
        for (DevBuf* b : L.bufs) b->release();
        for (FrameSlot& sl : L.slot) {
            sl.counters.release();
            if (sl.h_counters) (void)hipHostFree(sl.h_counters);
            if (sl.h_counters_k1) (void)hipHostFree(sl.h_counters_k1);
            if (sl.ev0) (void)hipEventDestroy(sl.ev0);
            if (sl.ev1) (void)hipEventDestroy(sl.ev1);
            if (sl.evm) (void)hipEventDestroy(sl.evm);
        }
        if (L.ev_done) (void)hipEventDestroy(L.ev_done);
        if (L.ev_dep) (void)hipEventDestroy(L.ev_dep);
        if (L.stream && L.stream != ctx->stream) (void)hipStreamDestroy(L.stream);
    }
    if (ctx->evb0) (void)hipEventDestroy(ctx->evb0);
    if (ctx->evb1) (void)hipEventDestroy(ctx->evb1);
    if (ctx->ev_noalloc) (void)hipEventDestroy(ctx->ev_noalloc);
    if (ctx->ev_acc) (void)hipEventDestroy(ctx->ev_acc);
    if (ctx->debug_stream) (void)hipStreamDestroy(ctx->debug_stream);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return PT_OK;
}

int pt_device_info(pt_ctx* ctx, char* buf, size_t buf_bytes) {
    if (!ctx || !buf || buf_bytes == 0) return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_device_info: null argument");
    std::snprintf(buf, buf_bytes, "%s", ctx->info);
    return PT_OK;
}

// ---- meshes ---------------------------------------------------------------------------------
int pt_upload_meshes(pt_ctx* ctx, const PtMesh* meshes, uint32_t n_meshes, const uint32_t* indices, uint32_t n_indices,
                     const PtVertex* vertices, uint32_t n_vertices) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    try {
        { int rc_wait = finish_all(ctx); if (rc_wait != PT_OK) return rc_wait; }   // no frame may still be reading the scene
        if ((n_meshes && !meshes) || (n_indices && !indices) || (n_vertices && !vertices))
            return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_upload_meshes: null array with non-zero count");
        PT_HIP(ctx, hipSetDevice(ctx->device));
        // validate (the shader would read out of bounds; here it is an error)
        uint64_t total_tris = 0;
        for (uint32_t m = 0; m < n_meshes; ++m) {
            const PtMesh& me = meshes[m];
            if ((uint64_t)me.ihead + 3ull * me.tri_count > n_indices)
                return fail(ctx, PT_ERR_INVALID_SCENE, "pt_upload_meshes: mesh " + std::to_string(m) + " index range exceeds indices[]");
            for (uint32_t i = 0; i < 3u * me.tri_count; ++i) {
                if ((uint64_t)me.vhead + indices[me.ihead + i] >= n_vertices)
                    return fail(ctx, PT_ERR_INVALID_SCENE, "pt_upload_meshes: mesh " + std::to_string(m) + " references a vertex past vertices[]");
                // The mesh box is used to cull by distance (the reference only gates on it), which is only equivalent
                // when it encloses every triangle, as the boxes src/extract.rs:77-93 computes do.
                const float* pos = vertices[me.vhead + indices[me.ihead + i]].position;
                for (int k = 0; k < 3; ++k)
                    if (!(pos[k] >= me.aabb_min[k] && pos[k] <= me.aabb_max[k]))
                        return fail(ctx, PT_ERR_INVALID_SCENE, "pt_upload_meshes: mesh " + std::to_string(m) + " has a vertex outside its aabb_min/aabb_max");
            }
            total_tris += me.tri_count;
        }
        if (total_tris >= 0x7FFFFFFFull) return fail(ctx, PT_ERR_UNSUPPORTED, "pt_upload_meshes: too many triangles");
        for (uint32_t v = 0; v < n_vertices; ++v)
            if (!finite3(vertices[v].position) || !finite3(vertices[v].normal) || !std::isfinite(vertices[v].uv[0]) || !std::isfinite(vertices[v].uv[1]))
                return fail(ctx, PT_ERR_INVALID_SCENE, "pt_upload_meshes: vertex " + std::to_string(v) + " is not finite");

        hipStream_t s = ctx->stream;
        PT_HIP(ctx, hipStreamSynchronize(s));
        const uint32_t n_tris = (uint32_t)total_tris;
        // host-side mesh table
        std::vector<DMesh> hm(n_meshes);
        std::vector<uint32_t> tri_mesh(n_tris), mesh_first(n_meshes);
        uint32_t acc = 0, n_nodes = 0;
        std::vector<uint32_t> node_base(n_meshes, PT_NONE);
        for (uint32_t m = 0; m < n_meshes; ++m) {
            const PtMesh& me = meshes[m];
            DMesh& d = hm[m];
            float ma = 0.0f;
            for (int k = 0; k < 3; ++k) {
                d.bmin[k] = me.aabb_min[k];
                d.bmax[k] = me.aabb_max[k];
                ma = std::max(ma, std::max(std::fabs(me.aabb_min[k]), std::fabs(me.aabb_max[k])));
            }
            d.tri_first = acc;
            d.tri_count = me.tri_count;
            d.blas_root = PT_NONE;
            d.blas_need = 0;
            d.max_abs = std::isfinite(ma) ? ma : 0.0f;
            d._pad = 0;
            mesh_first[m] = acc;
            for (uint32_t i = 0; i < me.tri_count; ++i) tri_mesh[acc + i] = m;
            bool want_blas = (ctx->cfg.accel == PT_ACCEL_BVH && me.tri_count >= 2) ||
                             (ctx->cfg.accel == PT_ACCEL_AUTO && me.tri_count > kBlasMinTris);
            // (a BLAS leaf names its first triangle by its position in the mesh in PT_WLEAF_COUNT_SHIFT bits)
            if (want_blas && me.tri_count > PT_WLEAF_POS_MASK + 1u)
                return fail(ctx, PT_ERR_UNSUPPORTED, "pt_upload_meshes: mesh " + std::to_string(m) + " has more than 2^29 triangles");
            if (want_blas) {
                node_base[m] = n_nodes;
                n_nodes += me.tri_count - 1;
            }
            acc += me.tri_count;
        }
        PT_HIP(ctx, ctx->w_meshes.ensure(sizeof(PtMesh) * (size_t)n_meshes));
        PT_HIP(ctx, ctx->w_indices.ensure(sizeof(uint32_t) * (size_t)n_indices));
        PT_HIP(ctx, ctx->w_vertices.ensure(sizeof(PtVertex) * (size_t)n_vertices));
        PT_HIP(ctx, ctx->d_meshes.ensure(sizeof(DMesh) * (size_t)n_meshes));
        PT_HIP(ctx, ctx->d_tris.ensure(sizeof(DTri) * (size_t)n_tris));
        PT_HIP(ctx, ctx->d_attrs.ensure(sizeof(DTriAttr) * (size_t)n_tris));
        PT_HIP(ctx, ctx->d_tri_lo.ensure(sizeof(float4) * (size_t)n_tris));
        PT_HIP(ctx, ctx->d_tri_hi.ensure(sizeof(float4) * (size_t)n_tris));
        PT_HIP(ctx, ctx->t_tri_mesh.ensure(sizeof(uint32_t) * (size_t)n_tris));
        PT_HIP(ctx, ctx->t_mesh_first.ensure(sizeof(uint32_t) * (size_t)n_meshes));
        PT_HIP(ctx, ctx->d_blas_bnodes.ensure(sizeof(BvhNode) * (size_t)n_nodes));   // binary trees: what the build makes first
        PT_HIP(ctx, ctx->d_blas_nodes.ensure(sizeof(WNode) * (size_t)n_nodes));       // the 4-wide trees the rays walk (never more nodes)
        PT_HIP(ctx, ctx->d_blas_order.ensure(sizeof(uint32_t) * (size_t)n_tris));
        PT_HIP(ctx, ctx->d_tri_ids.ensure(sizeof(uint32_t) * (size_t)n_tris));
        PT_HIP(ctx, ctx->t_info.ensure(sizeof(uint32_t) * 8 * (size_t)(n_meshes + 1)));   // per mesh (and one for the TLAS): binary depth, -, -, -, collapse info[4]
        if (n_meshes) PT_HIP(ctx, hipMemcpyAsync(ctx->w_meshes.p, meshes, sizeof(PtMesh) * (size_t)n_meshes, hipMemcpyHostToDevice, s));
        if (n_indices) PT_HIP(ctx, hipMemcpyAsync(ctx->w_indices.p, indices, sizeof(uint32_t) * (size_t)n_indices, hipMemcpyHostToDevice, s));
        if (n_vertices) PT_HIP(ctx, hipMemcpyAsync(ctx->w_vertices.p, vertices, sizeof(PtVertex) * (size_t)n_vertices, hipMemcpyHostToDevice, s));
        if (n_tris) PT_HIP(ctx, hipMemcpyAsync(ctx->t_tri_mesh.p, tri_mesh.data(), sizeof(uint32_t) * (size_t)n_tris, hipMemcpyHostToDevice, s));
        if (n_meshes) PT_HIP(ctx, hipMemcpyAsync(ctx->t_mesh_first.p, mesh_first.data(), sizeof(uint32_t) * (size_t)n_meshes, hipMemcpyHostToDevice, s));
        if (n_tris) {
            k_build_tris<<<div_up(n_tris, 256), 256, 0, s>>>(ctx->w_meshes.as<WMesh>(), ctx->t_tri_mesh.as<uint32_t>(),
                                                             ctx->t_mesh_first.as<uint32_t>(), ctx->w_indices.as<uint32_t>(),
                                                             ctx->w_vertices.as<WVertex>(), n_tris, ctx->d_tris.as<DTri>(),
                                                             ctx->d_attrs.as<DTriAttr>(), ctx->d_tri_lo.as<float4>(),
                                                             ctx->d_tri_hi.as<float4>());
            k_iota<<<div_up(n_tris, 256), 256, 0, s>>>(ctx->d_tri_ids.as<uint32_t>(), n_tris);   // (meshes without a BLAS keep their own order)
            PT_HIP(ctx, hipGetLastError());
        }
        PT_HIP(ctx, hipMemsetAsync(ctx->t_info.p, 0, sizeof(uint32_t) * 8 * (size_t)(n_meshes + 1), s));
        // One mesh's BLAS: binary tree (surface-area builder on the host for moderate meshes, Morton order on the device
        // otherwise), triangle records in leaf order, 4-wide tree.
        auto build_blas = [&](uint32_t m, bool sah) -> int {
            const uint32_t first = hm[m].tri_first, cnt = hm[m].tri_count;
            uint32_t* info = ctx->t_info.as<uint32_t>() + 8 * (size_t)m;
            BvhNode* bn = ctx->d_blas_bnodes.as<BvhNode>() + node_base[m];
            int rc = sah ? build_sah_host(ctx, ctx->d_tri_lo.as<float4>(), ctx->d_tri_hi.as<float4>(), first, cnt, bn, node_base[m], info)
                         : build_lbvh(ctx, ctx->d_tri_lo.as<float4>(), ctx->d_tri_hi.as<float4>(), first, cnt, bn, node_base[m], info, true);
            if (rc != PT_OK) return rc;
            // the mesh's triangle records in the order of the tree's leaves, and which triangle each one is (from the copy
            // in the mesh's own order made below)
            k_permute_tris<<<div_up(cnt, 256), 256, 0, s>>>(ctx->t_tris_tmp.as<DTri>() + first, ctx->t_vals0.as<uint32_t>(), cnt, ctx->d_tris.as<DTri>() + first,
                                                            ctx->d_tri_ids.as<uint32_t>() + first);
            PT_HIP(ctx, hipGetLastError());
            PT_HIP(ctx, hipMemcpyAsync(ctx->d_blas_order.as<uint32_t>() + first, ctx->t_vals0.p, sizeof(uint32_t) * (size_t)cnt, hipMemcpyDeviceToDevice, s));
            return collapse_wide(ctx, bn, node_base[m], cnt, true, ctx->d_blas_nodes.as<WNode>() + node_base[m], node_base[m], info + 4, info);
        };
        if (n_nodes) {   // (a copy of the records in mesh order: a BLAS that is built a second time permutes from it again)
            PT_HIP(ctx, ctx->t_tris_tmp.ensure(sizeof(DTri) * (size_t)n_tris));
            PT_HIP(ctx, hipMemcpyAsync(ctx->t_tris_tmp.p, ctx->d_tris.p, sizeof(DTri) * (size_t)n_tris, hipMemcpyDeviceToDevice, s));
        }
        std::vector<char> used_sah(n_meshes, 0);
        for (uint32_t m = 0; m < n_meshes; ++m) {
            if (node_base[m] == PT_NONE) continue;
            // moderate meshes get the better tree from the host, very large ones the fast one from the device
            used_sah[m] = ctx->blas_sah && hm[m].tri_count <= kSahHostMaxTris;
            const int rc = build_blas(m, used_sah[m] != 0);
            if (rc != PT_OK) return rc;
        }
        std::vector<uint32_t> infos(8 * (size_t)(n_meshes + 1), 0);
        PT_HIP(ctx, hipMemcpyAsync(infos.data(), ctx->t_info.p, sizeof(uint32_t) * infos.size(), hipMemcpyDeviceToHost, s));
        PT_HIP(ctx, hipStreamSynchronize(s));
        // The surface-area builder bounds neither the depth of its tree nor, with it, the stack a walk needs: triangles
        // spaced geometrically make it peel one off per level. The Morton-order tree is at most as deep as its keys
        // have bits, so a mesh whose tree would leave too little of the traversal stack is built again that way.
        bool rebuilt = false;
        for (uint32_t m = 0; m < n_meshes; ++m) {
            if (node_base[m] == PT_NONE || !used_sah[m] || infos[8 * (size_t)m + 4 + 3] <= kBlasNeedMaxSah) continue;
            PT_HIP(ctx, hipMemsetAsync(ctx->t_info.as<uint32_t>() + 8 * (size_t)m, 0, 8 * sizeof(uint32_t), s));
            if (ctx->debug) std::fprintf(stderr, "[pt_amd] mesh %u: its surface-area BLAS needs %u stack entries, built again in Morton order\n", m, infos[8 * (size_t)m + 4 + 3]);
            const int rc = build_blas(m, false);
            if (rc != PT_OK) return rc;
            rebuilt = true;
        }
        if (rebuilt) {
            PT_HIP(ctx, hipMemcpyAsync(infos.data(), ctx->t_info.p, sizeof(uint32_t) * infos.size(), hipMemcpyDeviceToHost, s));
            PT_HIP(ctx, hipStreamSynchronize(s));
        }
        uint32_t max_depth = 0, max_need = 0;
        std::vector<uint32_t> n_wide(n_meshes, 0);
        for (uint32_t m = 0; m < n_meshes; ++m) {
            if (node_base[m] == PT_NONE) continue;
            const uint32_t* ci = &infos[8 * (size_t)m + 4];
            n_wide[m] = ci[0];
            hm[m].blas_root = ci[2];
            hm[m].blas_need = ci[3];
            max_depth = std::max(max_depth, ci[1]);
            max_need = std::max(max_need, ci[3]);
        }
        if (n_meshes) PT_HIP(ctx, hipMemcpyAsync(ctx->d_meshes.p, hm.data(), sizeof(DMesh) * (size_t)n_meshes, hipMemcpyHostToDevice, s));
        PT_HIP(ctx, hipStreamSynchronize(s));
        ctx->h_blas_n_wide.swap(n_wide);
        ctx->max_blas_need = max_need;
        ctx->h_meshes.swap(hm);
        ctx->h_blas_nodes_base.swap(node_base);
        ctx->n_meshes = n_meshes;
        ctx->n_tris = n_tris;
        ctx->n_blas_nodes = n_nodes;
        ctx->max_blas_depth = max_depth;
        ctx->have_meshes = true;
        ctx->scene_version++;
        ctx->acc_frames = 0;
        // instance data depends on the mesh table (world boxes, flat triangle ids); no frame is in flight, so the current
        // version is rebuilt where it is, without the tree of the old meshes
        ctx->tlas_built_n = 0;
        if (ctx->have_objects) return rebuild_objects(ctx, ctx->cur_set);
        return PT_OK;
    } catch (const std::bad_alloc&) {
        return fail(ctx, PT_ERR_OUT_OF_MEMORY, "pt_upload_meshes: host allocation failed");
    } catch (...) {
        return fail(ctx, PT_ERR_DEVICE, "pt_upload_meshes: unexpected exception");
    }
}

// ---- materials --------------------------------------------------------------------------------
int pt_upload_materials(pt_ctx* ctx, const PtMaterial* materials, uint32_t n_materials, const PtTexture* textures,
                        uint32_t n_textures, const float* texture_data, uint32_t n_texture_data) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    try {
        { int rc_wait = finish_all(ctx); if (rc_wait != PT_OK) return rc_wait; }   // no frame may still be reading the scene
        if ((n_materials && !materials) || (n_textures && !textures) || (n_texture_data && !texture_data))
            return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_upload_materials: null array with non-zero count");
        for (uint32_t m = 0; m < n_materials; ++m) {
            const uint32_t t[4] = {materials[m].albedo_texture, materials[m].emissive_texture,
                                   materials[m].metallic_roughness_texture, materials[m].normal_map_texture};
            for (uint32_t k : t)
                if (k != PT_U32_MAX && k >= n_textures)
                    return fail(ctx, PT_ERR_INVALID_SCENE, "pt_upload_materials: material " + std::to_string(m) + " references a texture past textures[]");
        }
        PT_HIP(ctx, hipSetDevice(ctx->device));
        hipStream_t s = ctx->stream;
        PT_HIP(ctx, hipStreamSynchronize(s));
        PT_HIP(ctx, ctx->d_materials.ensure(sizeof(PtMaterial) * (size_t)n_materials));
        PT_HIP(ctx, ctx->d_textures.ensure(sizeof(PtTexture) * (size_t)n_textures));
        PT_HIP(ctx, ctx->d_texdata.ensure(sizeof(float) * (size_t)n_texture_data));
        if (n_materials) PT_HIP(ctx, hipMemcpyAsync(ctx->d_materials.p, materials, sizeof(PtMaterial) * (size_t)n_materials, hipMemcpyHostToDevice, s));
        if (n_textures) PT_HIP(ctx, hipMemcpyAsync(ctx->d_textures.p, textures, sizeof(PtTexture) * (size_t)n_textures, hipMemcpyHostToDevice, s));
        if (n_texture_data) PT_HIP(ctx, hipMemcpyAsync(ctx->d_texdata.p, texture_data, sizeof(float) * (size_t)n_texture_data, hipMemcpyHostToDevice, s));
        PT_HIP(ctx, hipStreamSynchronize(s));
        ctx->n_materials = n_materials;
        ctx->n_textures = n_textures;
        ctx->n_texdata = n_texture_data;
        ctx->have_materials = true;
        ctx->scene_version++;
        ctx->acc_frames = 0;
        return PT_OK;
    } catch (...) {
        return fail(ctx, PT_ERR_DEVICE, "pt_upload_materials: unexpected exception");
    }
}

// ---- objects + TLAS -----------------------------------------------------------------------------
int pt_upload_objects(pt_ctx* ctx, const PtObject* objects, uint32_t n_objects) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    try {
        if (n_objects && !objects) return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_upload_objects: null array with non-zero count");
        if (n_objects >= 0x7FFFFFFFu) return fail(ctx, PT_ERR_UNSUPPORTED, "pt_upload_objects: too many objects");
        for (uint32_t o = 0; o < n_objects; ++o) {
            for (int k = 0; k < 16; ++k)
                if (!std::isfinite(objects[o].local_to_world[k]) || !std::isfinite(objects[o].world_to_local[k]))
                    return fail(ctx, PT_ERR_INVALID_SCENE, "pt_upload_objects: object " + std::to_string(o) + " has a non-finite matrix");
        }
        // The reference rewrites the object buffer every frame (src/extract.rs:297-346), mostly with the same bytes:
        // then the instance data, the TLAS, the cost hints and the frames in flight all stay as they are.
        if (ctx->have_objects && ctx->sets[ctx->cur_set].built && ctx->h_objects.size() == n_objects &&
            (n_objects == 0 || std::memcmp(ctx->h_objects.data(), objects, sizeof(PtObject) * (size_t)n_objects) == 0))
            return PT_OK;
        PT_HIP(ctx, hipSetDevice(ctx->device));
        // The new version goes into a set of buffers that no frame in flight reads (frames keep the version they were
        // enqueued with), so an upload per frame does not drain the frames in flight.
        uint32_t target = ctx->cur_set;
        { int rc_set = acquire_object_set(ctx, &target); if (rc_set != PT_OK) return rc_set; }
        ctx->h_objects.assign(objects, objects + n_objects);
        ctx->have_objects = true;
        ctx->scene_version++;
        ctx->acc_frames = 0;
        if (!ctx->have_meshes) {  // instance data is derived once the mesh table exists
            ctx->sets[target].built = false;
            ctx->sets[target].n_objects = n_objects;
            ctx->cur_set = target;
            return PT_OK;
        }
        const int rc_build = rebuild_objects(ctx, target);
        // (a build that failed half-way leaves no usable version: the array on record is the new one, the current version
        // still the old one's, so it must not pass for built; frames in flight keep what they were enqueued with)
        if (rc_build != PT_OK) ctx->sets[ctx->cur_set].built = false;
        return rc_build;
    } catch (const std::bad_alloc&) {
        return fail(ctx, PT_ERR_OUT_OF_MEMORY, "pt_upload_objects: host allocation failed");
    } catch (...) {
        return fail(ctx, PT_ERR_DEVICE, "pt_upload_objects: unexpected exception");
    }
}

}  // extern "C"

namespace {

// Events, counters and (for the lanes of pt_render_submit) a stream of its own.
hipError_t lane_init(pt_ctx* ctx, Lane& L, bool own_stream) {
    if (L.ready) return hipSuccess;
    hipError_t e = hipSuccess;
    if (own_stream) {
        // PT_AMD_SLOT_PRIORITY=1: a priority of its own kind. The runtime keeps the hardware queues of each priority apart, so
        // the slots' streams then do not have to share the few queues a process has by default (four) with its other
        // streams (the null stream, the context's upload stream, the host application's own); where they do, the slots
        // mostly run their frames one after the other (scene.glb with four slots: 21.5 ms per frame against 10.7).
        // Not the default, because work of the two priorities side by side has stalled for good on this platform, waves
        // standing still in the middle of their work: frames on more than four such streams while their waves still
        // waited for each other inside k_paths (one process start in thirty), and copies on a caller's stream of the
        // default priority that were ordered against two such slots (one run in ten of a test file). Raising
        // GPU_MAX_HW_QUEUES in the environment gives the same overlap without priorities.
        int lo = 0, hi = 0;
        if (ctx->slot_priority && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi < lo)
            e = hipStreamCreateWithPriority(&L.stream, hipStreamNonBlocking, hi);
        else
            e = hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking);
    }
    else L.stream = ctx->stream;
    if (own_stream) {
        if (e == hipSuccess && !L.ev_done) e = hipEventCreateWithFlags(&L.ev_done, hipEventDisableTiming);
        if (e == hipSuccess && !L.ev_dep) e = hipEventCreateWithFlags(&L.ev_dep, hipEventDisableTiming);
    }
    for (FrameSlot& sl : L.slot) {
        if (e == hipSuccess && !sl.ev0) e = hipEventCreate(&sl.ev0);
        if (e == hipSuccess && !sl.ev1) e = hipEventCreate(&sl.ev1);
        if (e == hipSuccess && !sl.evm) e = hipEventCreate(&sl.evm);
        if (e == hipSuccess) e = sl.counters.ensure(sizeof(DCounters));
        if (e == hipSuccess && !sl.h_counters) e = hipHostMalloc(reinterpret_cast<void**>(&sl.h_counters), sizeof(DCounters), hipHostMallocDefault);
        if (e == hipSuccess && !sl.h_counters_k1) e = hipHostMalloc(reinterpret_cast<void**>(&sl.h_counters_k1), sizeof(DCounters), hipHostMallocDefault);
    }
    L.ready = e == hipSuccess;
    return e;
}

// Does a frame that has been enqueued and may still be running read object set k?
bool object_set_busy(pt_ctx* ctx, uint32_t k) {
    for (Lane& L : ctx->lanes)
        for (FrameSlot& sl : L.slot)
            if (sl.pending && sl.obj_set == k && hipEventQuery(sl.ev1) != hipSuccess) return true;
    if (ctx->noalloc_pending && ctx->noalloc_set == k && hipEventQuery(ctx->ev_noalloc) != hipSuccess) return true;
    return false;
}

int finish_slot(pt_ctx* ctx, Lane& L, FrameSlot& sl);

// The set of buffers the next version of the instance data goes into: the current one when no frame in flight reads it
// (callers that render one frame at a time keep a single version), else one that is free, else a new one, else -- every
// version in use -- the one after the current, once its frames have finished.
int acquire_object_set(pt_ctx* ctx, uint32_t* target) {
    if (!object_set_busy(ctx, ctx->cur_set)) { *target = ctx->cur_set; return PT_OK; }
    for (uint32_t k = 0; k < ctx->n_sets; ++k)
        if (k != ctx->cur_set && !object_set_busy(ctx, k)) { *target = k; return PT_OK; }
    if (ctx->n_sets < kMaxObjectSets) { *target = ctx->n_sets++; return PT_OK; }
    const uint32_t k = (ctx->cur_set + 1u) % ctx->n_sets;
    for (Lane& L : ctx->lanes)
        for (FrameSlot& sl : L.slot)
            if (sl.pending && sl.obj_set == k) { int rc = finish_slot(ctx, L, sl); if (rc != PT_OK) return rc; }
    if (ctx->noalloc_pending && ctx->noalloc_set == k) { PT_HIP(ctx, hipEventSynchronize(ctx->ev_noalloc)); ctx->noalloc_pending = false; }
    *target = k;
    return PT_OK;
}

// Derives version `target` of the instance data (records, world boxes, TLAS) from h_objects and the current meshes, on the
// context's stream, and makes it the current one. Frames in flight keep reading the versions they were enqueued with; the
// caller has made sure that none of them reads `target`.
int rebuild_objects(pt_ctx* ctx, uint32_t target) {
    const uint32_t n = (uint32_t)ctx->h_objects.size();
    PT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    PT_HIP(ctx, hipStreamSynchronize(s));
    ObjectSet& T = ctx->sets[target];
    const ObjectSet& P = ctx->sets[ctx->cur_set];   // the version before (may be the same buffers)
    std::vector<uint32_t> tri_base(n);
    uint64_t acc = 0;
    uint32_t max_mesh = 0, max_mat = 0;
    for (uint32_t o = 0; o < n; ++o) {
        const PtObject& ob = ctx->h_objects[o];
        if (ob.mesh >= ctx->n_meshes) {
            ctx->have_objects = false;
            return fail(ctx, PT_ERR_INVALID_SCENE, "pt_upload_objects: object " + std::to_string(o) + " references a mesh past meshes[]");
        }
        tri_base[o] = (uint32_t)acc;
        acc += ctx->h_meshes[ob.mesh].tri_count;
        max_mesh = std::max(max_mesh, ob.mesh);
        max_mat = std::max(max_mat, ob.mat);
    }
    if (acc >= 0xFFFFFFFFull) {
        ctx->have_objects = false;
        return fail(ctx, PT_ERR_UNSUPPORTED, "pt_upload_objects: more than 2^32-1 instanced triangles");
    }
    // of the tree a refit keeps
    const bool have_kept = P.built && P.use_tlas && P.tlas_n_wide > 0;
    const uint32_t kept[4] = {have_kept ? P.tlas_n_wide : 0u, P.tlas_depth, P.tlas_root, P.tlas_need};
    const WNode* kept_nodes = P.d_tlas_nodes.as<WNode>();
    T.obj_max_mesh = max_mesh;
    T.obj_max_mat = max_mat;
    T.n_objects = n;
    T.tlas_root = PT_NONE;
    T.tlas_depth = 0;
    T.tlas_need = 0;
    T.tlas_n_wide = 0;
    T.use_tlas = false;
    T.refitted = false;
    T.built = false;
    ctx->build_ms = 0.0;
    if (n == 0) { T.built = true; ctx->cur_set = target; return PT_OK; }

    PT_HIP(ctx, T.w_objects.ensure(sizeof(PtObject) * (size_t)n));
    PT_HIP(ctx, T.d_objects.ensure(sizeof(DObject) * (size_t)n));
    PT_HIP(ctx, T.d_obj_l2w.ensure(sizeof(Affine) * (size_t)n));
    PT_HIP(ctx, T.d_obj_lo.ensure(sizeof(float4) * (size_t)n));
    PT_HIP(ctx, T.d_obj_hi.ensure(sizeof(float4) * (size_t)n));
    PT_HIP(ctx, T.d_tri_base.ensure(sizeof(uint32_t) * (size_t)n));
    PT_HIP(ctx, ctx->t_info.ensure(sizeof(uint32_t) * 8 * (size_t)(ctx->n_meshes + 1)));
    PT_HIP(ctx, hipMemcpyAsync(T.w_objects.p, ctx->h_objects.data(), sizeof(PtObject) * (size_t)n, hipMemcpyHostToDevice, s));
    PT_HIP(ctx, hipMemcpyAsync(T.d_tri_base.p, tri_base.data(), sizeof(uint32_t) * (size_t)n, hipMemcpyHostToDevice, s));
    PT_HIP(ctx, hipEventRecord(ctx->evb0, s));
    k_prep_objects<<<div_up(n, 256), 256, 0, s>>>(T.w_objects.as<WObject>(), T.d_tri_base.as<uint32_t>(),
                                                  ctx->d_meshes.as<DMesh>(), n, T.d_objects.as<DObject>(), T.d_obj_l2w.as<Affine>(),
                                                  T.d_obj_lo.as<float4>(), T.d_obj_hi.as<float4>());
    PT_HIP(ctx, hipGetLastError());
    const bool want_tlas = (ctx->cfg.accel == PT_ACCEL_BVH) || (ctx->cfg.accel == PT_ACCEL_AUTO && n > kTlasMinObjects);
    uint32_t cinfo[4] = {0, 0, PT_NONE, 0};   // collapse info: nodes, levels, root, stack need
    bool refitted = false;
    if (!want_tlas || n < 2) ctx->tlas_built_n = 0;
    if (want_tlas) {
        if (n == 1) {
            cinfo[2] = PT_LEAF_BIT | 0u;
        } else {
            uint32_t* slot = ctx->t_info.as<uint32_t>() + 8 * (size_t)ctx->n_meshes;
            PT_HIP(ctx, hipMemsetAsync(slot, 0, 8 * sizeof(uint32_t), s));
            const size_t nb = sizeof(uint32_t) * (size_t)n;
            PT_HIP(ctx, T.d_tlas_nodes.ensure(sizeof(WNode) * (size_t)(n - 1)));
            // Same number of objects as the tree was built for: keep the binary tree's topology (leaf order and parent
            // links) and only fit the boxes again, bottom-up. Any valid tree gives the same image; a stale topology only
            // costs traversal steps, so a fresh build is made every kTlasMaxRefits updates. The 4-wide tree keeps its
            // topology too: its nodes (copied from the version before when this one has buffers of its own) take the new
            // boxes from the binary nodes they were made from.
            // The surface-area builder runs on the host and takes tens of milliseconds for tens of thousands of objects: worth
            // it for a scene that arrives and then stays (first upload, or a new count after the old one had lasted), not for
            // one whose count changes every few frames, which gets the device builder; once such a scene has kept its count
            // for kTlasSettledUploads uploads, its tree is built once more, by surface area.
            if (n == ctx->tlas_last_n) ctx->tlas_stable_uploads++;
            else { ctx->tlas_stable_before = ctx->tlas_stable_uploads; ctx->tlas_stable_uploads = 0; ctx->tlas_last_n = n; }
            const bool sah_possible = ctx->tlas_sah && n <= kSahHostMaxTris;
            const bool upgrade = sah_possible && ctx->tlas_built_n == n && ctx->tlas_upgrade_pending && ctx->tlas_stable_uploads >= kTlasSettledUploads;
            if (ctx->allow_refit && ctx->tlas_built_n == n && ctx->tlas_refits < kTlasMaxRefits && kept[0] > 0 && !upgrade) {
                PT_HIP(ctx, ctx->t_arrive.ensure(nb));
                PT_HIP(ctx, hipMemsetAsync(ctx->t_arrive.p, 0, nb, s));
                k_fit<<<div_up(n, 256), 256, 0, s>>>(T.d_obj_lo.as<float4>(), T.d_obj_hi.as<float4>(), 0, ctx->d_tlas_order.as<uint32_t>(), n,
                                                     ctx->d_tlas_par_int.as<uint32_t>(), ctx->d_tlas_par_leaf.as<uint32_t>(),
                                                     ctx->t_arrive.as<uint32_t>(), ctx->d_tlas_bnodes.as<BvhNode>(), 0, slot, 0u);
                PT_HIP(ctx, hipGetLastError());
                if (kept_nodes != T.d_tlas_nodes.as<WNode>())
                    PT_HIP(ctx, hipMemcpyAsync(T.d_tlas_nodes.p, kept_nodes, sizeof(WNode) * (size_t)kept[0], hipMemcpyDeviceToDevice, s));
                k_requantise<<<div_up(kept[0], 256), 256, 0, s>>>(T.d_tlas_nodes.as<WNode>(), 0, kept[0], ctx->d_tlas_bnodes.as<BvhNode>(),
                                                                  T.d_obj_lo.as<float4>(), T.d_obj_hi.as<float4>());
                PT_HIP(ctx, hipGetLastError());
                std::memcpy(cinfo, kept, sizeof cinfo);
                ctx->tlas_refits++;
                refitted = true;
            } else {
                PT_HIP(ctx, ctx->d_tlas_bnodes.ensure(sizeof(BvhNode) * (size_t)(n - 1)));
                // A scene that arrives (first upload, or another number of objects) gets the surface-area tree from the host when
                // it is of moderate size: it is walked by every ray of every frame until the objects change, and refits keep its
                // topology. The periodic fresh build of a scene whose objects keep moving uses the fast device builder.
                const bool arrived = ctx->tlas_built_n != n && (ctx->tlas_built_n == 0 || ctx->tlas_stable_before >= kTlasSettledUploads);
                const bool sah = sah_possible && (arrived || upgrade);
                if (sah) ctx->tlas_upgrade_pending = false;
                else if (ctx->tlas_built_n != n) ctx->tlas_upgrade_pending = true;   // (not the periodic fresh build of moving objects)
                int rc = sah ? build_sah_host(ctx, T.d_obj_lo.as<float4>(), T.d_obj_hi.as<float4>(), 0, n, ctx->d_tlas_bnodes.as<BvhNode>(), 0, slot, false)
                             : build_lbvh(ctx, T.d_obj_lo.as<float4>(), T.d_obj_hi.as<float4>(), 0, n, ctx->d_tlas_bnodes.as<BvhNode>(), 0, slot, false);
                if (rc != PT_OK) return rc;
                PT_HIP(ctx, ctx->d_tlas_order.ensure(nb));
                PT_HIP(ctx, ctx->d_tlas_par_int.ensure(nb));
                PT_HIP(ctx, ctx->d_tlas_par_leaf.ensure(nb));
                PT_HIP(ctx, hipMemcpyAsync(ctx->d_tlas_order.p, ctx->t_vals0.p, nb, hipMemcpyDeviceToDevice, s));
                PT_HIP(ctx, hipMemcpyAsync(ctx->d_tlas_par_int.p, ctx->t_par_int.p, nb, hipMemcpyDeviceToDevice, s));
                PT_HIP(ctx, hipMemcpyAsync(ctx->d_tlas_par_leaf.p, ctx->t_par_leaf.p, nb, hipMemcpyDeviceToDevice, s));
                ctx->tlas_built_n = n;
                ctx->tlas_refits = 0;
                rc = collapse_wide(ctx, ctx->d_tlas_bnodes.as<BvhNode>(), 0, n, false, T.d_tlas_nodes.as<WNode>(), 0, slot + 4, slot);
                if (rc != PT_OK) return rc;
                PT_HIP(ctx, hipMemcpyAsync(cinfo, slot + 4, sizeof cinfo, hipMemcpyDeviceToHost, s));
            }
        }
        T.use_tlas = true;
    }
    PT_HIP(ctx, hipEventRecord(ctx->evb1, s));
    PT_HIP(ctx, hipStreamSynchronize(s));
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, ctx->evb0, ctx->evb1) == hipSuccess) ctx->build_ms = ms;
    T.tlas_n_wide = cinfo[0];
    T.tlas_depth = cinfo[1];
    T.tlas_root = want_tlas ? cinfo[2] : PT_NONE;
    T.tlas_need = cinfo[3];
    T.refitted = refitted;
    T.built = true;
    ctx->cur_set = target;
    return PT_OK;
}

// Entries of the per-lane traversal stack: what a TLAS walk can hold (the children it has put aside on its way down,
// counted exactly when the tree is made) plus what the walk of the deepest BLAS adds on top of it, plus the three words
// of headroom wide_visit() writes into. (LDS per workgroup decides how many workgroups fit on a compute unit.)
uint32_t stack_depth_needed(const pt_ctx* ctx) {
    const ObjectSet& os = ctx->sets[ctx->cur_set];
    uint32_t d = ctx->max_blas_need;
    if (os.use_tlas) d += os.tlas_need;
    return d + 3u;
}

int check_ready(pt_ctx* ctx, const char* who) {
    const ObjectSet& os = ctx->sets[ctx->cur_set];
    if (!ctx->have_meshes || !ctx->have_materials || !ctx->have_objects || !os.built)
        return fail(ctx, PT_ERR_NOT_READY, std::string(who) + ": meshes, materials and objects must all have been uploaded");
    if (os.n_objects) {
        if (os.obj_max_mesh >= ctx->n_meshes) return fail(ctx, PT_ERR_INVALID_SCENE, std::string(who) + ": an object references a mesh past meshes[]");
        if (os.obj_max_mat >= ctx->n_materials) return fail(ctx, PT_ERR_INVALID_SCENE, std::string(who) + ": an object references a material past materials[]");
    }
    if (stack_depth_needed(ctx) > kMaxStackDepth)
        return fail(ctx, PT_ERR_UNSUPPORTED, std::string(who) + ": acceleration structure deeper than the traversal stack; create the context with PT_ACCEL_BRUTE");
    return PT_OK;
}

SceneView make_view(const pt_ctx* ctx) {
    const ObjectSet& os = ctx->sets[ctx->cur_set];
    SceneView v;
    v.objects = os.d_objects.as<DObject>();
    v.obj_l2w = os.d_obj_l2w.as<Affine>();
    v.meshes = ctx->d_meshes.as<DMesh>();
    v.tris = ctx->d_tris.as<DTri>();
    v.tri_ids = ctx->d_tri_ids.as<uint32_t>();
    v.tri_attrs = ctx->d_attrs.as<DTriAttr>();
    v.blas_nodes = ctx->d_blas_nodes.as<WNode>();
    v.tlas_nodes = os.d_tlas_nodes.as<WNode>();
    v.obj_lo = os.d_obj_lo.as<float4>();
    v.obj_hi = os.d_obj_hi.as<float4>();
    v.materials = ctx->d_materials.as<DMaterial>();
    v.textures = ctx->d_textures.as<DTexture>();
    v.texture_data = ctx->d_texdata.as<float>();
    v.n_objects = os.n_objects;
    v.n_texture_data = ctx->n_texdata;
    v.tlas_root = os.tlas_root;
    v.tie_lowest = (ctx->cfg.tie_rule == PT_TIE_LOWEST) ? 1u : 0u;
    v.cull_boxes = (ctx->cfg.accel == PT_ACCEL_BRUTE) ? 0u : 1u;
    v._pad0 = 0;
    v._pad1 = 0;
    v.obj_slab = nullptr;
    v.lb_nodes = nullptr; v.lb_tris = nullptr; v.lb_tri_ids = nullptr;
    v.lb_mesh = PT_NONE; v.lb_node_base = 0;
    v.tlas_top = nullptr; v.tlas_top_n = 0;
    return v;
}

// Wait for the frame recorded in `sl` and take its timings and counters into the lane's statistics.
int finish_slot(pt_ctx* ctx, Lane& L, FrameSlot& sl) {
    if (!sl.pending) return PT_OK;
    PT_HIP(ctx, hipEventSynchronize(sl.ev1));
    float ms = 0.0f;
    PT_HIP(ctx, hipEventElapsedTime(&ms, sl.ev0, sl.ev1));
    const DCounters c = *sl.h_counters;   // (copied behind the frame, before the event)
    float ms_primary = 0.0f;
    PT_HIP(ctx, hipEventElapsedTime(&ms_primary, sl.ev0, sl.evm));
    if (ctx->debug) std::fprintf(stderr, "[pt_amd] frame done: %.3f ms, work / longest chunk %u waves, %u chunks in parts, workgroups wanted %u, other frames running at its start %u\n", ms, c.waves_needed, c.n_split, c.n_groups, c.others_running);
    PtStats& st = L.stats;
    st.trace_ms = ms;
    st.primary_ms = ms_primary;
    st.paths_ms = ms - ms_primary;
    st.trace_launches = 2;
    st.rays = c.rays + sl.primary_traced;
    st.xforms = c.xforms;
    st.box_tests = c.box_tests;
    st.tri_tests = c.tri_tests;
    st.shades = c.shades;
    st.hits = c.hits;
    st.primary_xforms = st.primary_box_tests = st.primary_tri_tests = st.primary_hits = 0;
    if (sl.detail) {
        const DCounters k1 = *sl.h_counters_k1;
        st.primary_xforms = k1.xforms;
        st.primary_box_tests = k1.box_tests;
        st.primary_tri_tests = k1.tri_tests;
        st.primary_hits = k1.hits;
    }
    st.cycles_refill = c.cyc_refill;
    st.cycles_trace = c.cyc_trace;
    st.cycles_shade = c.cyc_shade;
    st.wave_rounds = c.rounds;
    st.lane_rounds = c.lane_rounds;
    st.object_passes = c.passes;
    L.last_n_paths = c.n_paths;
    if (c.waves_needed != 0u) ctx->last_waves_needed = c.waves_needed;
    st.wave_life_sum = c.wave_life_sum;
    st.wave_life_max = c.wave_life_max;
    st.waves = c.waves;
    for (int k = 0; k < 32; ++k) st.wave_life_hist[k] = c.life_hist[k];
    st.paths = sl.paths;
    st.rays_reference = c.rays + sl.primary_traced + sl.primary;
    st.accel_used = sl.accel_used;
    sl.pending = false;
    return PT_OK;
}

// The frames in flight of one lane, oldest first, so that its statistics end up describing the most recent one.
int finish_lane(pt_ctx* ctx, Lane& L) {
    int rc = finish_slot(ctx, L, L.slot[L.frame_seq & 1]);
    if (rc != PT_OK) return rc;
    return finish_slot(ctx, L, L.slot[(L.frame_seq + 1) & 1]);
}

// Every frame in flight of the context, whichever entry point enqueued it.
int finish_all(pt_ctx* ctx) {
    for (Lane& L : ctx->lanes) {
        if (!L.ready) continue;
        int rc = finish_lane(ctx, L);
        if (rc != PT_OK) return rc;
    }
    if (ctx->noalloc_pending) {
        PT_HIP(ctx, hipEventSynchronize(ctx->ev_noalloc));
        ctx->noalloc_pending = false;
    }
    return PT_OK;
}

// Which closest-hit organisation the frame kernels use, and the LDS layout of the staged scene for PT_MODE_LDS:
// scenes with at most 32 objects whose record arrays (object boxes, instances, meshes, triangles, attributes, BLAS
// nodes, materials, texture descriptors) fit in kLdsSceneMaxBytes are copied into LDS by every workgroup.
constexpr uint32_t kLdsSceneMaxBytes = 24 * 1024;
int scene_mode(const pt_ctx* ctx, LdsScene* ls) {
    std::memset(ls, 0, sizeof *ls);
    const ObjectSet& os = ctx->sets[ctx->cur_set];
    if (os.use_tlas) return PT_MODE_TLAS;
    if (os.n_objects == 0 || os.n_objects > 32 || ctx->cfg.accel == PT_ACCEL_BRUTE || !ctx->lds_scene) return PT_MODE_LIST;
    uint32_t off = 0;
    auto place = [&](uint32_t bytes) { uint32_t o = off; off += (bytes + 15u) & ~15u; return o; };
    ls->bytes_obj_box = 16u * os.n_objects;
    ls->bytes_objects = (uint32_t)sizeof(DObject) * os.n_objects;
    ls->bytes_l2w = (uint32_t)sizeof(Affine) * os.n_objects;
    ls->bytes_meshes = (uint32_t)sizeof(DMesh) * ctx->n_meshes;
    ls->bytes_tris = (uint32_t)sizeof(DTri) * ctx->n_tris;
    ls->bytes_tri_ids = ctx->n_blas_nodes ? (uint32_t)sizeof(uint32_t) * ctx->n_tris : 0u;   // only BLAS leaves look at them
    ls->bytes_attrs = (uint32_t)sizeof(DTriAttr) * ctx->n_tris;
    ls->bytes_blas = (uint32_t)sizeof(WNode) * ctx->n_blas_nodes;
    ls->bytes_materials = (uint32_t)sizeof(DMaterial) * ctx->n_materials;
    ls->bytes_textures = (uint32_t)sizeof(DTexture) * ctx->n_textures;
    const uint64_t sum = 5ull * ls->bytes_obj_box + ls->bytes_objects + ls->bytes_l2w + ls->bytes_meshes + ls->bytes_tris + ls->bytes_tri_ids +
                         ls->bytes_attrs + ls->bytes_blas + ls->bytes_materials + ls->bytes_textures + 64ull;
    if (sum > kLdsSceneMaxBytes) return PT_MODE_LIST;
    ls->off_obj_slab = place(3u * ls->bytes_obj_box);   // 12 floats per object
    ls->off_obj_lo = place(ls->bytes_obj_box);
    ls->off_obj_hi = place(ls->bytes_obj_box);
    ls->off_objects = place(ls->bytes_objects);
    ls->off_l2w = place(ls->bytes_l2w);
    ls->off_meshes = place(ls->bytes_meshes);
    ls->off_tris = place(ls->bytes_tris);
    ls->off_tri_ids = place(ls->bytes_tri_ids);
    ls->off_attrs = place(ls->bytes_attrs);
    ls->off_blas = place(ls->bytes_blas);
    ls->off_materials = place(ls->bytes_materials);
    ls->off_textures = place(ls->bytes_textures);
    ls->total_bytes = off;
    return PT_MODE_LDS;
}

// Device memory a frame works in besides the scene and its image.
struct FrameWork {
    uint32_t *aux_obj, *aux_flat, *aux_rays, *cost, *sweeps_done;
    float* aux_t;
    PathRec* recs;
    PathCarry* carry;
    DCounters* counters;
};

// One thread: takes back what k_chunk_order signed in when the k_paths launch that would have signed it off failed.
__global__ void k_sign_off() { atomicSub(&g_frames_running, 1u); }

// Enqueue one frame of lane `L` on `s`, image to device memory `out`; with `out16` the image is also packed to binary16
// there, inside the span the frame's completion event covers. `noalloc`: the frame works in the device's static array,
// keeps no record in the context (no statistics, no aux planes to read back, no schedule hints) and nothing is allocated.
int render_enqueue(pt_ctx* ctx, Lane& L, const PtFrame* f, float* out, hipStream_t s, uint2* out16 = nullptr, bool noalloc = false) {
    int rc = check_ready(ctx, "pt_render");
    if (rc != PT_OK) return rc;
    if (f->width == 0 || f->height == 0) return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_render: zero-sized frame");
    if ((uint64_t)f->width * f->height > (1ull << 28)) return fail(ctx, PT_ERR_UNSUPPORTED, "pt_render: frame larger than 2^28 pixels");
    const uint32_t pc = f->part_count ? f->part_count : 1u;
    if (f->part_index >= pc) return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_render: part_index >= part_count");
    if (f->settings.samples > (1u << 30)) return fail(ctx, PT_ERR_UNSUPPORTED, "pt_render: more than 2^30 samples per pixel");
    const size_t n_pix = (size_t)f->width * f->height;
    const size_t n_padded = (size_t)div_up(f->width, 8) * div_up(f->height, 8) * 64;   // whole 8x8 tiles
    FrameSlot& sl = L.slot[L.frame_seq & 1];
    const ObjectSet& os = ctx->sets[ctx->cur_set];
    FrameWork w;
    bool capturing = false;
    if (noalloc) {
        if (!ctx->scratch) return fail(ctx, PT_ERR_UNSUPPORTED, "pt_render_device_noalloc: no work array on this device");
        if (n_padded > kNoallocPix)
            return fail(ctx, PT_ERR_UNSUPPORTED, "pt_render_device_noalloc: frame larger than PT_NOALLOC_MAX_PIXELS (" + std::to_string(kNoallocPix) +
                                                     " pixels, sizes rounded up to multiples of 8); use pt_render_device");
        if (f->flags != 0u) return fail(ctx, PT_ERR_UNSUPPORTED, "pt_render_device_noalloc: frame flags are not supported");
        // frames of the allocating entry points still in flight: their records are read first (not under stream capture,
        // where nothing may be waited for: the caller then has to have waited for them)
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
        capturing = cap != hipStreamCaptureStatusNone;
        if (capturing) {
            for (const Lane& o : ctx->lanes)
                if (o.slot[0].pending || o.slot[1].pending)
                    return fail(ctx, PT_ERR_NOT_READY, "pt_render_device_noalloc: frames of the other entry points are pending; wait for them (pt_get_stats) before capturing");
        } else {
            rc = finish_all(ctx);
            if (rc != PT_OK) return rc;
        }
        unsigned char* b = ctx->scratch;
        w.recs = reinterpret_cast<PathRec*>(b + kScrRecs);
        w.carry = reinterpret_cast<PathCarry*>(b + kScrCarry);
        w.aux_obj = reinterpret_cast<uint32_t*>(b + kScrAux);
        w.aux_flat = w.aux_obj + kNoallocPix;
        w.aux_t = reinterpret_cast<float*>(w.aux_obj + 2 * kNoallocPix);
        w.aux_rays = w.aux_obj + 3 * kNoallocPix;
        w.cost = reinterpret_cast<uint32_t*>(b + kScrCost);
        w.sweeps_done = reinterpret_cast<uint32_t*>(b + kScrSweeps);
        w.counters = reinterpret_cast<DCounters*>(b + kScrCounters);
        L.aux_w = L.aux_h = 0;   // the lane's aux planes no longer describe the last frame
        L.hint_valid = false;
    } else {
        // This frame's record replaces the one of the frame before the previous one, which has to be complete; the
        // previous frame may still be running (same stream: its kernels come first). Anything that is not plain
        // back-to-back rendering on one stream waits for everything.
        const FrameSlot& prev = L.slot[(L.frame_seq + 1) & 1];
        if (s != L.last_frame_stream || ctx->stats_detail || prev.detail || !ctx->queue_ahead) rc = finish_lane(ctx, L);
        else rc = finish_slot(ctx, L, sl);
        if (rc != PT_OK) return rc;
        if ((L.aux_w != f->width || L.aux_h != f->height) && prev.pending) {   // buffers are about to be re-allocated
            rc = finish_lane(ctx, L);
            if (rc != PT_OK) return rc;
        }
        PT_HIP(ctx, L.d_aux_obj.ensure(4 * n_pix));
        PT_HIP(ctx, L.d_aux_flat.ensure(4 * n_pix));
        PT_HIP(ctx, L.d_aux_t.ensure(4 * n_pix));
        PT_HIP(ctx, L.d_aux_rays.ensure(4 * n_pix));
        L.aux_w = f->width;
        L.aux_h = f->height;
        // path list: at most one entry per pixel (of this partition: sized for the whole frame, partitions are rare)
        PT_HIP(ctx, L.d_paths.ensure(sizeof(PathRec) * n_padded));
        PT_HIP(ctx, L.d_cost.ensure(sizeof(uint32_t) * n_padded));
        w.aux_obj = L.d_aux_obj.as<uint32_t>();
        w.aux_flat = L.d_aux_flat.as<uint32_t>();
        w.aux_t = L.d_aux_t.as<float>();
        w.aux_rays = L.d_aux_rays.as<uint32_t>();
        w.recs = L.d_paths.as<PathRec>();
        w.cost = L.d_cost.as<uint32_t>();
        w.counters = sl.counters.as<DCounters>();
        w.sweeps_done = nullptr;   // allocated below, when the frame runs in sweeps
        w.carry = nullptr;
    }

    FrameParams fp;
    std::memcpy(fp.wfv, f->world_from_view, sizeof fp.wfv);
    std::memcpy(fp.vfc, f->view_from_clip, sizeof fp.vfc);
    fp.width = f->width;
    fp.height = f->height;
    fp.time = f->time;
    fp.bounces = f->settings.bounces;
    fp.samples = f->settings.samples;
    fp.sky[0] = f->settings.sky_color[0];
    fp.sky[1] = f->settings.sky_color[1];
    fp.sky[2] = f->settings.sky_color[2];
    fp.part_index = f->part_index;
    fp.part_count = pc;
    fp.tiles_x = div_up(f->width, 8);
    fp.tiles_y = div_up(f->height, 8);
    const uint32_t n_tiles = fp.tiles_x * fp.tiles_y;
    fp.n_local_tiles = (f->part_index < n_tiles) ? (n_tiles - f->part_index + pc - 1) / pc : 0u;
    PT_HIP(ctx, hipMemsetAsync(w.counters, 0, sizeof(DCounters), s));
    if (pc > 1) {
        PT_HIP(ctx, hipMemsetAsync(out, 0, 16 * n_pix, s));
        PT_HIP(ctx, hipMemsetAsync(w.aux_obj, 0xFF, 4 * n_pix, s));
        PT_HIP(ctx, hipMemsetAsync(w.aux_flat, 0xFF, 4 * n_pix, s));
        PT_HIP(ctx, hipMemsetAsync(w.aux_t, 0, 4 * n_pix, s));
        PT_HIP(ctx, hipMemsetAsync(w.aux_rays, 0, 4 * n_pix, s));
    }
    // Per-pixel ray counts of the previous frame, still in the aux plane, predict what this frame's pixels cost when
    // it is a frame of the same size and sample count (the usual case: a renderer that draws frame after frame; a camera
    // or an object that has moved a little since leaves most pixels looking at what they looked at). They only decide
    // the order in which k_paths takes the pixels, never a result. strict_hints: only for the same view of the same
    // scene, compared bytewise.
    FrameKey key;
    std::memset(&key, 0, sizeof key);
    key.scene_version = ctx->scene_version;
    key.width = f->width; key.height = f->height;
    key.samples = f->settings.samples; key.bounces = f->settings.bounces;
    key.part_index = f->part_index; key.part_count = pc;
    if (ctx->strict_hints) {
        std::memcpy(key.wfv, f->world_from_view, sizeof key.wfv);
        std::memcpy(key.vfc, f->view_from_clip, sizeof key.vfc);
    } else {
        key.scene_version = 0;
    }
    const bool use_hint = !noalloc && L.hint_valid && std::memcmp(&key, &L.hint_key, sizeof key) == 0 && ctx->cost_hints &&
                          f->settings.samples >= 8u && f->settings.bounces > 1u;
    if (!noalloc) L.hint_valid = false;   // until this frame has been enqueued completely
    SceneView sv = make_view(ctx);
    const uint32_t depth = stack_depth_needed(ctx);
    const bool st = ctx->stats_detail && !noalloc;
    if (!noalloc) sl.detail = st;
    float4* o4 = reinterpret_cast<float4*>(out);
    uint32_t* ao = w.aux_obj;
    uint32_t* af = w.aux_flat;
    float* at = w.aux_t;
    uint32_t* ar = w.aux_rays;
    PathRec* recs = w.recs;
    DCounters* dc = w.counters;

    if (!noalloc) PT_HIP(ctx, hipEventRecord(sl.ev0, s));
    // 1. primary pass: one wave per tile
    LdsScene ls;
    const int mode = scene_mode(ctx, &ls);
    {
        const size_t scene_lds = (mode == PT_MODE_LDS ? ls.total_bytes : 0u);
        const bool wide = (size_t)16 * depth * 256 + scene_lds <= 64 * 1024 && !ctx->k1_narrow;   // 16 waves per workgroup when their stacks fit
        const uint32_t waves = wide ? 16u : 4u;
        const size_t lds = (size_t)waves * depth * 64 * sizeof(uint32_t) + scene_lds;
        const uint32_t grid = (wide && pc == 1u) ? div_up(fp.tiles_x, 4) * div_up(fp.tiles_y, 4) : div_up(fp.n_local_tiles, waves);
        using PrimaryFn = void (*)(PT_SCENE_PARAMS, LdsScene, FrameParams, uint32_t, float4*, uint32_t*, uint32_t*, float*, uint32_t*, PathRec*, uint32_t*, uint32_t, DCounters*);
        static const PrimaryFn table[3][2][2] = {
            {{k_primary<PT_MODE_LIST, false, 4>, k_primary<PT_MODE_LIST, false, 16>}, {k_primary<PT_MODE_LIST, true, 4>, k_primary<PT_MODE_LIST, true, 16>}},
            {{k_primary<PT_MODE_TLAS, false, 4>, k_primary<PT_MODE_TLAS, false, 16>}, {k_primary<PT_MODE_TLAS, true, 4>, k_primary<PT_MODE_TLAS, true, 16>}},
            {{k_primary<PT_MODE_LDS, false, 4>, k_primary<PT_MODE_LDS, false, 16>}, {k_primary<PT_MODE_LDS, true, 4>, k_primary<PT_MODE_LDS, true, 16>}}};
        if (grid) {
            table[mode][st ? 1 : 0][wide ? 1 : 0]<<<grid, waves * 64, lds, s>>>(PT_SCENE_ARGS(sv), ls, fp, depth, o4, ao, af, at, ar, recs, w.cost, use_hint ? 1u : 0u, dc);
            PT_HIP(ctx, hipGetLastError());
        }
    }
    if (st) {
        PT_HIP(ctx, hipMemcpyAsync(sl.h_counters_k1, sl.counters.p, sizeof(DCounters), hipMemcpyDeviceToHost, s));
    }
    if (!noalloc) PT_HIP(ctx, hipEventRecord(sl.evm, s));
    // 2. path pass: persistent waves, as many workgroups as fit on the chip at once (fewer for small frames); they
    //    pull list entries from a counter until the list is exhausted
    if (fp.bounces > 0 && fp.samples > 0 && fp.n_local_tiles) {
        PathParams pp;
        pp.origin[0] = fp.wfv[12]; pp.origin[1] = fp.wfv[13]; pp.origin[2] = fp.wfv[14];
        pp.sky[0] = fp.sky[0]; pp.sky[1] = fp.sky[1]; pp.sky[2] = fp.sky[2];
        pp.bounces = fp.bounces;
        pp.samples = fp.samples;
        pp.chunk = ctx->chunk_pixels;
        // few objects: keep a wave on one tile's worth of pixels at a time (similar candidates); many objects: keep every lane busy
        pp.refill_min = ctx->refill_min ? ctx->refill_min : (os.use_tlas ? 1u : 64u);
        // PT_MODE_TLAS: the first lds_stack_max entries of a lane's traversal stack live in LDS, the rest (seldom reached)
        // in global memory, so that the stack does not decide how many workgroups a compute unit holds
        const uint32_t depth_paths = (mode == PT_MODE_TLAS && !noalloc) ? std::min(depth, ctx->lds_stack_max) : depth;
        const size_t lds = ((size_t)4 * (depth_paths + lane_slots(mode)) * 64 + 4 * PT_SCHED_WORDS) * sizeof(uint32_t) + (mode == PT_MODE_LDS ? ls.total_bytes : 0u);
        using PathsFn = void (*)(PT_SCENE_PARAMS, LdsScene, LdsBlas, PathParams, uint32_t, const PathRec*, PathCarry*, uint32_t*, const uint32_t*, const ChunkOrderInfo*, float4*,
                                 uint32_t*, DCounters*, uint4*);
        static const PathsFn ptable[4][2] = {{k_paths<PT_MODE_LIST, false>, k_paths<PT_MODE_LIST, true>},
                                             {k_paths<PT_MODE_TLAS, false>, k_paths<PT_MODE_TLAS, true>},
                                             {k_paths<PT_MODE_LDS, false>, k_paths<PT_MODE_LDS, true>},
                                             {k_paths<PT_MODE_LIST_LB, false>, k_paths<PT_MODE_LIST_LB, true>}};
        static const PathsFn ptable_roomy[4][2] = {{k_paths<PT_MODE_LIST, false, true>, k_paths<PT_MODE_LIST, true, true>}, {nullptr, nullptr}, {nullptr, nullptr},
                                                   {k_paths<PT_MODE_LIST_LB, false, true>, k_paths<PT_MODE_LIST_LB, true, true>}};
        // Frames of other slots that are still on the device as this one is enqueued: how deep the caller's pipeline really
        // is (a frame that has finished and only waits to be collected does not count: a caller who is paced by something
        // else, a display say, may hold many of those while each of its frames has the device to itself).
        uint32_t n_others_pending = 0;
        if (!noalloc)
            for (const Lane& o : ctx->lanes) {   // (slots only: a synchronous frame stays "pending" until its statistics are read)
                if (&o == &L || &o == &ctx->lanes[0] || !o.ready) continue;
                bool running = false;
                for (const FrameSlot& fs : o.slot)
                    if (fs.pending && fs.ev1 && hipEventQuery(fs.ev1) == hipErrorNotReady) running = true;
                if (running) n_others_pending++;
            }
        (void)hipGetLastError();   // (hipErrorNotReady is an answer, not an error to keep)
        // With five or more frames in flight an object-list scene is served for throughput: no BLAS in LDS and no chunks
        // in parts, both of which shorten a frame's longest chains at the price of workgroups per compute unit and idle
        // lanes -- the other frames' waves cover those chains instead (scene.glb with eight slots: 8.8 against 9.8 ms per
        // frame; with four the two arrangements are level, with three this one loses).
        const bool deep_pipeline = n_others_pending >= ctx->deep_pipeline_slots;
        // Object-list scenes with a big mesh: its whole BLAS (and its triangles if they fit as well) goes into LDS,
        // at the price of fewer workgroups per compute unit. Rays in such a mesh are chains of dependent node
        // visits, each a round trip to the L2 otherwise.
        LdsBlas lb;
        std::memset(&lb, 0, sizeof lb);
        int paths_mode = mode;
        // Only when the last finished frame of this view listed so few pixels that one workgroup per compute unit
        // takes them all anyway (a frame full of such pixels is better served by more workgroups).
        if (mode == PT_MODE_LIST && ctx->lds_blas && !deep_pipeline && use_hint && L.last_n_paths <= (uint64_t)ctx->n_cus * 256u * 4u) {
            const size_t budget = (size_t)ctx->lds_per_cu > lds + 1024 ? (size_t)ctx->lds_per_cu - 1024 - lds : 0;
            uint32_t best = PT_NONE;
            for (uint32_t m = 0; m < ctx->n_meshes; ++m) {
                if (ctx->h_blas_nodes_base[m] == PT_NONE || ctx->h_blas_n_wide[m] == 0) continue;
                if (lds_blas_bytes(ctx->h_blas_n_wide[m], 0) <= budget && (best == PT_NONE || ctx->h_meshes[m].tri_count > ctx->h_meshes[best].tri_count)) best = m;
            }
            if (best != PT_NONE) {
                lb.mesh = best;
                lb.node_base = ctx->h_blas_nodes_base[best];
                lb.n_nodes = ctx->h_blas_n_wide[best];
                lb.tri_first = ctx->h_meshes[best].tri_first;
                // its triangle records too, if two workgroups still fit on a compute unit then (two waves per SIMD hide each
                // other's LDS and memory waits; one wave alone does not: scene.glb 17.0 against 20.2 ms per frame)
                const size_t with_tris = lds + lds_blas_bytes(lb.n_nodes, ctx->h_meshes[best].tri_count);
                lb.n_tris = (2 * with_tris + 2048 <= (size_t)ctx->lds_per_cu && !ctx->lds_blas_nodes_only) ? ctx->h_meshes[best].tri_count : 0u;
                paths_mode = PT_MODE_LIST_LB;
            }
        }
        // PT_MODE_TLAS: the top levels of the object tree, the nodes every ray passes, are copied into LDS
        pp.tlas_top_n = 0;
        if (mode == PT_MODE_TLAS) {
            // as many as fit without costing a workgroup per compute unit (registers allow four)
            const size_t share = ctx->lds_per_cu / 4 - 512;
            const uint32_t fit = share > lds ? (uint32_t)((share - lds) / sizeof(WNode)) : 0u;
            pp.tlas_top_n = std::min(os.tlas_n_wide, ctx->tlas_top_max == PT_NONE ? fit : ctx->tlas_top_max);
        }
        const size_t lds_paths = lds + (paths_mode == PT_MODE_LIST_LB ? lds_blas_bytes(lb.n_nodes, lb.n_tris) : 0u) + (size_t)pp.tlas_top_n * sizeof(WNode);
        // Object-list scenes whose last measured frame (of any slot, of any view) had work for few waves (the sum of its
        // chunk lengths over the longest: see ChunkOrderInfo) run the build with more registers and fewer waves per SIMD,
        // which keeps nothing in scratch memory; so do frames of a context that has no such measurement yet (its first
        // frame, or cost hints switched off): scene.glb without hints 11.9 against 13.0 ms per frame, a scene of twelve
        // overlapping meshes, which has many times the work, 23.7 against 22.0 for its first frame only.
        bool roomy = false;
        if (paths_mode == PT_MODE_LIST || paths_mode == PT_MODE_LIST_LB) {
            const uint64_t roomy_waves = (uint64_t)ctx->n_cus * 4u * PT_PATHS_MIN_WAVES_LIST_ROOMY;
            roomy = ctx->list_regs == 2 || (ctx->list_regs == 0 && (ctx->last_waves_needed == 0u || (uint64_t)ctx->last_waves_needed <= 2u * roomy_waves));
        }
        PathsFn fn = (roomy ? ptable_roomy : ptable)[paths_mode][st ? 1 : 0];
        if (!noalloc) L.last_schedule = 0x80000000u | (uint32_t)paths_mode | (roomy ? 0x10u : 0u) | (deep_pipeline ? 0x20u : 0u) | (std::min(n_others_pending, 15u) << 8);
        if (lds_paths > 64 * 1024) PT_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_paths));
        int per_cu = 0;
        PT_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(fn), 256, lds_paths));
        if (per_cu < 1) return fail(ctx, PT_ERR_UNSUPPORTED, "pt_render: the path kernel does not fit on a compute unit with this traversal stack");
        if (ctx->blocks_per_cu >= 1 && ctx->blocks_per_cu <= per_cu) per_cu = ctx->blocks_per_cu;
        const uint32_t grid = std::min(div_up(fp.n_local_tiles, 4), (uint32_t)per_cu * ctx->n_cus);
        // Sample slices ("sweeps"): half of the samples, then a quarter, ... so that the items handed out last are
        // short. Only with whole-chunk refill (a wave then owns one chunk at a time and can report it complete).
        pp.n_sweeps = 1;
        pp.walk_quota = ctx->walk_quota;
        pp.split_chunks = (ctx->split_chunks && !(deep_pipeline && mode == PT_MODE_LIST)) ? 1u : 0u;
        pp.carry_acquire = ctx->carry_acquire ? 1u : 0u;
        pp.sweep_end[0] = fp.samples;
        // Not for a frame that will run beside other frames (k_chunk_order has the reason). The kernel sees that for itself
        // when another frame of the process is running as it is scheduled; the host sees the caller's pattern earlier: a
        // frame on a slot gets sweeps only when this and the last two submits found no other slot's frame on the device, so a caller
        // that keeps frames in flight never has a frame with sweeps among them, the first frames included.
        bool beside_others = false;
        if (!noalloc) {
            const bool others_pending = n_others_pending != 0u;
            if (&L != &ctx->lanes[0]) ctx->alone_streak = others_pending ? 0u : std::min(ctx->alone_streak + 1u, 1000u);
            beside_others = !ctx->sweeps_when_shared && (others_pending || (&L != &ctx->lanes[0] && ctx->alone_streak < 3u));
        }
        if (pp.refill_min == 64u && pp.chunk == 64u && fp.samples >= 16u && ctx->max_sweeps > 1u && !beside_others) {
            uint32_t n = 0, done = 0, left = fp.samples;
            while (n + 1u < ctx->max_sweeps && left >= 8u) {
                const uint32_t len = left / 2u;
                done += len;
                left -= len;
                pp.sweep_end[n++] = done;
            }
            pp.sweep_end[n++] = fp.samples;
            pp.n_sweeps = n;
        }
        if (pp.n_sweeps > 1u) {
            const size_t bytes = sizeof(uint32_t) * (size_t)fp.n_local_tiles;   // one flag per chunk of 64 entries
            if (!noalloc) {
                PT_HIP(ctx, L.d_sweeps_done.ensure(bytes));
                PT_HIP(ctx, L.d_carry.ensure(sizeof(PathCarry) * (size_t)fp.n_local_tiles * 64));
                w.sweeps_done = L.d_sweeps_done.as<uint32_t>();
                w.carry = L.d_carry.as<PathCarry>();
            }
            PT_HIP(ctx, hipMemsetAsync(w.sweeps_done, 0, bytes, s));
        }
        uint4* wlog = nullptr;
        if (!noalloc) L.wave_log_records = 0;
        if (st) {
            PT_HIP(ctx, L.d_wave_log.ensure((size_t)64 * grid * 4));
            PT_HIP(ctx, hipMemsetAsync(L.d_wave_log.p, 0, (size_t)64 * grid * 4, s));
            wlog = L.d_wave_log.as<uint4>();
            L.wave_log_records = grid * 4u;
        }
        if (ctx->debug) std::fprintf(stderr, "[pt_amd] k_paths: mode=%d roomy=%d per_cu=%d grid=%u lds=%zu depth=%u/%u tlas: nodes=%u levels=%u need=%u top=%u chunk=%u refill_min=%u sweeps=%u last_n_paths=%llu\n", paths_mode, (int)roomy, per_cu, grid, lds_paths, depth_paths, depth, os.tlas_n_wide, os.tlas_depth, os.tlas_need, pp.tlas_top_n, pp.chunk, pp.refill_min, pp.n_sweeps, (unsigned long long)L.last_n_paths);
        // (everything that can fail comes before k_chunk_order, which signs the frame in as running on the device)
        pp.stack_overflow = nullptr;
        if (depth_paths < depth) {
            PT_HIP(ctx, L.d_stack_overflow.ensure(sizeof(uint32_t) * (size_t)(depth - depth_paths) * grid * 256));
            pp.stack_overflow = L.d_stack_overflow.as<uint32_t>();
        }
        const uint32_t* chunk_order = nullptr;
        const ChunkOrderInfo* order_info = nullptr;
        if (use_hint && pp.chunk == PT_CHUNK) {
            const uint32_t max_chunks = fp.n_local_tiles;   // 64 entries per tile at most
            PT_HIP(ctx, L.d_chunk_bin.ensure(sizeof(uint32_t) * max_chunks));
            PT_HIP(ctx, L.d_chunk_order.ensure(sizeof(uint32_t) * max_chunks + sizeof(ChunkOrderInfo)));
            uint32_t* order = L.d_chunk_order.as<uint32_t>();
            ChunkOrderInfo* info = reinterpret_cast<ChunkOrderInfo*>(order + max_chunks);
            k_chunk_keys<<<div_up(max_chunks, 4), 256, 0, s>>>(w.cost, dc, fp.samples, L.d_chunk_bin.as<uint32_t>());
            PT_HIP(ctx, hipGetLastError());
            k_chunk_order<<<1, 1024, 0, s>>>(L.d_chunk_bin.as<uint32_t>(), dc, order, info, ctx->split_fraction[0], ctx->split_fraction[1], ctx->group_margin_8ths[0], ctx->group_margin_8ths[1], ctx->n_cus, grid * 4u, ctx->split_load_8ths, ctx->sweeps_when_shared ? 1u : (pp.n_sweeps == 1u ? 2u : 0u));
            if (hipGetLastError() == hipSuccess) {   // (not launched: the frame was not signed in and runs in list order)
                chunk_order = order;
                order_info = info;
            }
        }
        fn<<<grid, 256, lds_paths, s>>>(PT_SCENE_ARGS(sv), ls, lb, pp, depth_paths, recs, w.carry, w.sweeps_done, chunk_order,
                                  order_info, o4, ar, dc, wlog);
        {
            const hipError_t e_launch = hipGetLastError();
            if (e_launch != hipSuccess) {
                // the last wave of k_paths would have signed the frame off: do it for it, or every later frame of the process
                // on this device would be scheduled as sharing it
                if (chunk_order) k_sign_off<<<1, 1, 0, s>>>();
                (void)hipGetLastError();
                return fail(ctx, PT_ERR_DEVICE, std::string("pt_render: k_paths launch: ") + hipGetErrorString(e_launch));
            }
        }
    }
    if (noalloc) {
        // (no record of this frame is kept; the event lets uploads and pt_destroy wait for it -- recording allocates nothing)
        if (!capturing) {
            PT_HIP(ctx, hipEventRecord(ctx->ev_noalloc, s));
            ctx->noalloc_pending = true;
            ctx->noalloc_set = ctx->cur_set;
        }
        return PT_OK;
    }
    L.hint_key = key;
    L.hint_valid = f->settings.samples > 0u && f->settings.bounces > 0u;
    // pixels of this partition
    uint64_t local_pixels = 0;
    if (pc == 1) {
        local_pixels = n_pix;
    } else {
        for (uint32_t k = f->part_index; k < n_tiles; k += pc) {
            uint32_t ty = k / fp.tiles_x, tx = k - ty * fp.tiles_x;
            uint32_t w = std::min(8u, f->width - tx * 8), h = std::min(8u, f->height - ty * 8);
            local_pixels += (uint64_t)w * h;
        }
    }
    sl.paths = local_pixels * fp.samples;
    sl.primary = (fp.bounces > 0 && fp.samples > 0) ? local_pixels * (uint64_t)(fp.samples - 1) : 0;
    sl.primary_traced = (fp.bounces > 0 && fp.samples > 0) ? local_pixels : 0;  // one primary ray per pixel (k_primary)

    if (f->flags & (PT_FRAME_ACCUMULATE | PT_FRAME_RESET)) {
        // One accumulation buffer per context; frames add to it in the order they were enqueued, whichever streams they
        // run on (an f32 sum depends on its order): each waits for the previous one's addition.
        const bool restart = (f->flags & PT_FRAME_RESET) || ctx->acc_w != f->width || ctx->acc_h != f->height || ctx->acc_frames == 0;
        if (ctx->acc_pending && ctx->acc_stream != s) {
            if (restart && ctx->d_acc.cap < 16 * n_pix) PT_HIP(ctx, hipEventSynchronize(ctx->ev_acc));   // about to be re-allocated
            else PT_HIP(ctx, hipStreamWaitEvent(s, ctx->ev_acc, 0));
        }
        // (acc_frames == 0: nothing accumulated yet, or the scene changed since: the old sum belongs to another scene)
        if (restart) {
            if (ctx->d_acc.cap < 16 * n_pix && ctx->acc_pending) PT_HIP(ctx, hipEventSynchronize(ctx->ev_acc));
            PT_HIP(ctx, ctx->d_acc.ensure(16 * n_pix));
            PT_HIP(ctx, hipMemsetAsync(ctx->d_acc.p, 0, 16 * n_pix, s));
            ctx->acc_w = f->width;
            ctx->acc_h = f->height;
            ctx->acc_frames = 0;
        }
        if (f->flags & PT_FRAME_ACCUMULATE) {
            ctx->acc_frames += 1;
            k_accumulate<<<div_up((uint32_t)n_pix, 256), 256, 0, s>>>(ctx->d_acc.as<float4>(), reinterpret_cast<float4*>(out),
                                                                      (uint32_t)n_pix, (float)ctx->acc_frames);
            PT_HIP(ctx, hipGetLastError());
        }
        PT_HIP(ctx, hipEventRecord(ctx->ev_acc, s));
        ctx->acc_pending = true;
        ctx->acc_stream = s;
    }
    if (out16) {
        k_pack_f16<<<div_up((uint32_t)n_pix, 256), 256, 0, s>>>(reinterpret_cast<const float4*>(out), out16, (uint32_t)n_pix);
        PT_HIP(ctx, hipGetLastError());
    }
    PT_HIP(ctx, hipMemcpyAsync(sl.h_counters, sl.counters.p, sizeof(DCounters), hipMemcpyDeviceToHost, s));
    PT_HIP(ctx, hipEventRecord(sl.ev1, s));
    sl.pending = true;
    sl.obj_set = ctx->cur_set;
    sl.accel_used = (os.use_tlas ? 1u : 0u) | (ctx->n_blas_nodes ? 2u : 0u) | (os.refitted ? 4u : 0u);
    L.frame_seq++;
    L.last_frame_stream = s;
    return PT_OK;
}

}  // namespace

extern "C" {

int pt_render_device(pt_ctx* ctx, const PtFrame* frame, void* out_rgba_device, void* hip_stream) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    try {
        if (!frame || !out_rgba_device) return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_render_device: null argument");
        PT_HIP(ctx, hipSetDevice(ctx->device));
        Lane& L = ctx->lanes[0];
        hipStream_t s = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->stream;
        if (frame->flags & PT_FRAME_OUTPUT_F16) {
            const size_t n_pix = (size_t)frame->width * frame->height;
            if (L.d_out.cap < 16 * n_pix) { int rc = finish_lane(ctx, L); if (rc != PT_OK) return rc; }   // (a frame in flight may still write the old one)
            PT_HIP(ctx, L.d_out.ensure(16 * n_pix));
            int rc = render_enqueue(ctx, L, frame, L.d_out.as<float>(), s, reinterpret_cast<uint2*>(out_rgba_device));
            if (rc != PT_OK) return rc;
        } else {
            int rc = render_enqueue(ctx, L, frame, reinterpret_cast<float*>(out_rgba_device), s);
            if (rc != PT_OK) return rc;
        }
        ctx->cur_lane = 0;
        if (!hip_stream) PT_HIP(ctx, hipStreamSynchronize(s));
        return PT_OK;
    } catch (...) {
        return fail(ctx, PT_ERR_DEVICE, "pt_render_device: unexpected exception");
    }
}

int pt_render_device_noalloc(pt_ctx* ctx, const PtFrame* frame, void* out_rgba_device, void* hip_stream) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    try {
        if (!frame || !out_rgba_device) return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_render_device_noalloc: null argument");
        PT_HIP(ctx, hipSetDevice(ctx->device));
        hipStream_t s = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->stream;
        int rc = render_enqueue(ctx, ctx->lanes[0], frame, reinterpret_cast<float*>(out_rgba_device), s, nullptr, true);
        if (rc != PT_OK) return rc;
        ctx->cur_lane = 0;
        if (!hip_stream) PT_HIP(ctx, hipStreamSynchronize(s));
        return PT_OK;
    } catch (...) {
        return fail(ctx, PT_ERR_DEVICE, "pt_render_device_noalloc: unexpected exception");
    }
}

int pt_render(pt_ctx* ctx, const PtFrame* frame, void* out_rgba_host) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    try {
        if (!frame || !out_rgba_host) return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_render: null argument");
        PT_HIP(ctx, hipSetDevice(ctx->device));
        Lane& L = ctx->lanes[0];
        const size_t n_pix = (size_t)frame->width * frame->height;
        const bool f16 = (frame->flags & PT_FRAME_OUTPUT_F16) != 0;
        const size_t out_bytes = (f16 ? 8 : 16) * n_pix;
        if (L.d_out.cap < 16 * n_pix || (f16 && L.d_out16.cap < 8 * n_pix)) { int rc = finish_lane(ctx, L); if (rc != PT_OK) return rc; }
        PT_HIP(ctx, L.d_out.ensure(16 * n_pix));
        if (f16) PT_HIP(ctx, L.d_out16.ensure(8 * n_pix));
        int rc = render_enqueue(ctx, L, frame, L.d_out.as<float>(), ctx->stream, f16 ? L.d_out16.as<uint2>() : nullptr);
        if (rc != PT_OK) return rc;
        ctx->cur_lane = 0;
        const void* src = f16 ? L.d_out16.p : L.d_out.p;
        PT_HIP(ctx, hipMemcpyAsync(out_rgba_host, src, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
        PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return PT_OK;
    } catch (...) {
        return fail(ctx, PT_ERR_DEVICE, "pt_render: unexpected exception");
    }
}

}  // extern "C"

namespace {

// pt_render_submit / pt_render_submit_device: the frame goes onto the stream of lane 1 + slot; `host` or `device` (one of
// them) receives the image behind it on that stream.
int submit_frame(pt_ctx* ctx, const PtFrame* frame, uint32_t slot, void* host, void* device, const char* who) {
    if (!frame) return fail(ctx, PT_ERR_INVALID_ARGUMENT, std::string(who) + ": null argument");
    if (slot >= kMaxFrameSlots) return fail(ctx, PT_ERR_INVALID_ARGUMENT, std::string(who) + ": slot >= PT_MAX_FRAME_SLOTS");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    Lane& L = ctx->lanes[1u + slot];
    if (L.submitted) return fail(ctx, PT_ERR_INVALID_ARGUMENT, std::string(who) + ": slot " + std::to_string(slot) + " holds a frame that has not been waited for (pt_render_wait)");
    if (!L.ready) {
        const hipError_t e = lane_init(ctx, L, true);
        if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? PT_ERR_OUT_OF_MEMORY : PT_ERR_DEVICE, std::string(who) + ": " + hipGetErrorString(e));
    }
    const size_t n_pix = (size_t)frame->width * frame->height;
    const bool f16 = (frame->flags & PT_FRAME_OUTPUT_F16) != 0;
    // (the slot's previous frame has been waited for, so its image buffers are free to be re-allocated)
    float* out32 = static_cast<float*>(device);
    uint2* out16 = nullptr;
    if (!device || f16) {
        PT_HIP(ctx, L.d_out.ensure(16 * n_pix));
        out32 = L.d_out.as<float>();
    }
    if (f16) {
        if (device) out16 = static_cast<uint2*>(device);
        else { PT_HIP(ctx, L.d_out16.ensure(8 * n_pix)); out16 = L.d_out16.as<uint2>(); }
    }
    int rc = render_enqueue(ctx, L, frame, out32, L.stream, out16);
    if (rc != PT_OK) return rc;
    if (host) PT_HIP(ctx, hipMemcpyAsync(host, f16 ? static_cast<const void*>(out16) : static_cast<const void*>(out32), (f16 ? 8 : 16) * n_pix, hipMemcpyDeviceToHost, L.stream));
    L.submitted = true;   // (from here on the slot has to be waited for, whatever the record below returns)
    PT_HIP(ctx, hipEventRecord(L.ev_done, L.stream));
    // Hand everything enqueued so far to the device now: the runtime may hold the tail of a stream's commands (the copy of
    // the counters, the event markers) back until the stream is next touched, and with several slots in use a frame was
    // seen to sit finished on the device while pt_render_wait for it did not return until another stream was queried.
    (void)hipStreamQuery(L.stream);
    return PT_OK;
}

}  // namespace

extern "C" {

int pt_render_submit(pt_ctx* ctx, const PtFrame* frame, uint32_t slot, void* out_rgba_host) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    try {
        if (!out_rgba_host) return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_render_submit: null argument");
        return submit_frame(ctx, frame, slot, out_rgba_host, nullptr, "pt_render_submit");
    } catch (...) {
        return fail(ctx, PT_ERR_DEVICE, "pt_render_submit: unexpected exception");
    }
}

int pt_render_submit_device(pt_ctx* ctx, const PtFrame* frame, uint32_t slot, void* out_rgba_device) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    try {
        if (!out_rgba_device) return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_render_submit_device: null argument");
        return submit_frame(ctx, frame, slot, nullptr, out_rgba_device, "pt_render_submit_device");
    } catch (...) {
        return fail(ctx, PT_ERR_DEVICE, "pt_render_submit_device: unexpected exception");
    }
}

int pt_render_wait(pt_ctx* ctx, uint32_t slot) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    try {
        if (slot >= kMaxFrameSlots) return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_render_wait: slot >= PT_MAX_FRAME_SLOTS");
        Lane& L = ctx->lanes[1u + slot];
        if (!L.submitted) return fail(ctx, PT_ERR_NOT_READY, "pt_render_wait: no frame has been submitted to slot " + std::to_string(slot));
        PT_HIP(ctx, hipSetDevice(ctx->device));
        L.submitted = false;   // (whatever happens below, the slot can be submitted to again)
        for (Lane& other : ctx->lanes)      // (see submit_frame: nothing of any slot is to be left with the runtime)
            if (other.ready && other.stream && &other != &L) (void)hipStreamQuery(other.stream);
        PT_HIP(ctx, hipStreamSynchronize(L.stream));
        ctx->cur_lane = 1u + slot;
        return finish_lane(ctx, L);
    } catch (...) {
        return fail(ctx, PT_ERR_DEVICE, "pt_render_wait: unexpected exception");
    }
}

int pt_render_stream_wait(pt_ctx* ctx, uint32_t slot, void* hip_stream) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    try {
        if (slot >= kMaxFrameSlots) return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_render_stream_wait: slot >= PT_MAX_FRAME_SLOTS");
        Lane& L = ctx->lanes[1u + slot];
        if (!L.submitted) return fail(ctx, PT_ERR_NOT_READY, "pt_render_stream_wait: no frame has been submitted to slot " + std::to_string(slot));
        PT_HIP(ctx, hipSetDevice(ctx->device));
        PT_HIP(ctx, hipStreamWaitEvent(hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->stream, L.ev_done, 0));
        return PT_OK;
    } catch (...) {
        return fail(ctx, PT_ERR_DEVICE, "pt_render_stream_wait: unexpected exception");
    }
}

int pt_render_slot_wait_for(pt_ctx* ctx, uint32_t slot, void* hip_stream) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    try {
        if (slot >= kMaxFrameSlots) return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_render_slot_wait_for: slot >= PT_MAX_FRAME_SLOTS");
        PT_HIP(ctx, hipSetDevice(ctx->device));
        Lane& L = ctx->lanes[1u + slot];
        if (!L.ready) {
            const hipError_t e = lane_init(ctx, L, true);
            if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? PT_ERR_OUT_OF_MEMORY : PT_ERR_DEVICE, std::string("pt_render_slot_wait_for: ") + hipGetErrorString(e));
        }
        PT_HIP(ctx, hipEventRecord(L.ev_dep, hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->stream));
        PT_HIP(ctx, hipStreamWaitEvent(L.stream, L.ev_dep, 0));
        return PT_OK;
    } catch (...) {
        return fail(ctx, PT_ERR_DEVICE, "pt_render_slot_wait_for: unexpected exception");
    }
}

int pt_host_alloc(pt_ctx* ctx, size_t bytes, void** out_ptr) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    try {
        if (!out_ptr || bytes == 0) return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_host_alloc: null pointer or zero size");
        *out_ptr = nullptr;
        PT_HIP(ctx, hipSetDevice(ctx->device));
        void* p = nullptr;
        PT_HIP(ctx, hipHostMalloc(&p, bytes, hipHostMallocDefault));
        ctx->host_allocs.push_back(p);
        *out_ptr = p;
        return PT_OK;
    } catch (...) {
        return fail(ctx, PT_ERR_OUT_OF_MEMORY, "pt_host_alloc: host allocation failed");
    }
}

int pt_host_free(pt_ctx* ctx, void* ptr) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    if (!ptr) return PT_OK;
    auto it = std::find(ctx->host_allocs.begin(), ctx->host_allocs.end(), ptr);
    if (it == ctx->host_allocs.end()) return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_host_free: not a pointer from pt_host_alloc of this context");
    ctx->host_allocs.erase(it);
    (void)hipSetDevice(ctx->device);
    PT_HIP(ctx, hipHostFree(ptr));
    return PT_OK;
}

int pt_query(pt_ctx* ctx, const PtRay* rays_host, uint32_t n_rays, PtHit* hits_host) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    try {
        if (n_rays && (!rays_host || !hits_host)) return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_query: null argument");
        int rc = check_ready(ctx, "pt_query");
        if (rc != PT_OK) return rc;
        if (n_rays == 0) return PT_OK;
        PT_HIP(ctx, hipSetDevice(ctx->device));
        hipStream_t s = ctx->stream;
        PT_HIP(ctx, ctx->d_rays.ensure(sizeof(PtRay) * (size_t)n_rays));
        PT_HIP(ctx, ctx->d_hits.ensure(sizeof(PtHit) * (size_t)n_rays));
        PT_HIP(ctx, hipMemcpyAsync(ctx->d_rays.p, rays_host, sizeof(PtRay) * (size_t)n_rays, hipMemcpyHostToDevice, s));
        SceneView sv = make_view(ctx);
        const uint32_t depth = stack_depth_needed(ctx);
        // the same organisation of the scene as the frame kernels use (small scenes: every record staged in LDS)
        LdsScene ls;
        const int mode = scene_mode(ctx, &ls);
        const size_t lds = (size_t)4 * depth * 64 * sizeof(uint32_t) + (mode == PT_MODE_LDS ? ls.total_bytes : 0u);
        const uint32_t grid = div_up(n_rays, 256);
        if (mode == PT_MODE_TLAS) k_query<PT_MODE_TLAS><<<grid, 256, lds, s>>>(PT_SCENE_ARGS(sv), ls, depth, ctx->d_rays.as<DRay>(), n_rays, ctx->d_hits.as<DHit>());
        else if (mode == PT_MODE_LDS) k_query<PT_MODE_LDS><<<grid, 256, lds, s>>>(PT_SCENE_ARGS(sv), ls, depth, ctx->d_rays.as<DRay>(), n_rays, ctx->d_hits.as<DHit>());
        else k_query<PT_MODE_LIST><<<grid, 256, lds, s>>>(PT_SCENE_ARGS(sv), ls, depth, ctx->d_rays.as<DRay>(), n_rays, ctx->d_hits.as<DHit>());
        PT_HIP(ctx, hipGetLastError());
        PT_HIP(ctx, hipMemcpyAsync(hits_host, ctx->d_hits.p, sizeof(PtHit) * (size_t)n_rays, hipMemcpyDeviceToHost, s));
        PT_HIP(ctx, hipStreamSynchronize(s));
        return PT_OK;
    } catch (...) {
        return fail(ctx, PT_ERR_DEVICE, "pt_query: unexpected exception");
    }
}

int pt_read_aux(pt_ctx* ctx, uint32_t kind, void* out_host, size_t out_bytes) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    try {
        if (!out_host) return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_read_aux: null argument");
        Lane& L = ctx->lanes[ctx->cur_lane];
        if (L.aux_w == 0) return fail(ctx, PT_ERR_NOT_READY, "pt_read_aux: no frame has been rendered");
        const size_t need = (size_t)4 * L.aux_w * L.aux_h;
        if (out_bytes < need) return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_read_aux: buffer too small");
        const DevBuf* src = nullptr;
        switch (kind) {
            case PT_AUX_PRIMARY_OBJECT: src = &L.d_aux_obj; break;
            case PT_AUX_PRIMARY_FLAT_ID: src = &L.d_aux_flat; break;
            case PT_AUX_PRIMARY_T: src = &L.d_aux_t; break;
            case PT_AUX_RAY_COUNT: src = &L.d_aux_rays; break;
            default: return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_read_aux: unknown kind");
        }
        PT_HIP(ctx, hipSetDevice(ctx->device));
        int rc = finish_lane(ctx, L);
        if (rc != PT_OK) return rc;
        PT_HIP(ctx, hipMemcpy(out_host, src->p, need, hipMemcpyDeviceToHost));
        return PT_OK;
    } catch (...) {
        return fail(ctx, PT_ERR_DEVICE, "pt_read_aux: unexpected exception");
    }
}

int pt_debug_wave_log(pt_ctx* ctx, void* out_host, size_t out_bytes, uint32_t* n_records) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    try {
        if (!n_records) return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_debug_wave_log: null argument");
        PT_HIP(ctx, hipSetDevice(ctx->device));
        Lane& L = ctx->lanes[ctx->cur_lane];
        int rc = finish_lane(ctx, L);
        if (rc != PT_OK) return rc;
        *n_records = L.wave_log_records;
        const size_t bytes = std::min(out_bytes, (size_t)64 * L.wave_log_records);
        if (out_host && bytes) PT_HIP(ctx, hipMemcpy(out_host, L.d_wave_log.p, bytes, hipMemcpyDeviceToHost));
        return PT_OK;
    } catch (...) {
        return fail(ctx, PT_ERR_DEVICE, "pt_debug_wave_log: unexpected exception");
    }
}

int pt_debug_slot_state(pt_ctx* ctx, uint32_t slot, uint32_t* out, uint32_t n_out) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    try {
        if (!out || n_out < 20u || slot >= kMaxFrameSlots) return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_debug_slot_state: bad argument");
        PT_HIP(ctx, hipSetDevice(ctx->device));
        Lane& L = ctx->lanes[1u + slot];
        std::memset(out, 0, sizeof(uint32_t) * n_out);
        if (!ctx->debug_stream) PT_HIP(ctx, hipStreamCreateWithFlags(&ctx->debug_stream, hipStreamNonBlocking));
        out[0] = L.submitted ? 1u : 0u;
        out[1] = L.ready ? (uint32_t)hipStreamQuery(L.stream) : 0xFFFFFFFFu;
        out[2] = (uint32_t)L.frame_seq;
        out[3] = L.last_schedule;
        if (!L.ready) return PT_OK;
        for (int k = 0; k < 2; ++k) {
            FrameSlot& sl = L.slot[k];
            uint32_t* o = out + 4 + 8 * k;
            o[0] = sl.pending ? 1u : 0u;
            o[1] = (uint32_t)hipEventQuery(sl.ev1);
            DCounters c;
            if (sl.counters.p && hipMemcpyAsync(&c, sl.counters.p, sizeof c, hipMemcpyDeviceToHost, ctx->debug_stream) == hipSuccess &&
                hipStreamSynchronize(ctx->debug_stream) == hipSuccess) {
                o[2] = c.n_paths; o[3] = c.next_path; o[4] = c.waves_done; o[5] = (uint32_t)c.rays;
                o[6] = c.others_running; o[7] = c.n_split;
            }
        }
        (void)hipGetLastError();
        return PT_OK;
    } catch (...) {
        return fail(ctx, PT_ERR_DEVICE, "pt_debug_slot_state: unexpected exception");
    }
}

int pt_set_stats_detail(pt_ctx* ctx, int detail) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    ctx->stats_detail = detail != 0;
    return PT_OK;
}

int pt_get_stats(pt_ctx* ctx, PtStats* out) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    try {
        if (!out) return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_get_stats: null argument");
        PT_HIP(ctx, hipSetDevice(ctx->device));
        // every frame in flight is waited for (this is also how a caller of pt_render_device learns that its frames are done);
        // the numbers are those of the lane whose frame was completed last through pt_render* / pt_render_wait
        int rc = finish_all(ctx);
        if (rc != PT_OK) return rc;
        *out = ctx->lanes[ctx->cur_lane].stats;
        out->build_ms = ctx->build_ms;
        return PT_OK;
    } catch (...) {
        return fail(ctx, PT_ERR_DEVICE, "pt_get_stats: unexpected exception");
    }
}

int pt_debug_check_accel(pt_ctx* ctx, uint32_t* n_errors, uint32_t* tlas_depth, uint32_t* max_blas_depth) {
    if (!ctx) return PT_ERR_INVALID_ARGUMENT;
    try {
        if (!n_errors) return fail(ctx, PT_ERR_INVALID_ARGUMENT, "pt_debug_check_accel: null argument");
        PT_HIP(ctx, hipSetDevice(ctx->device));
        hipStream_t s = ctx->stream;
        // t_errors: [0] violations, [1] primitives found in the leaves of the wide trees
        PT_HIP(ctx, ctx->t_errors.ensure(2 * sizeof(uint32_t)));
        PT_HIP(ctx, hipMemsetAsync(ctx->t_errors.p, 0, 2 * sizeof(uint32_t), s));
        uint64_t want_leaves = 0;
        for (uint32_t m = 0; m < ctx->n_meshes; ++m) {
            if (ctx->h_blas_nodes_base[m] == PT_NONE) continue;
            const uint32_t base = ctx->h_blas_nodes_base[m], first = ctx->h_meshes[m].tri_first, ni = ctx->h_meshes[m].tri_count - 1;
            const BvhNode* bn = ctx->d_blas_bnodes.as<BvhNode>() + base;
            const uint32_t* order = ctx->d_blas_order.as<uint32_t>() + first;
            k_check_bvh<<<div_up(ni, 256), 256, 0, s>>>(bn, base, ni, ctx->d_tri_lo.as<float4>(), ctx->d_tri_hi.as<float4>(), first, order, ctx->t_errors.as<uint32_t>());
            if (const uint32_t nw = ctx->h_blas_n_wide[m]) {
                k_check_wide<<<div_up(nw, 256), 256, 0, s>>>(ctx->d_blas_nodes.as<WNode>() + base, base, nw, bn, ctx->d_tri_lo.as<float4>(),
                                                             ctx->d_tri_hi.as<float4>(), first, order, ctx->t_errors.as<uint32_t>());
                want_leaves += ctx->h_meshes[m].tri_count;
            }
        }
        const ObjectSet& os = ctx->sets[ctx->cur_set];
        if (os.built && os.use_tlas && os.n_objects > 1) {
            k_check_bvh<<<div_up(os.n_objects - 1, 256), 256, 0, s>>>(ctx->d_tlas_bnodes.as<BvhNode>(), 0, os.n_objects - 1,
                                                                      os.d_obj_lo.as<float4>(), os.d_obj_hi.as<float4>(), 0, nullptr,
                                                                      ctx->t_errors.as<uint32_t>());
            if (os.tlas_n_wide) {
                k_check_wide<<<div_up(os.tlas_n_wide, 256), 256, 0, s>>>(os.d_tlas_nodes.as<WNode>(), 0, os.tlas_n_wide, ctx->d_tlas_bnodes.as<BvhNode>(),
                                                                         os.d_obj_lo.as<float4>(), os.d_obj_hi.as<float4>(), 0, nullptr,
                                                                         ctx->t_errors.as<uint32_t>());
                want_leaves += os.n_objects;
            }
        }
        PT_HIP(ctx, hipGetLastError());
        uint32_t found[2] = {0, 0};
        PT_HIP(ctx, hipMemcpyAsync(found, ctx->t_errors.p, sizeof found, hipMemcpyDeviceToHost, s));
        PT_HIP(ctx, hipStreamSynchronize(s));
        *n_errors = found[0] + (found[1] == want_leaves ? 0u : 1u);   // every primitive in exactly one leaf
        if (tlas_depth) *tlas_depth = os.use_tlas ? os.tlas_depth : 0;
        if (max_blas_depth) *max_blas_depth = ctx->max_blas_depth;
        return PT_OK;
    } catch (...) {
        return fail(ctx, PT_ERR_DEVICE, "pt_debug_check_accel: unexpected exception");
    }
}

}  // extern "C"

#include "pt_multi.h"

// pt_scene.h — device-side scene layout (HBM resident, rebuilt from the wire
// structs of include/pt_amd.h at upload time).
#pragma once

#include "pt_math.h"

// One instance, the part every ray-object test reads. 64 B = 4 x dwordx4 (scalar loads when the object index is
// wave-uniform). local_to_world, needed once per ray for the winner only, is kept in an array of its own (obj_l2w), so
// that the records a traversal touches take half the cache.
struct DObject {
    Affine w2l;          // world_to_local rows (reference src/query.wgsl:73-74)
    uint32_t mesh;
    uint32_t tri_first;  // == meshes[mesh].tri_first, so that the hit record needs no trip through the mesh record
    uint32_t mat;
    uint32_t tri_base;   // flat triangle id of this object's triangle 0 (order of the reference's double loop)
};
static_assert(sizeof(DObject) == 64, "DObject");
static_assert(sizeof(Affine) == 48, "Affine");   // obj_l2w[]: local_to_world rows (reference src/query.wgsl:118-119)

// One mesh. 48 B.
struct DMesh {
    float bmin[3];
    uint32_t tri_first;   // first record of this mesh in tris[] / tri_attrs[]
    float bmax[3];
    uint32_t tri_count;
    uint32_t blas_root;   // node index into blas_nodes[], a leaf reference (WNode::child), or PT_NONE (= loop over all triangles)
    uint32_t blas_need;   // traversal stack entries a walk of this BLAS can need
    float max_abs;        // largest |coordinate| of the mesh AABB
    uint32_t _pad;
};
static_assert(sizeof(DMesh) == 48, "DMesh");

// Geometry of one triangle, precomputed with the arithmetic of reference src/query.wgsl:95-97:
//   a = va.position, e_ab = vb.position - va.position, e_ac = vc.position - va.position, n = cross(e_ab, e_ac)
// 48 B = 3 x dwordx4.
struct DTri {
    float4 q0;  // a.x a.y a.z e_ab.x
    float4 q1;  // e_ab.y e_ab.z e_ac.x e_ac.y
    float4 q2;  // e_ac.z n.x n.y n.z
};
static_assert(sizeof(DTri) == 48, "DTri");

// Shading attributes of one triangle, fetched once per ray for the winning triangle. 64 B.
struct DTriAttr {
    float4 q0;  // na.x na.y na.z nb.x
    float4 q1;  // nb.y nb.z nc.x nc.y
    float4 q2;  // nc.z uva.x uva.y uvb.x
    float4 q3;  // uvb.y uvc.x uvc.y 0
};
static_assert(sizeof(DTriAttr) == 64, "DTriAttr");

// Binary BVH node holding the boxes of both children. 64 B = 4 x dwordx4. What the LBVH build produces; the rays walk
// the 4-wide trees (WNode) made from these.
// A child reference with PT_LEAF_BIT set is a primitive (position in the sorted order for a BLAS, object for the TLAS).
#define PT_LEAF_BIT 0x80000000u
struct BvhNode {
    float4 lo_l;  // left  min.xyz, w = bits of the left child reference
    float4 hi_l;  // left  max.xyz, w = bits of the left subtree's depth (build bookkeeping)
    float4 lo_r;  // right min.xyz, w = bits of the right child reference
    float4 hi_r;  // right max.xyz, w = bits of the right subtree's depth
};
static_assert(sizeof(BvhNode) == 64, "BvhNode");

// Node of the 4-wide trees the rays walk (TLAS over objects, BLAS over a mesh's triangles). 64 B = 4 x dwordx4.
// The boxes of the (up to) four children are stored as 8-bit offsets on a per-node grid: child k covers, on axis a,
//   [ origin[a] + qlo[a].byte(k) * 2^(e[a] - 127),  origin[a] + qhi[a].byte(k) * 2^(e[a] - 127) ]
// which encloses the child's real box (lower corner rounded down, upper corner rounded up, checked at build time with
// the very expression above). A slot without a child holds PT_NONE and an inverted box.
// Child references: an index into the node array (a TLAS or BLAS of its own numbering), or, with PT_LEAF_BIT set, a leaf:
//   TLAS: PT_LEAF_BIT | object index
//   BLAS: PT_LEAF_BIT | (count - 1) << 29 | position: `count` (1..4) triangles from `position` on in the mesh's
//         triangle records, which are stored in the order of the tree's leaves (tri_ids[] gives their mesh numbering).
// Nodes are numbered breadth first: the top levels of a tree are a prefix of its array.
#define PT_WLEAF_COUNT_SHIFT 29
#define PT_WLEAF_POS_MASK 0x1FFFFFFFu
#define PT_WIDE_MAX_LEAF 4u
struct WNode {
    float origin[3];
    uint32_t exps;       // e[0] | e[1] << 8 | e[2] << 16 (biased exponents of the grid steps)
    uint32_t child[4];
    uint32_t qlo[3];     // per axis: byte k = lower corner of child k
    uint32_t qhi[3];     // per axis: byte k = upper corner of child k
    uint32_t _pad[2];
};
static_assert(sizeof(WNode) == 64, "WNode");

struct DMaterial {  // == PtMaterial (reference src/raytrace.wgsl:29-39)
    float albedo[3];
    uint32_t albedo_texture;
    float emissive[3];
    uint32_t emissive_texture;
    float roughness;
    float metallic;
    uint32_t metallic_roughness_texture;
    float reflectance;
    uint32_t normal_map_texture;
    uint32_t _pad[3];
};
static_assert(sizeof(DMaterial) == 64, "DMaterial");

struct DTexture {  // == PtTexture (reference src/raytrace.wgsl:41-46)
    uint32_t width, height, offset, format;
};

// Everything a trace kernel reads, passed by value as one kernel argument.
struct SceneView {
    const DObject* __restrict__ objects;
    const Affine* __restrict__ obj_l2w;
    const DMesh* __restrict__ meshes;
    const DTri* __restrict__ tris;         // meshes with a BLAS: in the order of the tree's leaves
    const uint32_t* __restrict__ tri_ids;  // ... and the triangle's number in its mesh for each of those records
    const DTriAttr* __restrict__ tri_attrs;
    const WNode* __restrict__ blas_nodes;
    const WNode* __restrict__ tlas_nodes;
    const float4* __restrict__ obj_lo;   // padded world box of every object (what the TLAS is built from)
    const float4* __restrict__ obj_hi;
    const DMaterial* __restrict__ materials;
    const DTexture* __restrict__ textures;
    const float* __restrict__ texture_data;
    uint32_t n_objects;
    uint32_t n_texture_data;
    uint32_t tlas_root;   // node index, leaf reference, or PT_NONE when there are no objects
    uint32_t cull_boxes;  // 0: every object is a candidate for every ray (exhaustive loop); 1: cull with the world boxes
    uint32_t _pad0;
    uint32_t tie_lowest;  // 0: equal distances go to the later (object, triangle) as in the reference; 1: to the earlier
    uint32_t _pad1;
    // one mesh's BLAS staged in LDS by the kernel (PT_MODE_LIST_LB), null otherwise
    // PT_MODE_LDS only: the padded world boxes staged in LDS as 12 floats per object, (lo, hi, lo, 0) per axis, so that a
    // lane reads (near, far) of an axis as two consecutive words from an address chosen by the sign of its ray direction
    const float* obj_slab;
    const WNode* lb_nodes;    // its nodes; a node reference r of that mesh is lb_nodes[r - lb_node_base]
    const DTri* lb_tris;      // its triangle records (null: they stay in global memory)
    const uint32_t* lb_tri_ids;
    uint32_t lb_mesh, lb_node_base;
    // PT_MODE_TLAS: the first tlas_top_n nodes of the TLAS (its top levels) staged in LDS by the kernel
    const WNode* tlas_top;
    uint32_t tlas_top_n;
};

// The kernels take the scene arrays as separate `const T* __restrict__` parameters (PT_SCENE_PARAMS) and rebuild a
// SceneView from them: only then does the compiler know that the image stores cannot alias the scene, which lets it
// fetch wave-uniform records (object boxes, mesh and triangle records in the mesh loop) through the scalar cache.
struct SceneScalars {
    uint32_t n_objects, n_texture_data, tlas_root, cull_boxes, tie_lowest;
};
#define PT_SCENE_PARAMS                                                                                                \
    const DObject *__restrict__ p_objects, const Affine *__restrict__ p_l2w, const DMesh *__restrict__ p_meshes,        \
        const DTri *__restrict__ p_tris, const uint32_t *__restrict__ p_tri_ids, const DTriAttr *__restrict__ p_attrs,  \
        const WNode *__restrict__ p_blas, const WNode *__restrict__ p_tlas,                                             \
        const float4 *__restrict__ p_obj_lo, const float4 *__restrict__ p_obj_hi,                                       \
        const DMaterial *__restrict__ p_materials, const DTexture *__restrict__ p_textures,                             \
        const float *__restrict__ p_texdata, SceneScalars ss
#define PT_SCENE_ARGS(v)                                                                                               \
    (v).objects, (v).obj_l2w, (v).meshes, (v).tris, (v).tri_ids, (v).tri_attrs, (v).blas_nodes, (v).tlas_nodes,         \
        (v).obj_lo, (v).obj_hi, (v).materials, (v).textures, (v).texture_data,                                          \
        SceneScalars { (v).n_objects, (v).n_texture_data, (v).tlas_root, (v).cull_boxes, (v).tie_lowest }
#define PT_SCENE_VIEW_FROM_PARAMS()                                                                                    \
    SceneView sc;                                                                                                       \
    sc.objects = p_objects; sc.obj_l2w = p_l2w; sc.meshes = p_meshes; sc.tris = p_tris; sc.tri_ids = p_tri_ids;         \
    sc.tri_attrs = p_attrs; sc.blas_nodes = p_blas;                                                                     \
    sc.tlas_nodes = p_tlas; sc.obj_lo = p_obj_lo; sc.obj_hi = p_obj_hi; sc.materials = p_materials;                     \
    sc.textures = p_textures; sc.texture_data = p_texdata; sc.n_objects = ss.n_objects;                                 \
    sc.n_texture_data = ss.n_texture_data; sc.tlas_root = ss.tlas_root; sc.cull_boxes = ss.cull_boxes;                  \
    sc._pad0 = 0; sc.tie_lowest = ss.tie_lowest; sc._pad1 = 0;                                                       \
    sc.obj_slab = nullptr; sc.lb_nodes = nullptr; sc.lb_tris = nullptr; sc.lb_tri_ids = nullptr;                        \
    sc.lb_mesh = 0xFFFFFFFFu; sc.lb_node_base = 0; sc.tlas_top = nullptr; sc.tlas_top_n = 0;

// Per-frame constants.
struct FrameParams {
    float wfv[16];   // view.world_from_view, column-major
    float vfc[16];   // view.view_from_clip, column-major
    uint32_t width, height;
    float time;
    uint32_t bounces, samples;
    float sky[3];
    uint32_t part_index, part_count;
    uint32_t tiles_x, tiles_y;
    uint32_t n_local_tiles;
};

// Work counters accumulated by the trace kernels (one atomic per wave and counter).
struct DCounters {
    unsigned long long rays;
    unsigned long long xforms;
    unsigned long long box_tests;
    unsigned long long tri_tests;
    unsigned long long shades;
    unsigned long long hits;
    unsigned int n_paths;      // entries k_primary appended to the path list
    unsigned int next_path;    // work queue head of k_paths: next unassigned list entry
    unsigned long long passes;   // object-test passes of k_paths waves (STATS only)
    // phase clocks of the frame kernel (shader cycles summed over waves), STATS variant only
    unsigned long long cyc_refill, cyc_trace, cyc_shade, rounds, lane_rounds, wave_life_sum, wave_life_max, waves;
    unsigned int life_hist[32];   // histogram of k_paths wave lifetimes, 2^19 cycles per bucket (STATS only)
    unsigned int waves_done;      // k_paths waves that have left (kept only for frames ordered by hints; the last one signs the frame off)
    unsigned int n_groups, others_running, waves_needed, n_split;   // what k_chunk_order decided and saw (for logs)
};

/*
 * pt_amd.h — C ABI of the MI355X (gfx950) path-tracing backend.
 *
 * This library replaces the GPU pass of the Bevy plugin PixelDoted/path-tracing:
 * the render pass that `RayTraceNode::run` records (reference src/shader.rs:121-213)
 * and the shader it runs (src/raytrace.wgsl, src/query.wgsl, src/math.wgsl).
 * Everything on the Rust side above that call (src/lib.rs, src/extract.rs,
 * src/data.rs) keeps producing the same buffers; they are handed to the functions
 * below instead of `StorageBuffer::write_buffer`.
 *
 * Conventions
 *  - Every function returns 0 (PT_OK) or a negative PT_ERR_* code and never
 *    throws or aborts across the boundary. `pt_last_error(ctx)` returns a
 *    message owned by the context (valid until the next call on that context).
 *  - Host pointers are borrowed for the duration of the call only.
 *  - Matrices are column-major float[16] (glam `Mat4`, WGSL `mat4x4<f32>`).
 *  - One caller thread per context. Calls on one context are serialised by the
 *    caller; several contexts (one per GPU) may be driven from several threads.
 *  - The struct layouts are the encase/std430 layouts of reference
 *    src/data.rs:15-73 and are checked by static assertions.
 */
#ifndef PT_AMD_H
#define PT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_ABI_VERSION 1

/* ---- error codes --------------------------------------------------------- */
enum {
    PT_OK = 0,
    PT_ERR_INVALID_ARGUMENT = -1, /* null pointer, bad size, bad enum            */
    PT_ERR_INVALID_SCENE = -2,    /* index out of range, non-finite geometry     */
    PT_ERR_NOT_READY = -3,        /* render/query before the scene was uploaded  */
    PT_ERR_DEVICE = -4,           /* HIP runtime error (message has the detail)  */
    PT_ERR_OUT_OF_MEMORY = -5,
    PT_ERR_UNSUPPORTED = -6,
    PT_ERR_NO_DEVICE = -7         /* no gfx950 device / device index out of range */
};

/* "no texture" / "miss" marker (reference src/math.wgsl:5, src/extract.rs:235) */
#define PT_U32_MAX 0xFFFFFFFFu

/* ---- wire structs: byte-identical to the reference's storage buffers ------ */

/* reference src/data.rs:23-30 == src/query.wgsl:16-22; stride 144 */
typedef struct PtObject {
    float local_to_world[16];
    float world_to_local[16];
    uint32_t mat;
    uint32_t mesh;
    uint32_t _pad[2];
} PtObject;

/* reference src/data.rs:58-66 == src/query.wgsl:24-31; stride 48 */
typedef struct PtMesh {
    float aabb_min[3];
    uint32_t _pad0;
    float aabb_max[3];
    uint32_t ihead;     /* first element of this mesh in indices[]            */
    uint32_t vhead;     /* first element of this mesh in vertices[]           */
    uint32_t tri_count;
    uint32_t _pad1[2];
} PtMesh;

/* reference src/data.rs:68-73 == src/query.wgsl:33-37; stride 48 */
typedef struct PtVertex {
    float position[3];
    uint32_t _pad0;
    float normal[3];
    uint32_t _pad1;
    float uv[2];
    uint32_t _pad2[2];
} PtVertex;

/* reference src/data.rs:32-43 == src/raytrace.wgsl:29-39; stride 64 */
typedef struct PtMaterial {
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
} PtMaterial;

/* reference src/data.rs:45-51 == src/raytrace.wgsl:41-46; stride 16.
 * `offset` counts f32 elements of texture_data[]; `format` = channels/texel. */
typedef struct PtTexture {
    uint32_t width;
    uint32_t height;
    uint32_t offset;
    uint32_t format;
} PtTexture;

/* reference src/data.rs:15-20 == src/raytrace.wgsl:23-27; size 32 */
typedef struct PtSettings {
    uint32_t bounces;
    uint32_t samples;
    uint32_t _pad[2];
    float sky_color[4]; /* LinearRgba on the Rust side; the path reads xyz      */
} PtSettings;

/* ---- per-frame parameters ------------------------------------------------- */

/* pt_render flags */
#define PT_FRAME_ACCUMULATE 0x1u /* add this frame into the context's radiance
                                    accumulation buffer and return the mean   */
#define PT_FRAME_RESET      0x2u /* clear the accumulation buffer first        */
#define PT_FRAME_OUTPUT_F16 0x4u /* the image is written as RGBA binary16 (8
                                    bytes per pixel, round to nearest even), the
                                    format of the reference's render target
                                    (src/shader.rs:321), instead of RGBA f32    */

/*
 * The subset of Bevy's `View` and `Globals` uniforms the shader reads
 * (reference src/raytrace.wgsl:210-217) plus the `Settings` uniform (:13).
 * `width`/`height` are `view.viewport.zw`.
 *
 * Pixel partition (for multi-GPU tile splitting): the image is cut into 8x8
 * pixel tiles numbered row-major; this call renders tile k iff
 * k % part_count == part_index. part_count == 0 or 1 renders everything.
 * Pixels outside the partition are written as zeros.
 */
typedef struct PtFrame {
    float world_from_view[16];
    float view_from_clip[16];
    uint32_t width;
    uint32_t height;
    float time;          /* Globals.time, seconds                             */
    uint32_t flags;      /* PT_FRAME_*                                         */
    PtSettings settings;
    uint32_t part_index;
    uint32_t part_count;
    uint32_t _reserved[2];
} PtFrame;

/* ---- context --------------------------------------------------------------- */

/* closest-hit strategy */
enum {
    PT_ACCEL_AUTO = 0,  /* two-level LBVH where it pays, uniform loops otherwise */
    PT_ACCEL_BRUTE = 1, /* the reference's loop nest (src/query.wgsl:55-125)     */
    PT_ACCEL_BVH = 2    /* force TLAS + BLAS everywhere                          */
};

/* tie rule for exactly equal hit distances */
enum {
    PT_TIE_REFERENCE = 0, /* highest (object, triangle) wins — what the
                             reference's `t > hit_record.t` reject test does
                             (src/query.wgsl:109, :55-64)                       */
    PT_TIE_LOWEST = 1     /* lowest (object, triangle) wins                     */
};

typedef struct PtConfig {
    uint32_t abi_version; /* PT_ABI_VERSION                                    */
    int32_t device;       /* HIP device ordinal                                */
    uint32_t accel;       /* PT_ACCEL_*                                        */
    uint32_t tie_rule;    /* PT_TIE_*                                          */
    uint32_t flags;       /* reserved, 0                                       */
    uint32_t _reserved[3];
} PtConfig;

typedef struct pt_ctx pt_ctx;

/* Replaces RayTracePlugin::build/finish + RayTracePipeline::from_world
 * (reference src/shader.rs:38-103, :225-339): creates the device state. */
int pt_create(const PtConfig* cfg, pt_ctx** out_ctx);
int pt_destroy(pt_ctx* ctx);

/* Message for the most recent error on this context ("" if none).
 * With ctx == NULL returns the message of the last failed pt_create. */
const char* pt_last_error(const pt_ctx* ctx);

/* ---- scene upload ---------------------------------------------------------- */

/* Replaces the three `write_buffer` calls of prepare_meshes
 * (reference src/extract.rs:131-144). Validates ranges, builds the per-mesh
 * triangle records and BLAS on the device. Zero counts are legal.
 * aabb_min/aabb_max of a mesh must enclose every vertex its triangles use, as
 * the boxes the reference computes do (src/extract.rs:77-93): the library culls
 * by distance with them, the reference only gates on them. A mesh whose box
 * does not enclose its triangles is rejected with PT_ERR_INVALID_SCENE. */
int pt_upload_meshes(pt_ctx* ctx,
                     const PtMesh* meshes, uint32_t n_meshes,
                     const uint32_t* indices, uint32_t n_indices,
                     const PtVertex* vertices, uint32_t n_vertices);

/* Replaces the `write_buffer` calls of extract_materials / extract_textures
 * (reference src/extract.rs:249-253, :284-292). */
int pt_upload_materials(pt_ctx* ctx,
                        const PtMaterial* materials, uint32_t n_materials,
                        const PtTexture* textures, uint32_t n_textures,
                        const float* texture_data, uint32_t n_texture_data);

/* Replaces the per-frame `objects.write_buffer` of extract_visible
 * (reference src/extract.rs:337-342); rebuilds the TLAS, or, when the number of
 * objects is unchanged, refits its boxes (the image does not depend on which).
 * An array identical to the one already uploaded returns at once and leaves
 * frames in flight, the radiance accumulation and the schedule hints alone; any
 * other upload starts a new accumulation (as PT_FRAME_RESET does).
 * The reference's `emissives` buffer is bound but never read by the shader and
 * is not taken. */
int pt_upload_objects(pt_ctx* ctx, const PtObject* objects, uint32_t n_objects);

/* ---- render ---------------------------------------------------------------- */

/* Replaces the body of RayTraceNode::run (reference src/shader.rs:121-213):
 * traces the frame and copies width*height RGBA pixels (row 0 = top, A = 1;
 * f32, or binary16 with PT_FRAME_OUTPUT_F16) to `out_rgba_host`. Synchronous. */
int pt_render(pt_ctx* ctx, const PtFrame* frame, void* out_rgba_host);

/* Same, but the image is written to device memory of the context's device and
 * the call returns after enqueueing on `hip_stream` (a hipStream_t, or NULL for
 * the context's own stream, in which case the call synchronises before
 * returning). Used by callers that keep the image on the GPU (multi-GPU
 * reduce/gather, on-device post-processing). Callers that overlap frames use
 * pt_render_submit below. The library sees on the device whether other frames of
 * the process are running (it keeps two scheduling words per device for that,
 * the only state it has outside its contexts) and schedules a frame for
 * throughput then (only as many workgroups as it has work for, fewer of its
 * pixels handed out in small portions) and for latency when it has the device to
 * itself; the image does not depend on the schedule. */
int pt_render_device(pt_ctx* ctx, const PtFrame* frame, void* out_rgba_device,
                     void* hip_stream);

/* The same frame as pt_render_device without any device-memory allocation, for
 * callers that cannot allocate: code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc. No counterpart in the reference.
 * Everything the kernels need besides the scene, the image and the context (path
 * list, parked pixel state, aux planes, work counters) lives in one fixed array
 * per GPU that is part of the library's code object, sized for frames of at most
 * PT_NOALLOC_MAX_PIXELS pixels, width and height each rounded up to a multiple
 * of 8 (about 29 MiB per GPU, held until the process ends). Larger frames and
 * frames with flags are refused with PT_ERR_UNSUPPORTED. The scene must have
 * been uploaded before (uploads allocate). The image is bit-identical to
 * pt_render_device's. The frame leaves no record in the context: pt_get_stats
 * and pt_read_aux do not describe it. All calls on one GPU, from any context,
 * share the array: calls that overlap in time on one GPU race; calls on one
 * stream, or otherwise ordered, are safe. The call records an event behind the
 * frame; pt_upload_*, pt_get_stats and pt_destroy wait for it, so the scene may be
 * changed or the context destroyed right after the call. Under stream capture
 * (hipStreamBeginCapture on `hip_stream`) nothing is recorded or waited for: no
 * frame of another entry point may be pending on the context when the capture
 * begins (PT_ERR_NOT_READY otherwise), and the scene must stay as it is for as long
 * as the captured graph is replayed. */
#define PT_NOALLOC_MAX_PIXELS 262144u   /* 2^18, e.g. 512 x 512 */
int pt_render_device_noalloc(pt_ctx* ctx, const PtFrame* frame, void* out_rgba_device,
                             void* hip_stream);

/* ---- frames in flight on one context ----------------------------------------- */
/* The asynchronous form of RayTraceNode::run (reference src/shader.rs:121-213; the
 * reference records its pass into a command buffer that Bevy's renderer submits
 * later, which is the same split): pt_render_submit enqueues a frame and returns,
 * pt_render_wait blocks until that frame's image is complete. A context has
 * PT_MAX_FRAME_SLOTS slots; each owns a HIP stream and the work memory of one
 * frame (about 130 bytes per pixel, allocated at the slot's first use), all slots
 * share the one scene and its acceleration structures. A caller that keeps K
 * frames in flight uses slots 0..K-1 in turn:
 *     submit(f0, 0) submit(f1, 1) submit(f2, 2) wait(0) submit(f3, 0) wait(1) ...
 * so that the start of a frame fills the chip while the previous frame's longest
 * pixels finish (on the example scenes three frames in flight give 1.3 to 2.8 times
 * the frame rate of one). Images do not depend on how many frames are in flight.
 *  - A slot holds one frame: a second submit without a wait in between is refused
 *    (PT_ERR_INVALID_ARGUMENT).
 *  - pt_render_submit copies the image to `out_rgba_host` behind the frame (RGBA f32,
 *    or binary16 with PT_FRAME_OUTPUT_F16); the copy overlaps other frames only when
 *    that memory is page-locked (pt_host_alloc). pt_render_submit_device leaves it in
 *    device memory of the context's device (`out_rgba_device`), valid after the wait.
 *  - pt_upload_objects may be called between submits: frames already submitted keep
 *    the objects they were submitted with (the library keeps as many versions of the
 *    instance data and TLAS as frames in flight need); it does not wait for them.
 *    pt_upload_meshes / pt_upload_materials wait for every frame in flight.
 *  - PT_FRAME_ACCUMULATE frames are added to the context's accumulation buffer in
 *    submit order, whatever order they finish in.
 *  - After pt_render_wait, pt_get_stats / pt_read_aux / pt_debug_wave_log describe
 *    that slot's frame.
 * Still one caller thread per context. */
#define PT_MAX_FRAME_SLOTS 8u
int pt_render_submit(pt_ctx* ctx, const PtFrame* frame, uint32_t slot, void* out_rgba_host);
int pt_render_submit_device(pt_ctx* ctx, const PtFrame* frame, uint32_t slot, void* out_rgba_device);
int pt_render_wait(pt_ctx* ctx, uint32_t slot);

/* Ordering a slot's frame against the caller's own GPU work without blocking the host.
 * The reference's pass is one node of a render graph (between Node3d::EndMainPass and
 * Node3d::MotionBlur, reference src/shader.rs:83-88): what reads its image -- motion
 * blur, bloom, TAA, tonemapping -- runs on the GPU behind it, and the host never waits.
 *  - pt_render_stream_wait: work enqueued on `hip_stream` after this call starts when
 *    the frame submitted to `slot` (and the copy of its image, if any) is complete.
 *    PT_ERR_NOT_READY when the slot holds no submitted frame. The slot still has to be
 *    waited for with pt_render_wait before its next submit.
 *  - pt_render_slot_wait_for: whatever is submitted to `slot` after this call starts
 *    when the work enqueued on `hip_stream` so far is complete (e.g. the last reader
 *    of the image buffer the next frame will overwrite).
 * `hip_stream`: a hipStream_t of the context's device; null: the context's own stream
 * (the one pt_render and pt_render_device(..., null) run on). */
int pt_render_stream_wait(pt_ctx* ctx, uint32_t slot, void* hip_stream);
int pt_render_slot_wait_for(pt_ctx* ctx, uint32_t slot, void* hip_stream);

/* Page-locked host memory owned by the context, for the image pt_render returns
 * (and for the arrays handed to pt_upload_*): copies to and from it run at the
 * PCIe rate, about three times the rate of pageable memory on the test machine.
 * Freed by pt_host_free or pt_destroy. No counterpart in the reference (wgpu's
 * staging buffers play this role there). */
int pt_host_alloc(pt_ctx* ctx, size_t bytes, void** out_ptr);
int pt_host_free(pt_ctx* ctx, void* ptr);

/* ---- several GPUs behind one handle ------------------------------------------ */
/* No counterpart in the reference (one wgpu device, src/shader.rs:121-213). One
 * process drives every listed GPU: the scene is replicated (uploads go to all of
 * them at once, one host thread per GPU inside the call), a frame is partitioned,
 * and the one exchange step brings the f32 images to the first listed GPU with
 * RCCL (ncclSend from every other GPU, ncclRecv on the first, one group call: on
 * xGMI each GPU has its own link to the first one), where they are added in list
 * order, image 0 + image 1 + ..., so the result is reproducible bit for bit for any
 * number of GPUs. librccl is opened at pt_multi_create, it is not a link dependency.
 * With one device the calls are exactly their single-GPU counterparts. */
#define PT_MULTI_MAX_DEVICES 16
enum {
    PT_MULTI_TILES = 0,          /* 8x8 tiles dealt round-robin to the GPUs (PtFrame.part_index/part_count);
                                    the sum of the disjoint images is bit-identical to one GPU's frame       */
    PT_MULTI_VIRTUAL_FRAMES = 1  /* GPU k renders the whole frame with time + k + 0.5 (k > 0): other seeds, as
                                    successive frames of the reference have (src/raytrace.wgsl:210); the result is
                                    (((f0 + f1) + f2) + ...) * (1/N) in f32, N x samples per pixel per call     */
};
/* PtMultiConfig.flags */
#define PT_MULTI_FLAG_HOST_SUM 0x1u /* diagnostic: sum the images on the host, in list order, instead of with RCCL;
                                       a device may then be listed more than once (rehearsal on a one-GPU machine) */
#define PT_MULTI_FLAG_FORCE_COLLECTIVE 0x2u /* diagnostic: a one-device handle also goes through RCCL (a communicator of
                                               one rank that sends its image to itself with the same ncclSend / ncclRecv
                                               calls), instead of being plain pt_render */
typedef struct PtMultiConfig {
    uint32_t abi_version; /* PT_ABI_VERSION */
    uint32_t n_devices;   /* 1..PT_MULTI_MAX_DEVICES */
    int32_t devices[16];  /* [PT_MULTI_MAX_DEVICES] HIP ordinals; the image is returned from devices[0] */
    uint32_t mode;        /* PT_MULTI_* */
    uint32_t accel;       /* PT_ACCEL_* */
    uint32_t tie_rule;    /* PT_TIE_* */
    uint32_t flags;       /* PT_MULTI_FLAG_* */
    uint32_t _reserved[4];
} PtMultiConfig;
typedef struct pt_multi pt_multi;
int pt_multi_create(const PtMultiConfig* cfg, pt_multi** out);
int pt_multi_destroy(pt_multi* m);
/* Message of the most recent error (with m == NULL: of the last failed pt_multi_create). */
const char* pt_multi_last_error(const pt_multi* m);
/* The uploads of pt_upload_*, to every GPU. */
int pt_multi_upload_meshes(pt_multi* m, const PtMesh* meshes, uint32_t n_meshes, const uint32_t* indices,
                           uint32_t n_indices, const PtVertex* vertices, uint32_t n_vertices);
int pt_multi_upload_materials(pt_multi* m, const PtMaterial* materials, uint32_t n_materials,
                              const PtTexture* textures, uint32_t n_textures, const float* texture_data,
                              uint32_t n_texture_data);
int pt_multi_upload_objects(pt_multi* m, const PtObject* objects, uint32_t n_objects);
/* pt_render over all GPUs: width*height RGBA f32 to the host, synchronous. On several GPUs frame flags are not
 * supported and the frame's own part_index/part_count must be unset. */
int pt_multi_render(pt_multi* m, const PtFrame* frame, void* out_rgba_host);
/* The single-GPU context of the k-th listed device (statistics, aux planes, queries). Owned by the handle. */
int pt_multi_context(pt_multi* m, uint32_t k, pt_ctx** out_ctx);
/* RCCL's version number (major*10000 + minor*100 + patch) if librccl can be opened and has the entry points the
 * library uses, else PT_ERR_UNSUPPORTED. */
int pt_multi_rccl_version(int* version);

/* ---- closest-hit query (the query.wgsl entry point on its own) -------------- */

typedef struct PtRay {
    float pos[3];
    float dir[3];
} PtRay;

/* What `hit_all` returns plus the `hit_record` side channel
 * (reference src/query.wgsl:45-64), plus the winning triangle. */
typedef struct PtHit {
    uint32_t object;   /* PT_U32_MAX on miss                                    */
    uint32_t triangle; /* triangle index inside the object's mesh               */
    uint32_t flat_id;  /* triangle id in the order of the reference's double
                          loop (objects outer, triangles inner)                 */
    float t;
    float p[3];
    float n[3];
    float uv[2];
} PtHit;

/* hit_all(ray) for n host rays (reference src/query.wgsl:55-64 with
 * hit_record.t preset to 1000.0 as at src/raytrace.wgsl:230). */
int pt_query(pt_ctx* ctx, const PtRay* rays_host, uint32_t n_rays, PtHit* hits_host);

/* ---- auxiliary read-back ---------------------------------------------------- */

enum {
    PT_AUX_PRIMARY_OBJECT = 0, /* u32[W*H]  object id of the primary hit        */
    PT_AUX_PRIMARY_FLAT_ID = 1,/* u32[W*H]  flat triangle id of the primary hit */
    PT_AUX_PRIMARY_T = 2,      /* f32[W*H]  its distance (1000.0 on miss)       */
    PT_AUX_RAY_COUNT = 3       /* u32[W*H]  closest-hit queries traced per pixel*/
};
/* Valid after a pt_render / pt_render_device of the same size. */
int pt_read_aux(pt_ctx* ctx, uint32_t kind, void* out_host, size_t out_bytes);

/* Work counters of the most recent render on this context. */
typedef struct PtStats {
    uint64_t rays;          /* closest-hit queries executed                     */
    uint64_t rays_reference;/* queries the reference would run for the same
                               image (it re-traces the identical primary ray
                               for every sample, src/raytrace.wgsl:223)          */
    uint64_t paths;         /* pixels * samples                                 */
    uint64_t xforms;        /* ray-object transforms + gate tests               */
    uint64_t box_tests;     /* BVH node-child box tests                          */
    uint64_t tri_tests;     /* ray-triangle tests                               */
    uint64_t shades;        /* scatter evaluations (calculate_brdf)             */
    uint64_t hits;          /* queries that found a surface (hit_record built)  */
    double trace_ms;        /* device time of the frame's kernels, HIP events   */
    double primary_ms;      /* of which the primary pass (k_primary)            */
    double paths_ms;        /* and the path pass (k_paths)                      */
    double build_ms;        /* device time of the last TLAS build                */
    uint32_t trace_launches;/* kernel launches that trace_ms covers              */
    uint32_t accel_used;    /* bit 0: TLAS traversal, bit 1: BLAS traversal,
                               bit 2: the last object upload refitted the TLAS   */
    /* detail only: shader cycles summed over waves spent handing out pixels, in
     * closest hit, and in shading; rounds of the wave loop; lanes tracing per round */
    uint64_t cycles_refill, cycles_trace, cycles_shade, wave_rounds, lane_rounds;
    uint64_t wave_life_sum, wave_life_max, waves; /* lifetime of the path kernel's waves, shader cycles */
    uint32_t wave_life_hist[32];                  /* ... histogram, 2^19 cycles per bucket              */
    /* detail only: the part of xforms / box_tests / tri_tests / hits done by the primary pass */
    uint64_t primary_xforms, primary_box_tests, primary_tri_tests, primary_hits;
    uint64_t object_passes; /* detail only: times a k_paths wave ran the per-object test for its lanes' candidates */
} PtStats;

/* detail != 0 re-runs nothing; it only selects whether the next renders also
 * gather xforms/box_tests/tri_tests/shades (slower kernel variant). */
int pt_set_stats_detail(pt_ctx* ctx, int detail);
int pt_get_stats(pt_ctx* ctx, PtStats* out);

/* Library / device description, for logs. */
int pt_device_info(pt_ctx* ctx, char* buf, size_t buf_bytes);

/* Per-wave records of the last frame's k_paths launches when it was rendered with detailed statistics:
 * 16 x u32 per wave {start, end on the 100 MHz clock, lifetime in 16-cycle units, rounds, lane rounds, first list
 * position, items taken << 8 | XCC_ID, HW_ID, longest item: 16-cycle units, rounds, queue index, 0, trace steps done with
 * four lanes to a ray, 3 x 0}.
 * Copies at most out_bytes; *n_records = records available. Profiling aid, no counterpart in the reference. */
int pt_debug_wave_log(pt_ctx* ctx, void* out_host, size_t out_bytes, uint32_t* n_records);
/* Where a slot's frames stand, without waiting for anything (for a watchdog that finds a wait not returning):
 * out[0] a frame is submitted, out[1] hipStreamQuery of the slot's stream (0: idle), out[2] frames enqueued so far, out[3] how
 * the path kernel of the slot's last frame was set up (bit 31 set; bits 0-3 closest-hit organisation: 0 object list, 1 TLAS,
 * 2 scene in LDS, 3 object list with a BLAS in LDS; bit 4 the build with more registers; bit 5 served for throughput because
 * of the frames in flight; bits 8-11 frames of other slots in flight when it was enqueued); then for
 * each of the slot's two frame records eight words: pending, hipEventQuery of its end event, and from its device counters
 * the entries of the path list, the queue head of k_paths, its waves that have left, the low word of its ray count, the
 * frames of the process that were running on the device when it was scheduled (bit 8: it was scheduled as sharing the
 * device), its chunks handed out in parts. `n_out` >= 20 words. No counterpart in the reference. */
int pt_debug_slot_state(pt_ctx* ctx, uint32_t slot, uint32_t* out, uint32_t n_out);

/* Diagnostics: verifies on the device that every node box of the TLAS and of every
 * BLAS encloses what lies below it; *n_errors = number of violations found.
 * Also reports the tree depths (in nodes) when the pointers are not NULL. */
int pt_debug_check_accel(pt_ctx* ctx, uint32_t* n_errors, uint32_t* tlas_depth, uint32_t* max_blas_depth);

#ifdef __cplusplus
} /* extern "C" */

static_assert(sizeof(PtObject) == 144, "Object stride (src/data.rs:23-30)");
static_assert(offsetof(PtObject, world_to_local) == 64, "Object.world_to_local");
static_assert(offsetof(PtObject, mat) == 128 && offsetof(PtObject, mesh) == 132, "Object.mat/mesh");
static_assert(sizeof(PtMesh) == 48, "GpuMesh stride (src/data.rs:58-66)");
static_assert(offsetof(PtMesh, aabb_max) == 16 && offsetof(PtMesh, ihead) == 28 &&
              offsetof(PtMesh, vhead) == 32 && offsetof(PtMesh, tri_count) == 36, "GpuMesh offsets");
static_assert(sizeof(PtVertex) == 48, "GpuVertex stride (src/data.rs:68-73)");
static_assert(offsetof(PtVertex, normal) == 16 && offsetof(PtVertex, uv) == 32, "GpuVertex offsets");
static_assert(sizeof(PtMaterial) == 64, "Material stride (src/data.rs:32-43)");
static_assert(offsetof(PtMaterial, albedo_texture) == 12 && offsetof(PtMaterial, emissive) == 16 &&
              offsetof(PtMaterial, emissive_texture) == 28 && offsetof(PtMaterial, roughness) == 32 &&
              offsetof(PtMaterial, metallic) == 36 && offsetof(PtMaterial, metallic_roughness_texture) == 40 &&
              offsetof(PtMaterial, reflectance) == 44 && offsetof(PtMaterial, normal_map_texture) == 48,
              "Material offsets");
static_assert(sizeof(PtTexture) == 16, "Texture stride (src/data.rs:45-51)");
static_assert(sizeof(PtSettings) == 32 && offsetof(PtSettings, sky_color) == 16, "Settings (src/data.rs:15-20)");
#endif

#endif /* PT_AMD_H */

// pt_multi.h — several GPUs behind one handle: the pt_multi_* entry points of include/pt_amd.h.
//
// One process, one pt_ctx and one HIP stream per GPU, the scene replicated on every GPU. A frame is partitioned over
// the GPUs in one of two ways (pixels are independent, the samples of one pixel are not: one RNG stream per pixel,
// reference src/raytrace.wgsl:210,221):
//   PT_MULTI_TILES           8x8 tiles dealt round-robin (PtFrame.part_index / part_count); the GPUs' images are
//                            disjoint and zero elsewhere, so their sum is the frame, bit-identical to one GPU's;
//   PT_MULTI_VIRTUAL_FRAMES  every GPU renders all pixels as a frame of its own (`time` + k + 0.5 on GPU k > 0: other
//                            seeds, as successive frames of the reference have); sum, then * 1/N: N x spp per call.
// The one exchange step brings the f32 images to the first GPU, which adds them up in list order: every other GPU sends
// its image (ncclSend), the first one receives them side by side (ncclRecv), all in one group call, then one kernel sums
// image 0 + image 1 + ... in that order. On xGMI every GPU has a link of its own to the first one, so the N - 1 transfers
// run in parallel (33 MB per GPU at 1080p, about 0.2 ms at 153 GB/s per link), and a sum in a fixed order is reproducible
// bit for bit whatever the number of GPUs, which a tree or ring reduction of f32 values is not.
// librccl is opened when the first multi-GPU handle is created, not linked: a process that also loads
// another copy of the collective library (PyTorch ships one) keeps the two apart.
// Included by pt_amd.hip (one translation unit).
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <thread>

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
    bool ok = false;
};
std::mutex g_rccl_mutex;
RcclApi g_rccl;

// Opens librccl once per process. Returns false (g_rccl.error says why) when it cannot be used.
bool rccl_load() {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.ok) return true;
    if (g_rccl.lib) return false;   // tried before
    // PT_AMD_RCCL_LIBRARY, when set, is the only file tried
    const char* named = std::getenv("PT_AMD_RCCL_LIBRARY");
    const bool only_named = named && *named;
    const char* names[] = {named, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    std::string why = "not found";
    for (const char* n : names) {
        if (!n || !*n || (only_named && n != named)) continue;
        g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.lib) break;
        const char* e = dlerror();   // (one call: it hands the message out once and then returns null)
        if (e) why = e;
    }
    if (!g_rccl.lib) {
        g_rccl.error = "librccl could not be opened: " + why;
        g_rccl.lib = reinterpret_cast<void*>(1);
        return false;
    }
    auto sym = [&](const char* name) { return dlsym(g_rccl.lib, name); };
    g_rccl.GetVersion = reinterpret_cast<decltype(g_rccl.GetVersion)>(sym("ncclGetVersion"));
    g_rccl.CommInitAll = reinterpret_cast<decltype(g_rccl.CommInitAll)>(sym("ncclCommInitAll"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
    g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
    g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
    g_rccl.Send = reinterpret_cast<decltype(g_rccl.Send)>(sym("ncclSend"));
    g_rccl.Recv = reinterpret_cast<decltype(g_rccl.Recv)>(sym("ncclRecv"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
    if (!g_rccl.GetVersion || !g_rccl.CommInitAll || !g_rccl.CommDestroy || !g_rccl.GroupStart || !g_rccl.GroupEnd || !g_rccl.Send ||
        !g_rccl.Recv || !g_rccl.GetErrorString) {
        g_rccl.error = "librccl lacks one of ncclGetVersion/CommInitAll/CommDestroy/GroupStart/GroupEnd/Send/Recv/GetErrorString";
        return false;
    }
    g_rccl.ok = true;
    return true;
}

// out = (((first + others[0]) + others[1]) + ...) * s: the images of the GPUs added in list order. `others` holds
// n_others images of n_pixels each, one after the other.
__global__ void k_sum_in_order(float4* out, const float4* first, const float4* __restrict__ others, uint32_t n_others,   // (out may be first)
                               uint32_t n_pixels, float s) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pixels) return;
    float4 v = first[i];
    for (uint32_t k = 0; k < n_others; ++k) {
        const float4 o = others[(size_t)k * n_pixels + i];
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    out[i] = make_float4(v.x * s, v.y * s, v.z * s, v.w * s);
}

}  // namespace

struct pt_multi {
    PtMultiConfig cfg{};
    uint32_t n = 0;
    std::vector<pt_ctx*> ctx;
    std::vector<hipStream_t> stream;
    std::vector<DevBuf> image;        // per GPU: its f32 image
    DevBuf gathered;                  // first GPU: the other GPUs' images, one after the other
    std::vector<ncclComm_t> comm;     // empty: one GPU, or the host-side sum
    std::vector<float> host_a, host_b;  // PT_MULTI_FLAG_HOST_SUM only
    std::string err;
};

namespace {

thread_local std::string g_multi_create_error;

int mfail(pt_multi* m, int code, const std::string& msg) {
    if (m) m->err = msg; else g_multi_create_error = msg;
    return code;
}
// an error of one of the per-GPU contexts, passed on
int mfail_ctx(pt_multi* m, int code, uint32_t k) {
    return mfail(m, code, "GPU " + std::to_string(m->cfg.devices[k]) + ": " + pt_last_error(m->ctx[k]));
}
#define PT_MHIP(m, call)                                                                                    \
    do {                                                                                                    \
        hipError_t e__ = (call);                                                                            \
        if (e__ != hipSuccess)                                                                              \
            return mfail(m, e__ == hipErrorOutOfMemory ? PT_ERR_OUT_OF_MEMORY : PT_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)
#define PT_MNCCL(m, call)                                                                                   \
    do {                                                                                                    \
        ncclResult_t r__ = (call);                                                                          \
        if (r__ != ncclSuccess) return mfail(m, PT_ERR_DEVICE, std::string(#call) + ": " + g_rccl.GetErrorString(r__)); \
    } while (0)

// `time` of GPU k's virtual frame: GPU 0 keeps the caller's; the others add k + 0.5, which keeps the seed product
// (x + 0.5) * (time + 1) of src/raytrace.wgsl:210 away from integer boundaries when the caller's time is a whole number.
float multi_virtual_time(float base, uint32_t k) { return k == 0 ? base : base + (float)k + 0.5f; }

}  // namespace

extern "C" {

const char* pt_multi_last_error(const pt_multi* m) { return m ? m->err.c_str() : g_multi_create_error.c_str(); }

int pt_multi_rccl_version(int* version) {
    if (!version) return PT_ERR_INVALID_ARGUMENT;
    *version = 0;
    if (!rccl_load()) return mfail(nullptr, PT_ERR_UNSUPPORTED, g_rccl.error);
    return g_rccl.GetVersion(version) == ncclSuccess ? PT_OK : PT_ERR_DEVICE;
}

int pt_multi_destroy(pt_multi* m) {
    if (!m) return PT_OK;
    for (uint32_t k = 0; k < m->stream.size(); ++k) {
        if (!m->stream[k]) continue;   // (a handle whose creation failed half-way: devices not reached are left alone)
        (void)hipSetDevice(m->cfg.devices[k]);
        (void)hipStreamSynchronize(m->stream[k]);
    }
    for (ncclComm_t c : m->comm)
        if (c) (void)g_rccl.CommDestroy(c);
    for (uint32_t k = 0; k < m->ctx.size(); ++k) {
        if (!m->ctx[k]) continue;
        (void)hipSetDevice(m->cfg.devices[k]);
        if (k < m->image.size()) m->image[k].release();
        if (k == 0) m->gathered.release();
        if (m->ctx[k]) (void)pt_destroy(m->ctx[k]);
        if (k < m->stream.size() && m->stream[k]) (void)hipStreamDestroy(m->stream[k]);
    }
    delete m;
    return PT_OK;
}

int pt_multi_create(const PtMultiConfig* cfg, pt_multi** out) {
    try {
        if (!cfg || !out) return mfail(nullptr, PT_ERR_INVALID_ARGUMENT, "pt_multi_create: null argument");
        *out = nullptr;
        if (cfg->abi_version != PT_ABI_VERSION) return mfail(nullptr, PT_ERR_INVALID_ARGUMENT, "pt_multi_create: abi_version mismatch");
        if (cfg->n_devices < 1 || cfg->n_devices > PT_MULTI_MAX_DEVICES) return mfail(nullptr, PT_ERR_INVALID_ARGUMENT, "pt_multi_create: n_devices out of range");
        if (cfg->mode > PT_MULTI_VIRTUAL_FRAMES) return mfail(nullptr, PT_ERR_INVALID_ARGUMENT, "pt_multi_create: unknown mode");
        const bool host_sum = (cfg->flags & PT_MULTI_FLAG_HOST_SUM) != 0;
        if (!host_sum)
            for (uint32_t a = 0; a < cfg->n_devices; ++a)
                for (uint32_t b = a + 1; b < cfg->n_devices; ++b)
                    if (cfg->devices[a] == cfg->devices[b]) return mfail(nullptr, PT_ERR_INVALID_ARGUMENT, "pt_multi_create: a device is listed twice");
        pt_multi* m = new pt_multi();
        m->cfg = *cfg;
        m->n = cfg->n_devices;
        m->ctx.assign(m->n, nullptr);
        m->stream.assign(m->n, nullptr);
        m->image.resize(m->n);
        for (uint32_t k = 0; k < m->n; ++k) {
            PtConfig c{};
            c.abi_version = PT_ABI_VERSION;
            c.device = cfg->devices[k];
            c.accel = cfg->accel;
            c.tie_rule = cfg->tie_rule;
            const int rc = pt_create(&c, &m->ctx[k]);
            if (rc != PT_OK) {
                const std::string msg = std::string("pt_multi_create: GPU ") + std::to_string(cfg->devices[k]) + ": " + pt_last_error(nullptr);
                pt_multi_destroy(m);
                return mfail(nullptr, rc, msg);
            }
            hipError_t e = hipSetDevice(cfg->devices[k]);
            if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->stream[k], hipStreamNonBlocking);
            if (e != hipSuccess) {
                const std::string msg = std::string("pt_multi_create: ") + hipGetErrorString(e);
                pt_multi_destroy(m);
                return mfail(nullptr, PT_ERR_DEVICE, msg);
            }
        }
        if ((m->n > 1 || (cfg->flags & PT_MULTI_FLAG_FORCE_COLLECTIVE)) && !host_sum) {
            if (!rccl_load()) {
                const std::string msg = "pt_multi_create: " + g_rccl.error;
                pt_multi_destroy(m);
                return mfail(nullptr, PT_ERR_UNSUPPORTED, msg);
            }
            m->comm.assign(m->n, nullptr);
            const ncclResult_t r = g_rccl.CommInitAll(m->comm.data(), (int)m->n, cfg->devices);
            if (r != ncclSuccess) {
                const std::string msg = std::string("pt_multi_create: ncclCommInitAll: ") + g_rccl.GetErrorString(r);
                m->comm.clear();
                pt_multi_destroy(m);
                return mfail(nullptr, PT_ERR_DEVICE, msg);
            }
        }
        *out = m;
        return PT_OK;
    } catch (const std::bad_alloc&) {
        return mfail(nullptr, PT_ERR_OUT_OF_MEMORY, "pt_multi_create: host allocation failed");
    } catch (...) {
        return mfail(nullptr, PT_ERR_DEVICE, "pt_multi_create: unexpected exception");
    }
}

int pt_multi_context(pt_multi* m, uint32_t k, pt_ctx** out_ctx) {
    if (!m || !out_ctx || k >= m->n) return mfail(m, PT_ERR_INVALID_ARGUMENT, "pt_multi_context: bad argument");
    *out_ctx = m->ctx[k];
    return PT_OK;
}

}  // extern "C"

namespace {

// One upload call per GPU, all at once: an upload validates, copies, builds trees and waits for its device, which for a
// big scene is milliseconds per GPU that need not be paid N times in a row. Each context is driven by one thread, as the
// library expects; the first GPU's upload runs on the calling thread.
template <class Fn>
int multi_upload_all(pt_multi* m, Fn upload) {
    std::vector<int> rc(m->n, PT_OK);
    std::vector<std::thread> workers;
    try {
        for (uint32_t k = 1; k < m->n; ++k) workers.emplace_back([&, k] { rc[k] = upload(m->ctx[k]); });
    } catch (...) {   // (no thread to be had: the rest one after the other)
        for (uint32_t k = (uint32_t)workers.size() + 1u; k < m->n; ++k) rc[k] = upload(m->ctx[k]);
    }
    rc[0] = upload(m->ctx[0]);
    for (std::thread& t : workers) t.join();
    for (uint32_t k = 0; k < m->n; ++k)
        if (rc[k] != PT_OK) return mfail_ctx(m, rc[k], k);
    return PT_OK;
}

}  // namespace

extern "C" {

int pt_multi_upload_meshes(pt_multi* m, const PtMesh* meshes, uint32_t n_meshes, const uint32_t* indices, uint32_t n_indices,
                           const PtVertex* vertices, uint32_t n_vertices) {
    if (!m) return PT_ERR_INVALID_ARGUMENT;
    return multi_upload_all(m, [=](pt_ctx* c) { return pt_upload_meshes(c, meshes, n_meshes, indices, n_indices, vertices, n_vertices); });
}

int pt_multi_upload_materials(pt_multi* m, const PtMaterial* materials, uint32_t n_materials, const PtTexture* textures, uint32_t n_textures,
                              const float* texture_data, uint32_t n_texture_data) {
    if (!m) return PT_ERR_INVALID_ARGUMENT;
    return multi_upload_all(m, [=](pt_ctx* c) { return pt_upload_materials(c, materials, n_materials, textures, n_textures, texture_data, n_texture_data); });
}

int pt_multi_upload_objects(pt_multi* m, const PtObject* objects, uint32_t n_objects) {
    if (!m) return PT_ERR_INVALID_ARGUMENT;
    return multi_upload_all(m, [=](pt_ctx* c) { return pt_upload_objects(c, objects, n_objects); });
}

int pt_multi_render(pt_multi* m, const PtFrame* frame, void* out_rgba_host) {
    if (!m) return PT_ERR_INVALID_ARGUMENT;
    try {
        if (!frame || !out_rgba_host) return mfail(m, PT_ERR_INVALID_ARGUMENT, "pt_multi_render: null argument");
        if (m->n == 1 && !(m->cfg.flags & PT_MULTI_FLAG_FORCE_COLLECTIVE)) {   // one GPU: exactly pt_render
            const int rc = pt_render(m->ctx[0], frame, out_rgba_host);
            return rc == PT_OK ? PT_OK : mfail_ctx(m, rc, 0);
        }
        if (frame->flags != 0u) return mfail(m, PT_ERR_UNSUPPORTED, "pt_multi_render: frame flags are not supported on several GPUs");
        if (frame->part_count > 1u) return mfail(m, PT_ERR_INVALID_ARGUMENT, "pt_multi_render: the frame is partitioned by the library, part_count must be 0 or 1");
        const size_t n_pix = (size_t)frame->width * frame->height, n_floats = 4 * n_pix, bytes = 16 * n_pix;
        const bool tiles = m->cfg.mode == PT_MULTI_TILES;
        // 1. every GPU traces its share on its own stream
        for (uint32_t k = 0; k < m->n; ++k) {
            PT_MHIP(m, hipSetDevice(m->cfg.devices[k]));
            PT_MHIP(m, m->image[k].ensure(bytes));
            PtFrame f = *frame;
            if (tiles) { f.part_index = k; f.part_count = m->n; }
            else { f.part_index = 0; f.part_count = 1; f.time = multi_virtual_time(frame->time, k); }
            const int rc = pt_render_device(m->ctx[k], &f, m->image[k].p, m->stream[k]);
            if (rc != PT_OK) return mfail_ctx(m, rc, k);
        }
        const float scale = tiles ? 1.0f : 1.0f / (float)m->n;
        if (m->cfg.flags & PT_MULTI_FLAG_HOST_SUM) {
            // diagnostic path (one GPU listed several times, no collective library): the images are summed on the host, in
            // GPU order
            m->host_a.assign(n_floats, 0.0f);
            m->host_b.resize(n_floats);
            for (uint32_t k = 0; k < m->n; ++k) {
                PT_MHIP(m, hipSetDevice(m->cfg.devices[k]));
                PT_MHIP(m, hipMemcpyAsync(m->host_b.data(), m->image[k].p, bytes, hipMemcpyDeviceToHost, m->stream[k]));
                PT_MHIP(m, hipStreamSynchronize(m->stream[k]));
                if (k == 0) m->host_a = m->host_b;
                else for (size_t i = 0; i < n_floats; ++i) m->host_a[i] += m->host_b[i];
            }
            float* out = static_cast<float*>(out_rgba_host);
            for (size_t i = 0; i < n_floats; ++i) out[i] = tiles ? m->host_a[i] : m->host_a[i] * scale;
            return PT_OK;
        }
        // 2. the images come to the first GPU: every other GPU sends, the first one receives them side by side, each call
        //    behind its GPU's frame on that GPU's stream, all in one group. (A one-GPU handle with the diagnostic flag sends
        //    its image to itself, which takes the same calls.)
        const bool self = m->n == 1;
        const uint32_t n_others = self ? 1u : m->n - 1u;
        PT_MHIP(m, hipSetDevice(m->cfg.devices[0]));
        PT_MHIP(m, m->gathered.ensure(bytes * n_others));
        PT_MNCCL(m, g_rccl.GroupStart());
        for (uint32_t k = self ? 0u : 1u; k < m->n; ++k) {
            ncclResult_t r = g_rccl.Send(m->image[k].p, n_floats, ncclFloat, 0, m->comm[k], m->stream[k]);
            if (r == ncclSuccess)
                r = g_rccl.Recv(m->gathered.as<float>() + (size_t)(self ? 0u : k - 1u) * n_floats, n_floats, ncclFloat, (int)k, m->comm[0], m->stream[0]);
            if (r != ncclSuccess) { (void)g_rccl.GroupEnd(); return mfail(m, PT_ERR_DEVICE, std::string("ncclSend/ncclRecv: ") + g_rccl.GetErrorString(r)); }
        }
        PT_MNCCL(m, g_rccl.GroupEnd());
        // 3. sum in list order (mean of the virtual frames), copy to the host
        PT_MHIP(m, hipSetDevice(m->cfg.devices[0]));
        if (self)   // (the image that came back, times one)
            k_sum_in_order<<<div_up((uint32_t)n_pix, 256), 256, 0, m->stream[0]>>>(m->image[0].as<float4>(), m->gathered.as<float4>(), nullptr, 0u, (uint32_t)n_pix, 1.0f);
        else
            k_sum_in_order<<<div_up((uint32_t)n_pix, 256), 256, 0, m->stream[0]>>>(m->image[0].as<float4>(), m->image[0].as<float4>(), m->gathered.as<float4>(), n_others,
                                                                                  (uint32_t)n_pix, scale);
        PT_MHIP(m, hipGetLastError());
        PT_MHIP(m, hipMemcpyAsync(out_rgba_host, m->image[0].p, bytes, hipMemcpyDeviceToHost, m->stream[0]));
        for (uint32_t k = 0; k < m->n; ++k) {
            PT_MHIP(m, hipSetDevice(m->cfg.devices[k]));
            PT_MHIP(m, hipStreamSynchronize(m->stream[k]));
        }
        return PT_OK;
    } catch (const std::bad_alloc&) {
        return mfail(m, PT_ERR_OUT_OF_MEMORY, "pt_multi_render: host allocation failed");
    } catch (...) {
        return mfail(m, PT_ERR_DEVICE, "pt_multi_render: unexpected exception");
    }
}

}  // extern "C"

// raytrace_plugin.hpp — C++ mirror of the reference plugin's surface for its render pass, over the C ABI.
//
// Same names and argument meaning as the Rust side (reference src/lib.rs:1-7, src/shader.rs:35-213,
// src/data.rs:15-97): RayTraceSettings, RayTraceMeta, RayTracePlugin{build,finish}, RayTraceNode::run.
// The wgpu render pass the reference records in RayTraceNode::run is replaced by pt_render().
// Error behaviour: like the reference, run() draws nothing (returns false) while the plugin is not ready
// (src/shader.rs:133-145); where the reference would panic on unusable buffers (src/shader.rs:168-187) a
// PtException carrying the library's error code is thrown on this side of the ABI.
#pragma once

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "pt_amd.h"

namespace path_tracing {

struct PtException : std::runtime_error {
    int code;
    PtException(int c, const std::string& m) : std::runtime_error("pt_amd error " + std::to_string(c) + ": " + m), code(c) {}
};

// reference src/data.rs:15-20 (#[derive(Default)]: all zero)
struct RayTraceSettings {
    uint32_t bounces = 0;
    uint32_t samples = 0;
    float sky_color[4] = {0, 0, 0, 0};  // LinearRgba
};

// the fields of bevy_render::view::View the shader reads (reference src/raytrace.wgsl:210-217)
struct View {
    float world_from_view[16];
    float view_from_clip[16];
    float viewport[4];  // x, y, width, height
};

// bevy_render::globals::Globals (only `time` is read, reference src/raytrace.wgsl:210)
struct Globals {
    float time = 0.0f;
};

// reference src/data.rs:83-97: the storage buffers the extract systems fill
struct RayTraceMeta {
    std::vector<PtObject> objects;
    std::vector<uint32_t> emissives;  // bound but never read by the shader (src/query.wgsl:9); not uploaded
    std::vector<PtMesh> meshes;
    std::vector<uint32_t> indices;
    std::vector<PtVertex> vertices;
    std::vector<PtMaterial> materials;
    std::vector<PtTexture> textures;
    std::vector<float> texture_data;
    bool meshes_dirty = true, materials_dirty = true, objects_dirty = true;
};

// bevy_render::render_resource::TextureFormat, the variants the reference accepts (src/data.rs:104-150)
enum class TextureFormat { Rgba8UnormSrgb, Rgba16Float, Rgb9e5Ufloat, R8Unorm, Rg8Unorm, Other };

// binary16 bit pattern -> f32 (what `f16 as f32` does: exact)
inline float half_bits_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else {  // subnormal half: normalise
            int e = -1;
            do { man <<= 1; ++e; } while (!(man & 0x400u));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    static_assert(sizeof f == sizeof bits, "f32");
    std::memcpy(&f, &bits, sizeof f);
    return f;
}

// reference src/data.rs:53-56 and :99-160 (TextureData::append_texture): how every image asset is flattened into
// the f32 texel pool, quirks included (Rgba16Float texels are also scaled by 1/255; the Rgb9e5Ufloat branch does its
// own u8 bit arithmetic). A format the reference panics on throws std::invalid_argument here.
struct TextureData {
    std::vector<float> data;

    PtTexture append_texture(const uint8_t* bytes, size_t n_bytes, uint32_t width, uint32_t height, TextureFormat format) {
        const float DIV_255 = 1.0f / 255.0f;
        const uint32_t offset = (uint32_t)data.size();
        uint32_t channels = 0;
        switch (format) {
            case TextureFormat::Rgba8UnormSrgb:
            case TextureFormat::R8Unorm:
            case TextureFormat::Rg8Unorm:
                for (size_t i = 0; i < n_bytes; ++i) data.push_back((float)bytes[i] * DIV_255);
                channels = format == TextureFormat::Rgba8UnormSrgb ? 4u : (format == TextureFormat::R8Unorm ? 1u : 2u);
                break;
            case TextureFormat::Rgba16Float:
                if (n_bytes % 2) throw std::invalid_argument("Rgba16Float image data has an odd number of bytes");
                for (size_t i = 0; i < n_bytes; i += 2)
                    data.push_back(half_bits_to_float((uint16_t)(bytes[i] | (bytes[i + 1] << 8))) * DIV_255);
                channels = 4;
                break;
            case TextureFormat::Rgb9e5Ufloat:
                if (n_bytes % 4) throw std::invalid_argument("Rgb9e5Ufloat image data is not a multiple of 4 bytes");
                for (size_t i = 0; i < n_bytes; i += 4) {
                    const uint8_t* d = bytes + i;
                    const uint8_t e = (uint8_t)((uint8_t)(d[3] << 2) & 0b011111);
                    const uint8_t r[2] = {(uint8_t)(e | (d[0] >> 7)), (uint8_t)((uint8_t)(d[0] << 1) | ((d[1] & 0b1) >> 7))};
                    const uint8_t g[2] = {(uint8_t)(e | ((d[1] & 0b01) >> 6)), (uint8_t)((uint8_t)(d[1] << 2) | ((d[2] & 0b11) >> 6))};
                    const uint8_t b[2] = {(uint8_t)(e | ((d[2] & 0b001) >> 5)), (uint8_t)((uint8_t)(d[2] << 3) | ((d[3] & 0b111) >> 5))};
                    data.push_back(half_bits_to_float((uint16_t)(r[0] | (r[1] << 8))));
                    data.push_back(half_bits_to_float((uint16_t)(g[0] | (g[1] << 8))));
                    data.push_back(half_bits_to_float((uint16_t)(b[0] | (b[1] << 8))));
                }
                channels = 3;
                break;
            default:
                throw std::invalid_argument("Texture format is not supported.");
        }
        PtTexture t;
        t.width = width;
        t.height = height;
        t.offset = offset;
        t.format = channels;
        return t;
    }
};

class RayTracePlugin;

// reference src/shader.rs:111-214
class RayTraceNode {
public:
    explicit RayTraceNode(RayTracePlugin* p) : plugin_(p) {}
    // Renders the view into `target` (width*height*4 floats, row 0 on top). Returns false without touching
    // `target` when the plugin is not ready or the camera has no settings.
    bool run(const View& view, const Globals& globals, const RayTraceSettings* settings, std::vector<float>& target, uint32_t flags = 0);
    // The same pass with frames in flight (pt_render_submit / pt_render_wait): `submit` enqueues the view's frame on `slot`
    // (0 .. PT_MAX_FRAME_SLOTS - 1) and returns; `target` (width*height*4 floats, best from pt_host_alloc) holds the image
    // once `wait(slot)` has returned. A node that keeps K frames in flight submits frame n to slot n % K and waits for
    // that slot right before it submits to it again. Returns false (nothing enqueued) when the plugin is not ready.
    bool submit(const View& view, const Globals& globals, const RayTraceSettings* settings, uint32_t slot, float* target, uint32_t flags = 0);
    void wait(uint32_t slot);
    // For a renderer whose later passes read the image on the GPU (the reference's node sits between EndMainPass and MotionBlur,
    // src/shader.rs:83-88): `submit_device` leaves the image in device memory at `device_target`; `stream_wait` orders the work
    // enqueued on `hip_stream` afterwards behind the slot's frame, `slot_wait_for` the slot's next frame behind the work enqueued
    // on `hip_stream` so far. Neither blocks the host; `wait(slot)` is still due before the slot's next submit.
    bool submit_device(const View& view, const Globals& globals, const RayTraceSettings* settings, uint32_t slot, void* device_target, uint32_t flags = 0);
    void stream_wait(uint32_t slot, void* hip_stream);
    void slot_wait_for(uint32_t slot, void* hip_stream);

private:
    static PtFrame make_frame(const View& view, const Globals& globals, const RayTraceSettings& settings, uint32_t flags);
    RayTracePlugin* plugin_;
};

// reference src/shader.rs:35-104
class RayTracePlugin {
public:
    explicit RayTracePlugin(int device = 0, uint32_t accel = PT_ACCEL_AUTO, uint32_t tie_rule = PT_TIE_REFERENCE)
        : node(this), device_(device), accel_(accel), tie_(tie_rule) {}
    ~RayTracePlugin() { if (ctx_) pt_destroy(ctx_); }
    RayTracePlugin(const RayTracePlugin&) = delete;
    RayTracePlugin& operator=(const RayTracePlugin&) = delete;

    void build() { built_ = true; meta = RayTraceMeta(); }  // :38-95 inserts RayTraceMeta with empty buffers
    void finish() {                                           // :97-103 creates the pipeline; here the device context
        if (!built_) throw std::logic_error("RayTracePlugin::finish() before build()");
        PtConfig cfg{};
        cfg.abi_version = PT_ABI_VERSION;
        cfg.device = device_;
        cfg.accel = accel_;
        cfg.tie_rule = tie_;
        int rc = pt_create(&cfg, &ctx_);
        if (rc != PT_OK) throw PtException(rc, pt_last_error(nullptr));
    }
    // the write_buffer calls of prepare_meshes / extract_materials / extract_textures / extract_visible
    void prepare() {
        check(meta.meshes_dirty ? pt_upload_meshes(ctx_, meta.meshes.data(), (uint32_t)meta.meshes.size(), meta.indices.data(),
                                                   (uint32_t)meta.indices.size(), meta.vertices.data(), (uint32_t)meta.vertices.size()) : PT_OK);
        check(meta.materials_dirty ? pt_upload_materials(ctx_, meta.materials.data(), (uint32_t)meta.materials.size(), meta.textures.data(),
                                                         (uint32_t)meta.textures.size(), meta.texture_data.data(), (uint32_t)meta.texture_data.size()) : PT_OK);
        check(meta.objects_dirty ? pt_upload_objects(ctx_, meta.objects.data(), (uint32_t)meta.objects.size()) : PT_OK);
        meta.meshes_dirty = meta.materials_dirty = meta.objects_dirty = false;
    }
    bool ready() const { return ctx_ != nullptr; }
    pt_ctx* context() const { return ctx_; }
    void check(int rc) const { if (rc != PT_OK) throw PtException(rc, pt_last_error(ctx_)); }

    RayTraceMeta meta;
    RayTraceNode node;

private:
    int device_;
    uint32_t accel_, tie_;
    bool built_ = false;
    pt_ctx* ctx_ = nullptr;
};

inline PtFrame RayTraceNode::make_frame(const View& view, const Globals& globals, const RayTraceSettings& settings, uint32_t flags) {
    PtFrame f{};
    for (int i = 0; i < 16; ++i) { f.world_from_view[i] = view.world_from_view[i]; f.view_from_clip[i] = view.view_from_clip[i]; }
    f.width = (uint32_t)view.viewport[2];
    f.height = (uint32_t)view.viewport[3];
    f.time = globals.time;
    f.flags = flags;
    f.settings.bounces = settings.bounces;
    f.settings.samples = settings.samples;
    for (int i = 0; i < 4; ++i) f.settings.sky_color[i] = settings.sky_color[i];
    f.part_index = 0;
    f.part_count = 1;
    return f;
}

inline bool RayTraceNode::run(const View& view, const Globals& globals, const RayTraceSettings* settings, std::vector<float>& target, uint32_t flags) {
    if (!plugin_->ready() || settings == nullptr) return false;
    plugin_->prepare();
    const PtFrame f = make_frame(view, globals, *settings, flags);
    target.resize((size_t)f.width * f.height * 4);
    plugin_->check(pt_render(plugin_->context(), &f, target.data()));
    return true;
}

inline bool RayTraceNode::submit(const View& view, const Globals& globals, const RayTraceSettings* settings, uint32_t slot, float* target, uint32_t flags) {
    if (!plugin_->ready() || settings == nullptr) return false;
    plugin_->prepare();   // (an object upload here does not wait for the frames in flight: they keep their version)
    const PtFrame f = make_frame(view, globals, *settings, flags);
    plugin_->check(pt_render_submit(plugin_->context(), &f, slot, target));
    return true;
}

inline void RayTraceNode::wait(uint32_t slot) { plugin_->check(pt_render_wait(plugin_->context(), slot)); }

inline bool RayTraceNode::submit_device(const View& view, const Globals& globals, const RayTraceSettings* settings, uint32_t slot, void* device_target, uint32_t flags) {
    if (!plugin_->ready() || settings == nullptr) return false;
    plugin_->prepare();
    const PtFrame f = make_frame(view, globals, *settings, flags);
    plugin_->check(pt_render_submit_device(plugin_->context(), &f, slot, device_target));
    return true;
}

inline void RayTraceNode::stream_wait(uint32_t slot, void* hip_stream) { plugin_->check(pt_render_stream_wait(plugin_->context(), slot, hip_stream)); }
inline void RayTraceNode::slot_wait_for(uint32_t slot, void* hip_stream) { plugin_->check(pt_render_slot_wait_for(plugin_->context(), slot, hip_stream)); }

}  // namespace path_tracing

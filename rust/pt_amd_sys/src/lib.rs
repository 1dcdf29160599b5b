//! Raw bindings to `include/pt_amd.h` (ABI version 1) and a thin safe wrapper.
//!
//! The wire structs are `#[repr(C)]` mirrors of the C header, which in turn are the encase/std430 layouts of the
//! plugin's `src/data.rs` (`Object` 144 B, `GpuMesh` 48 B, `GpuVertex` 48 B, `Material` 64 B, `Texture` 16 B,
//! `RayTraceSettings` 32 B). The plugin can therefore keep filling its `encase::StorageBuffer<Vec<T>>` scratch
//! buffers and hand their bytes to `upload_*`, or build `Vec<PtObject>` etc. directly.
//!
//! This crate was written without a Rust toolchain at hand; the `layout` tests below restate the header's
//! `static_assert`s, and the repository's Python test `tests/test_rust_binding.py` checks every `assert_size!` /
//! `assert_offset!` line of this file against the C header as compiled.
#![allow(non_camel_case_types)]

use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

pub const PT_ABI_VERSION: u32 = 1;

pub const PT_OK: c_int = 0;
pub const PT_ERR_INVALID_ARGUMENT: c_int = -1;
pub const PT_ERR_INVALID_SCENE: c_int = -2;
pub const PT_ERR_NOT_READY: c_int = -3;
pub const PT_ERR_DEVICE: c_int = -4;
pub const PT_ERR_OUT_OF_MEMORY: c_int = -5;
pub const PT_ERR_UNSUPPORTED: c_int = -6;
pub const PT_ERR_NO_DEVICE: c_int = -7;

pub const PT_U32_MAX: u32 = 0xFFFF_FFFF;

pub const PT_FRAME_ACCUMULATE: u32 = 0x1;
pub const PT_FRAME_RESET: u32 = 0x2;
pub const PT_FRAME_OUTPUT_F16: u32 = 0x4;

pub const PT_ACCEL_AUTO: u32 = 0;
pub const PT_ACCEL_BRUTE: u32 = 1;
pub const PT_ACCEL_BVH: u32 = 2;

pub const PT_TIE_REFERENCE: u32 = 0;
pub const PT_TIE_LOWEST: u32 = 1;

pub const PT_AUX_PRIMARY_OBJECT: u32 = 0;
pub const PT_AUX_PRIMARY_FLAT_ID: u32 = 1;
pub const PT_AUX_PRIMARY_T: u32 = 2;
pub const PT_AUX_RAY_COUNT: u32 = 3;

/// `Object` (src/data.rs:23-30)
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PtObject {
    pub local_to_world: [f32; 16],
    pub world_to_local: [f32; 16],
    pub mat: u32,
    pub mesh: u32,
    pub _pad: [u32; 2],
}

/// `GpuMesh` (src/data.rs:58-66)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtMesh {
    pub aabb_min: [f32; 3],
    pub _pad0: u32,
    pub aabb_max: [f32; 3],
    pub ihead: u32,
    pub vhead: u32,
    pub tri_count: u32,
    pub _pad1: [u32; 2],
}

/// `GpuVertex` (src/data.rs:68-73)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtVertex {
    pub position: [f32; 3],
    pub _pad0: u32,
    pub normal: [f32; 3],
    pub _pad1: u32,
    pub uv: [f32; 2],
    pub _pad2: [u32; 2],
}

/// `Material` (src/data.rs:32-43)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtMaterial {
    pub albedo: [f32; 3],
    pub albedo_texture: u32,
    pub emissive: [f32; 3],
    pub emissive_texture: u32,
    pub roughness: f32,
    pub metallic: f32,
    pub metallic_roughness_texture: u32,
    pub reflectance: f32,
    pub normal_map_texture: u32,
    pub _pad: [u32; 3],
}

/// `Texture` (src/data.rs:45-51)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtTexture {
    pub width: u32,
    pub height: u32,
    pub offset: u32,
    pub format: u32,
}

/// `RayTraceSettings` (src/data.rs:15-20)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtSettings {
    pub bounces: u32,
    pub samples: u32,
    pub _pad: [u32; 2],
    pub sky_color: [f32; 4],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct PtFrame {
    pub world_from_view: [f32; 16],
    pub view_from_clip: [f32; 16],
    pub width: u32,
    pub height: u32,
    pub time: f32,
    pub flags: u32,
    pub settings: PtSettings,
    pub part_index: u32,
    pub part_count: u32,
    pub _reserved: [u32; 2],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtConfig {
    pub abi_version: u32,
    pub device: i32,
    pub accel: u32,
    pub tie_rule: u32,
    pub flags: u32,
    pub _reserved: [u32; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtRay {
    pub pos: [f32; 3],
    pub dir: [f32; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtHit {
    pub object: u32,
    pub triangle: u32,
    pub flat_id: u32,
    pub t: f32,
    pub p: [f32; 3],
    pub n: [f32; 3],
    pub uv: [f32; 2],
}

#[repr(C)]
pub struct pt_ctx {
    _private: [u8; 0],
}

pub const PT_MULTI_MAX_DEVICES: u32 = 16;
pub const PT_MULTI_TILES: u32 = 0;
pub const PT_MULTI_VIRTUAL_FRAMES: u32 = 1;
pub const PT_MULTI_FLAG_HOST_SUM: u32 = 0x1;
pub const PT_MULTI_FLAG_FORCE_COLLECTIVE: u32 = 0x2;
pub const PT_NOALLOC_MAX_PIXELS: u32 = 262144;
pub const PT_MAX_FRAME_SLOTS: u32 = 8;

/// Several GPUs behind one handle (pt_multi_*): device list, partition mode, accel and tie rule for every GPU.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct PtMultiConfig {
    pub abi_version: u32,
    pub n_devices: u32,
    pub devices: [i32; 16],
    pub mode: u32,
    pub accel: u32,
    pub tie_rule: u32,
    pub flags: u32,
    pub _reserved: [u32; 4],
}

#[repr(C)]
pub struct pt_multi {
    _private: [u8; 0],
}

extern "C" {
    pub fn pt_create(cfg: *const PtConfig, out_ctx: *mut *mut pt_ctx) -> c_int;
    pub fn pt_destroy(ctx: *mut pt_ctx) -> c_int;
    pub fn pt_last_error(ctx: *const pt_ctx) -> *const c_char;
    pub fn pt_upload_meshes(ctx: *mut pt_ctx, meshes: *const PtMesh, n_meshes: u32, indices: *const u32, n_indices: u32,
                            vertices: *const PtVertex, n_vertices: u32) -> c_int;
    pub fn pt_upload_materials(ctx: *mut pt_ctx, materials: *const PtMaterial, n_materials: u32, textures: *const PtTexture,
                               n_textures: u32, texture_data: *const f32, n_texture_data: u32) -> c_int;
    pub fn pt_upload_objects(ctx: *mut pt_ctx, objects: *const PtObject, n_objects: u32) -> c_int;
    pub fn pt_render(ctx: *mut pt_ctx, frame: *const PtFrame, out_rgba_host: *mut c_void) -> c_int;
    pub fn pt_render_device(ctx: *mut pt_ctx, frame: *const PtFrame, out_rgba_device: *mut c_void, hip_stream: *mut c_void) -> c_int;
    /// The same frame without device-memory allocation (frames of at most PT_NOALLOC_MAX_PIXELS pixels).
    pub fn pt_render_device_noalloc(ctx: *mut pt_ctx, frame: *const PtFrame, out_rgba_device: *mut c_void, hip_stream: *mut c_void) -> c_int;
    /// Frames in flight on one context: enqueue on `slot` (0..PT_MAX_FRAME_SLOTS), image copied to `out_rgba_host` behind the frame.
    pub fn pt_render_submit(ctx: *mut pt_ctx, frame: *const PtFrame, slot: u32, out_rgba_host: *mut c_void) -> c_int;
    pub fn pt_render_submit_device(ctx: *mut pt_ctx, frame: *const PtFrame, slot: u32, out_rgba_device: *mut c_void) -> c_int;
    /// Blocks until the frame submitted to `slot` (and its copy) is complete.
    pub fn pt_render_wait(ctx: *mut pt_ctx, slot: u32) -> c_int;
    /// GPU-side ordering, no host wait: work enqueued on `hip_stream` afterwards starts when the slot's frame is complete.
    pub fn pt_render_stream_wait(ctx: *mut pt_ctx, slot: u32, hip_stream: *mut c_void) -> c_int;
    /// ... and what is submitted to `slot` afterwards starts when the work enqueued on `hip_stream` so far is complete.
    pub fn pt_render_slot_wait_for(ctx: *mut pt_ctx, slot: u32, hip_stream: *mut c_void) -> c_int;
    pub fn pt_multi_create(cfg: *const PtMultiConfig, out: *mut *mut pt_multi) -> c_int;
    pub fn pt_multi_destroy(m: *mut pt_multi) -> c_int;
    pub fn pt_multi_last_error(m: *const pt_multi) -> *const c_char;
    pub fn pt_multi_upload_meshes(m: *mut pt_multi, meshes: *const PtMesh, n_meshes: u32, indices: *const u32, n_indices: u32,
                                  vertices: *const PtVertex, n_vertices: u32) -> c_int;
    pub fn pt_multi_upload_materials(m: *mut pt_multi, materials: *const PtMaterial, n_materials: u32, textures: *const PtTexture,
                                     n_textures: u32, texture_data: *const f32, n_texture_data: u32) -> c_int;
    pub fn pt_multi_upload_objects(m: *mut pt_multi, objects: *const PtObject, n_objects: u32) -> c_int;
    pub fn pt_multi_render(m: *mut pt_multi, frame: *const PtFrame, out_rgba_host: *mut c_void) -> c_int;
    pub fn pt_multi_context(m: *mut pt_multi, k: u32, out_ctx: *mut *mut pt_ctx) -> c_int;
    pub fn pt_multi_rccl_version(version: *mut c_int) -> c_int;
    pub fn pt_host_alloc(ctx: *mut pt_ctx, bytes: usize, out_ptr: *mut *mut c_void) -> c_int;
    pub fn pt_host_free(ctx: *mut pt_ctx, ptr: *mut c_void) -> c_int;
    pub fn pt_query(ctx: *mut pt_ctx, rays_host: *const PtRay, n_rays: u32, hits_host: *mut PtHit) -> c_int;
    pub fn pt_read_aux(ctx: *mut pt_ctx, kind: u32, out_host: *mut c_void, out_bytes: usize) -> c_int;
    pub fn pt_device_info(ctx: *mut pt_ctx, buf: *mut c_char, buf_bytes: usize) -> c_int;
}

/// Error code of the library plus its message.
#[derive(Debug, Clone)]
pub struct PtError {
    pub code: c_int,
    pub message: String,
}

impl std::fmt::Display for PtError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "pt_amd error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for PtError {}

/// Owns a `pt_ctx`. One per GPU; `Send` (the library allows one caller thread at a time per context), not `Sync`.
pub struct Context {
    raw: *mut pt_ctx,
}
unsafe impl Send for Context {}

impl Context {
    /// `RayTracePlugin::finish` creates this instead of `RayTracePipeline` (src/shader.rs:93-103, :225-339).
    pub fn new(device: i32, accel: u32, tie_rule: u32) -> Result<Self, PtError> {
        let cfg = PtConfig { abi_version: PT_ABI_VERSION, device, accel, tie_rule, flags: 0, _reserved: [0; 3] };
        let mut raw: *mut pt_ctx = std::ptr::null_mut();
        let rc = unsafe { pt_create(&cfg, &mut raw) };
        if rc != PT_OK {
            let message = unsafe { CStr::from_ptr(pt_last_error(std::ptr::null())) }.to_string_lossy().into_owned();
            return Err(PtError { code: rc, message });
        }
        Ok(Self { raw })
    }

    fn check(&self, rc: c_int) -> Result<(), PtError> {
        if rc == PT_OK {
            return Ok(());
        }
        let message = unsafe { CStr::from_ptr(pt_last_error(self.raw)) }.to_string_lossy().into_owned();
        Err(PtError { code: rc, message })
    }

    /// Replaces the three `write_buffer` calls of `prepare_meshes` (src/extract.rs:131-144).
    pub fn upload_meshes(&mut self, meshes: &[PtMesh], indices: &[u32], vertices: &[PtVertex]) -> Result<(), PtError> {
        let rc = unsafe {
            pt_upload_meshes(self.raw, meshes.as_ptr(), meshes.len() as u32, indices.as_ptr(), indices.len() as u32,
                             vertices.as_ptr(), vertices.len() as u32)
        };
        self.check(rc)
    }

    /// Replaces the `write_buffer` calls of `extract_materials` / `extract_textures` (src/extract.rs:249-253, :284-292).
    pub fn upload_materials(&mut self, materials: &[PtMaterial], textures: &[PtTexture], texture_data: &[f32]) -> Result<(), PtError> {
        let rc = unsafe {
            pt_upload_materials(self.raw, materials.as_ptr(), materials.len() as u32, textures.as_ptr(), textures.len() as u32,
                                texture_data.as_ptr(), texture_data.len() as u32)
        };
        self.check(rc)
    }

    /// Replaces `objects.write_buffer` of `extract_visible` (src/extract.rs:337-342).
    pub fn upload_objects(&mut self, objects: &[PtObject]) -> Result<(), PtError> {
        let rc = unsafe { pt_upload_objects(self.raw, objects.as_ptr(), objects.len() as u32) };
        self.check(rc)
    }

    /// Replaces the body of `RayTraceNode::run` (src/shader.rs:121-213). `out` holds `width * height * 4` f32.
    pub fn render(&mut self, frame: &PtFrame, out: &mut [f32]) -> Result<(), PtError> {
        assert!(frame.flags & PT_FRAME_OUTPUT_F16 == 0, "use render_f16 for binary16 output");
        assert!(out.len() >= frame.width as usize * frame.height as usize * 4);
        let rc = unsafe { pt_render(self.raw, frame, out.as_mut_ptr() as *mut c_void) };
        self.check(rc)
    }

    /// Same with `PT_FRAME_OUTPUT_F16`: `out` holds `width * height * 4` binary16 bit patterns, ready for
    /// `queue.write_texture` into the `Rgba16Float` view target (src/shader.rs:321).
    pub fn render_f16(&mut self, frame: &PtFrame, out: &mut [u16]) -> Result<(), PtError> {
        assert!(out.len() >= frame.width as usize * frame.height as usize * 4);
        let mut f = *frame;
        f.flags |= PT_FRAME_OUTPUT_F16;
        let rc = unsafe { pt_render(self.raw, &f, out.as_mut_ptr() as *mut c_void) };
        self.check(rc)
    }

    /// Enqueues a frame on `slot` and returns; the binary16 image is in `out` once `wait(slot)` has returned. `out` must be
    /// page-locked memory from `host_alloc` that stays valid until then (hence the raw pointer): `width * height * 4` u16.
    /// A render node that keeps K frames in flight submits frame n to slot n % K and waits for frame n - K + 1.
    ///
    /// # Safety
    /// `out` must point to at least `width * height * 8` writable bytes that outlive the matching `wait`.
    pub unsafe fn submit_f16(&mut self, frame: &PtFrame, slot: u32, out: *mut u16) -> Result<(), PtError> {
        let mut f = *frame;
        f.flags |= PT_FRAME_OUTPUT_F16;
        let rc = pt_render_submit(self.raw, &f, slot, out as *mut c_void);
        self.check(rc)
    }

    /// Blocks until the frame submitted to `slot` is complete.
    pub fn wait(&mut self, slot: u32) -> Result<(), PtError> {
        let rc = unsafe { pt_render_wait(self.raw, slot) };
        self.check(rc)
    }

    /// Orders the caller's stream behind the frame of `slot` on the GPU (the render graph's downstream nodes).
    ///
    /// # Safety
    /// `hip_stream` must be a live hipStream_t of the context's device, or null for the context's own stream.
    pub unsafe fn stream_wait(&mut self, slot: u32, hip_stream: *mut c_void) -> Result<(), PtError> {
        let rc = pt_render_stream_wait(self.raw, slot, hip_stream);
        self.check(rc)
    }

    /// Orders the slot's next frame behind the work enqueued on `hip_stream` so far.
    ///
    /// # Safety
    /// As for `stream_wait`.
    pub unsafe fn slot_wait_for(&mut self, slot: u32, hip_stream: *mut c_void) -> Result<(), PtError> {
        let rc = pt_render_slot_wait_for(self.raw, slot, hip_stream);
        self.check(rc)
    }

    /// Page-locked staging memory owned by the context (freed with it): copies into it run at the PCIe rate.
    pub fn host_alloc(&mut self, bytes: usize) -> Result<*mut c_void, PtError> {
        let mut p: *mut c_void = std::ptr::null_mut();
        let rc = unsafe { pt_host_alloc(self.raw, bytes, &mut p) };
        self.check(rc).map(|_| p)
    }

    pub fn raw(&self) -> *mut pt_ctx {
        self.raw
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { pt_destroy(self.raw) };
    }
}

macro_rules! assert_size {
    ($t:ty, $n:expr) => {
        const _: () = assert!(std::mem::size_of::<$t>() == $n);
    };
}
macro_rules! assert_offset {
    ($t:ty, $f:ident, $n:expr) => {
        const _: () = assert!(std::mem::offset_of!($t, $f) == $n);
    };
}

// The same numbers as the static_asserts of include/pt_amd.h (checked at compile time here).
assert_size!(PtObject, 144);
assert_size!(PtMesh, 48);
assert_size!(PtVertex, 48);
assert_size!(PtMaterial, 64);
assert_size!(PtTexture, 16);
assert_size!(PtSettings, 32);
assert_size!(PtFrame, 192);
assert_size!(PtConfig, 32);
assert_size!(PtRay, 24);
assert_size!(PtHit, 48);
assert_offset!(PtObject, world_to_local, 64);
assert_offset!(PtObject, mat, 128);
assert_offset!(PtObject, mesh, 132);
assert_offset!(PtMesh, aabb_max, 16);
assert_offset!(PtMesh, ihead, 28);
assert_offset!(PtMesh, vhead, 32);
assert_offset!(PtMesh, tri_count, 36);
assert_offset!(PtVertex, normal, 16);
assert_offset!(PtVertex, uv, 32);
assert_offset!(PtMaterial, albedo_texture, 12);
assert_offset!(PtMaterial, emissive, 16);
assert_offset!(PtMaterial, emissive_texture, 28);
assert_offset!(PtMaterial, roughness, 32);
assert_offset!(PtMaterial, metallic, 36);
assert_offset!(PtMaterial, metallic_roughness_texture, 40);
assert_offset!(PtMaterial, reflectance, 44);
assert_offset!(PtMaterial, normal_map_texture, 48);
assert_offset!(PtSettings, sky_color, 16);
assert_offset!(PtFrame, view_from_clip, 64);
assert_offset!(PtFrame, width, 128);
assert_offset!(PtFrame, height, 132);
assert_offset!(PtFrame, time, 136);
assert_offset!(PtFrame, flags, 140);
assert_offset!(PtFrame, settings, 144);
assert_offset!(PtFrame, part_index, 176);
assert_offset!(PtFrame, part_count, 180);
assert_offset!(PtHit, t, 12);
assert_offset!(PtHit, p, 16);
assert_offset!(PtHit, n, 28);
assert_offset!(PtHit, uv, 40);
assert_size!(PtMultiConfig, 104);
assert_offset!(PtMultiConfig, devices, 8);
assert_offset!(PtMultiConfig, mode, 72);

#[cfg(test)]
mod layout {
    use super::*;

    #[test]
    fn strides_match_the_plugins_storage_buffers() {
        // encase strides of src/data.rs: Object 144, GpuMesh 48, GpuVertex 48, Material 64, Texture 16
        assert_eq!(std::mem::size_of::<PtObject>(), 144);
        assert_eq!(std::mem::size_of::<PtMesh>(), 48);
        assert_eq!(std::mem::size_of::<PtVertex>(), 48);
        assert_eq!(std::mem::size_of::<PtMaterial>(), 64);
        assert_eq!(std::mem::size_of::<PtTexture>(), 16);
        assert_eq!(std::mem::align_of::<PtFrame>(), 4);
    }
}

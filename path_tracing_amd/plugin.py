"""Host-side mirror of the reference plugin's surface for its render pass.

Same names and argument meaning as the Rust side (reference src/lib.rs:1-7, src/shader.rs:35-213,
src/data.rs:15-97) so that tests read like code written against the plugin:

    plugin = RayTracePlugin(); plugin.build(); plugin.finish()
    plugin.meta.set_meshes(...); plugin.meta.set_materials(...); plugin.meta.set_objects(...)
    image = plugin.node.run(view, globals_, settings)

`RayTraceNode.run` does what the reference's node does per view per frame, with the wgpu render pass
replaced by the C ABI (include/pt_amd.h). Like the reference it draws nothing (returns None) while the
plugin is not ready; unlike the reference, which panics on unusable buffers (src/shader.rs:168-187), it
raises `abi.PtError` carrying the library's error code.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import abi


@dataclass
class RayTraceSettings:
    """reference src/data.rs:15-20 (`#[derive(Default)]`: everything zero)."""
    bounces: int = 0
    samples: int = 0
    sky_color: tuple = (0.0, 0.0, 0.0, 0.0)   # LinearRgba


@dataclass
class View:
    """The fields of bevy_render::view::View the shader reads (reference src/raytrace.wgsl:210-217)."""
    world_from_view: np.ndarray   # 16 floats, column-major
    view_from_clip: np.ndarray    # 16 floats, column-major
    viewport: tuple               # (x, y, width, height)


@dataclass
class Globals:
    """bevy_render::globals::Globals; only `time` is read (reference src/raytrace.wgsl:210)."""
    time: float = 0.0


@dataclass
class RayTraceMeta:
    """reference src/data.rs:83-97: the storage buffers the extract systems fill. `emissives` is kept for
    source compatibility; the shader binds but never reads it (src/query.wgsl:9), so it is not uploaded."""
    objects: np.ndarray = field(default_factory=lambda: np.zeros(0, abi.OBJECT_DT))
    emissives: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint32))
    meshes: np.ndarray = field(default_factory=lambda: np.zeros(0, abi.MESH_DT))
    indices: np.ndarray = field(default_factory=lambda: np.zeros(0, np.uint32))
    vertices: np.ndarray = field(default_factory=lambda: np.zeros(0, abi.VERTEX_DT))
    materials: np.ndarray = field(default_factory=lambda: np.zeros(0, abi.MATERIAL_DT))
    textures: np.ndarray = field(default_factory=lambda: np.zeros(0, abi.TEXTURE_DT))
    texture_data: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))
    _dirty: set = field(default_factory=lambda: {"meshes", "materials", "objects"})

    # what prepare_meshes / extract_materials+extract_textures / extract_visible write (src/extract.rs)
    def set_meshes(self, meshes, indices, vertices):
        self.meshes, self.indices, self.vertices = meshes, indices, vertices
        self._dirty.add("meshes")

    def set_materials(self, materials, textures, texture_data):
        self.materials, self.textures, self.texture_data = materials, textures, texture_data
        self._dirty.add("materials")

    def set_images(self, materials, images):
        """extract_textures (reference src/extract.rs:257-295): flatten every image asset, in order, with
        TextureData::append_texture. `images` is a sequence of (data, width, height, format_name)."""
        from .textures import TextureData
        td = TextureData()
        for data, width, height, fmt in images:
            td.append_texture(data, width, height, fmt)
        self.set_materials(materials, td.texture_array(), td.data)

    def set_objects(self, objects, emissives=None):
        self.objects = objects
        if emissives is not None:
            self.emissives = emissives
        self._dirty.add("objects")

    def set_scene(self, scene):
        self.set_meshes(scene.meshes, scene.indices, scene.vertices)
        self.set_materials(scene.materials, scene.textures, scene.texture_data)
        self.set_objects(scene.objects)


class RayTraceNode:
    """reference src/shader.rs:111-214 (`impl ViewNode for RayTraceNode`)."""

    def __init__(self, plugin: "RayTracePlugin"):
        self._plugin = plugin

    def run(self, view: View, globals_: Globals, settings: RayTraceSettings | None, *, flags: int = 0,
            part_index: int = 0, part_count: int = 1):
        p = self._plugin
        # reference :133-145: pipeline / uniforms not ready -> Ok(()) without drawing
        if p.ctx is None or settings is None or view is None or globals_ is None:
            return None
        p.prepare()
        w, h = int(view.viewport[2]), int(view.viewport[3])
        frame = abi.make_frame(view.world_from_view, view.view_from_clip, w, h, time=globals_.time,
                               bounces=settings.bounces, samples=settings.samples, sky_color=tuple(settings.sky_color)[:4],
                               flags=flags, part_index=part_index, part_count=part_count)
        return p.ctx.render(frame)

    def submit(self, view: View, globals_: Globals, settings: RayTraceSettings | None, slot: int, target: np.ndarray, *, flags: int = 0) -> bool:
        """The same pass with frames in flight (pt_render_submit): enqueue the view's frame on `slot` and return; `target`
        (from plugin.ctx.host_array) holds the image once wait(slot) has returned. Returns False when nothing was enqueued
        (plugin not ready, camera without settings), as `run` returns None then."""
        p = self._plugin
        if p.ctx is None or settings is None or view is None or globals_ is None:
            return False
        p.prepare()    # (an object upload here does not wait for the frames in flight: they keep their version)
        w, h = int(view.viewport[2]), int(view.viewport[3])
        frame = abi.make_frame(view.world_from_view, view.view_from_clip, w, h, time=globals_.time,
                               bounces=settings.bounces, samples=settings.samples, sky_color=tuple(settings.sky_color)[:4], flags=flags)
        p.ctx.submit(frame, slot, target)
        return True

    def wait(self, slot: int):
        self._plugin.ctx.wait(slot)

    def submit_device(self, view: View, globals_: Globals, settings: RayTraceSettings | None, slot: int, device_ptr: int, *, flags: int = 0) -> bool:
        """`submit` with the image left in device memory at `device_ptr`, for a renderer whose later passes read it on the GPU
        (the reference's node sits between EndMainPass and MotionBlur, src/shader.rs:83-88)."""
        p = self._plugin
        if p.ctx is None or settings is None or view is None or globals_ is None:
            return False
        p.prepare()
        w, h = int(view.viewport[2]), int(view.viewport[3])
        frame = abi.make_frame(view.world_from_view, view.view_from_clip, w, h, time=globals_.time,
                               bounces=settings.bounces, samples=settings.samples, sky_color=tuple(settings.sky_color)[:4], flags=flags)
        p.ctx.submit_device(frame, slot, device_ptr)
        return True

    def stream_wait(self, slot: int, stream: int = 0):
        """Work enqueued on the HIP stream `stream` from now on starts when the slot's frame is complete (no host wait)."""
        self._plugin.ctx.stream_wait(slot, stream)

    def slot_wait_for(self, slot: int, stream: int = 0):
        """What is submitted to `slot` from now on starts when the work enqueued on `stream` so far is complete."""
        self._plugin.ctx.slot_wait_for(slot, stream)


class RayTracePlugin:
    """reference src/shader.rs:35-104 (`impl Plugin for RayTracePlugin`)."""

    def __init__(self, device: int = 0, accel: int = abi.PT_ACCEL_AUTO, tie_rule: int = abi.PT_TIE_REFERENCE):
        self._cfg = (device, accel, tie_rule)
        self.meta: RayTraceMeta | None = None
        self.ctx: abi.Context | None = None
        self.node = RayTraceNode(self)

    def build(self):
        """reference :38-95: registers the resources (RayTraceMeta with empty buffers)."""
        self.meta = RayTraceMeta()
        return self

    def finish(self):
        """reference :97-103: creates the pipeline; here, the device context."""
        if self.meta is None:
            raise RuntimeError("RayTracePlugin.finish() before build()")
        device, accel, tie = self._cfg
        self.ctx = abi.Context(device=device, accel=accel, tie_rule=tie)
        return self

    def prepare(self):
        """Upload what changed since the last frame (the Render-schedule `prepare_meshes` step and the
        `write_buffer` calls of the extract systems, reference src/extract.rs:102-147, :249-292, :337-345)."""
        m = self.meta
        if "meshes" in m._dirty:
            self.ctx.upload_meshes(m.meshes, m.indices, m.vertices)
        if "materials" in m._dirty:
            self.ctx.upload_materials(m.materials, m.textures, m.texture_data)
        if "objects" in m._dirty:
            self.ctx.upload_objects(m.objects)
        m._dirty.clear()

    def close(self):
        if self.ctx is not None:
            self.ctx.close()
            self.ctx = None

"""Host-side mirror of the reference's `class RVPT` for the path this backend replaces.

Same names, argument meaning and frame-counter behaviour as src/rvpt/rvpt.{h,cpp}:
  add_material / add_triangle      rvpt.cpp:1041-1043
  initialize                        rvpt.cpp:56-94   (BVH build + permute, rvpt.cpp:83-86; resource creation)
  update                            rvpt.cpp:96-126  (accumulate-or-reset rule :102-111, uniform upload)
  draw                              rvpt.cpp:346-354 (record + submit of the compute pass)
  shutdown                          rvpt.cpp:407-442
Presentation (swapchain blit, ImGui, debug raster) is out of scope; read_frame() replaces the blit as the
way to get pixels out.  All GPU work goes through the C ABI (rvpt_amd.native); there is no CPU fallback.
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import native
from .camera import Camera


@dataclass
class RenderSettings:
    """rvpt.h:77-89 (defaults included: current_frame starts at 1 and is reset by the first update())."""
    max_bounces: int = 8
    aa: int = 1
    current_frame: int = 1
    camera_mode: int = 0
    top_left_render_mode: int = 9
    top_right_render_mode: int = 9
    bottom_left_render_mode: int = 9
    bottom_right_render_mode: int = 9
    split_ratio: tuple = field(default_factory=lambda: (0.5, 0.5))

    def pack(self) -> np.ndarray:
        """The 40-byte std140 block (compute_pass.comp:28-40)."""
        s = np.zeros(10, dtype=np.int32)
        s[0], s[1] = self.max_bounces, self.aa
        s.view(np.uint32)[2] = self.current_frame & 0xFFFFFFFF
        s[3] = self.camera_mode
        s[4:8] = (self.top_left_render_mode, self.top_right_render_mode, self.bottom_left_render_mode,
                  self.bottom_right_render_mode)
        s.view(np.float32)[8:10] = self.split_ratio
        return s

    def _reset_key(self):
        # PreviousFrameState::operator== (rvpt.cpp:21-29): aa and max_bounces are NOT part of it
        return ((float(np.float32(self.split_ratio[0])), float(np.float32(self.split_ratio[1]))), self.top_left_render_mode, self.top_right_render_mode,
                self.bottom_left_render_mode, self.bottom_right_render_mode, self.camera_mode)


def launch_sizes(frames: int, batch: int, in_flight: int = 3) -> list[int]:
    """How a run of `frames` accumulation frames with a still camera goes out: as few launches as `batch` allows, of near-equal size
    (20 frames at batch 8: 7 + 7 + 6; at batch 64: one launch of 20).  Measured on MI355X with the packet kernel
    (tools/archive/sweep_launch_shapes.sh, profiles/r03_launch_shapes.txt): on one GPU the shape of a 20-frame run does not matter (0.253-0.260 ms
    per frame for 20 / 10+10 / 7+7+6 / 5x4 / 4x5); on an eighth of the image ONE launch is best (0.0366 ms per frame against 0.0437 for
    7+7+6): a work-group of that kernel fills its CU's LDS share, launches in flight do not overlap, and every extra launch is an
    extra ramp and tail.  (Round 3 briefly forced at least `in_flight` launches — VERDICT r2 #2 — which cost the 8-rank case 19 %;
    the parameter is kept for callers and ignored.)  rvpt_host.cpp::launch_sizes is the same rule."""
    if frames <= 0:
        return []
    n = -(-frames // max(batch, 1))
    base, extra = divmod(frames, n)
    return [base + 1] * extra + [base] * (n - extra)


class RVPT:
    def __init__(self, width: int, height: int, device: int = 0, traversal: str = "brute", tile_rank: int = 0,
                 tile_world: int = 1, flags: int = 0):
        if traversal not in ("brute", "bvh", "bvh_ordered"):
            raise ValueError("traversal must be 'brute', 'bvh' (the reference's visiting order) or 'bvh_ordered'")
        self.width, self.height = int(width), int(height)
        self.device, self.traversal = device, traversal
        self.tile_rank, self.tile_world = tile_rank, tile_world
        self._flags = flags | {"brute": native.TRAVERSAL_BRUTE, "bvh": native.TRAVERSAL_BVH,
                               "bvh_ordered": native.TRAVERSAL_BVH_ORDERED}[traversal]
        self.scene_camera = Camera(self.width / self.height)  # Window::get_aspect_ratio, window.cpp:89-92
        self.render_settings = RenderSettings()
        self.triangles: list[np.ndarray] = []
        self.materials: list[np.ndarray] = []
        self.bvh_nodes: np.ndarray | None = None
        self.primitive_indices: np.ndarray | None = None
        self.sorted_triangles: np.ndarray | None = None
        self._previous_key = None  # default-constructed PreviousFrameState never compares equal (empty camera data)
        self._ctx: native.Context | None = None

    # -- scene -------------------------------------------------------------------------------------------
    def add_material(self, material) -> None:
        self.materials.append(np.asarray(material, dtype=np.float32).reshape(12))

    def add_triangle(self, triangle) -> None:
        self.triangles.append(np.asarray(triangle, dtype=np.float32).reshape(1, 16))

    def add_triangles(self, triangles) -> None:
        self.triangles.append(np.asarray(triangles, dtype=np.float32).reshape(-1, 16))

    # -- lifecycle -----------------------------------------------------------------------------------------
    def initialize(self) -> bool:
        tris = np.concatenate(self.triangles) if self.triangles else np.zeros((0, 16), np.float32)
        mats = np.stack(self.materials) if self.materials else np.zeros((0, 12), np.float32)
        if tris.shape[0]:
            # top_level_bvh = bvh_builder.build_bvh(triangles); sorted_triangles = permute_primitives (rvpt.cpp:83-86)
            self.bvh_nodes, self.primitive_indices = native.build_bvh(tris)
            self.sorted_triangles = tris[self.primitive_indices]
        else:
            self.bvh_nodes, self.primitive_indices, self.sorted_triangles = None, np.zeros(0, np.uint32), tris
        self._ctx = native.Context(self.width, self.height, self.device, self.tile_rank, self.tile_world, self._flags)
        self._ctx.upload_scene(self.bvh_nodes if self.traversal != "brute" else None, self.sorted_triangles, mats)
        return True

    def update(self) -> bool:
        camera_data = self.scene_camera.get_data()
        rs = self.render_settings
        rs.camera_mode = self.scene_camera.mode
        # PreviousFrameState comparison (rvpt.cpp:21-29,102-111); camera_data is cached by the camera, so an
        # identity check short-cuts the byte comparison on the steady accumulate path
        # (compared by VALUE, float32 like the uniform block: a caller mutating rs.split_ratio in place must reset too)
        key = rs._reset_key()
        prev = self._previous_key
        same = prev is not None and prev[0] == key and (prev[1] is camera_data or prev[2] == camera_data.tobytes())
        if not same:
            rs.current_frame = 0
            self._previous_key = (key, camera_data, camera_data.tobytes())
        else:
            rs.current_frame += 1
        self._ctx.set_frame_fast(rs, camera_data)
        return True

    def draw(self) -> None:
        self._ctx.dispatch()

    def draw_frames(self, n_frames: int) -> None:
        """update(); draw_frames(n) == n x (update(); draw()) with nothing changed in between: the frames
        current_frame .. current_frame + n - 1 go out as one launch (rvpt_hip_dispatch_frames) and the frame
        counter moves on, so the next update() continues the accumulation."""
        self._ctx.dispatch_frames(n_frames)
        self.render_settings.current_frame += n_frames - 1

    def wait(self) -> None:
        self._ctx.wait()

    def read_frame(self, fmt: int = native.FORMAT_RGBA32F) -> np.ndarray:
        return self._ctx.read(fmt)

    def shutdown(self) -> None:
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None

    @property
    def context(self) -> native.Context:
        return self._ctx

// rvpt_host.h — C++ host layer above the C ABI (include/rvpt_hip.h), mirroring the part of the reference's
// host interface that feeds the compute pass.  Same names, argument meaning and frame-counter behaviour as
// (paths relative to the reference tree):
//
//   Triangle, AABB-less          src/rvpt/geometry.h:76-111   (4 x vec4, face normal packed into the .w lanes)
//   Material                     src/rvpt/material.h:9-26
//   Camera                       src/rvpt/camera.h:14-57, camera.cpp:17-66  (T * R(UP,rx) * R(RIGHT,ry) * R(FORWARD,rz))
//   RVPT::RenderSettings         src/rvpt/rvpt.h:77-89
//   RVPT::add_material/add_triangle/initialize/update/draw/shutdown   src/rvpt/rvpt.{h,cpp}
//   load_model                   src/rvpt/main.cpp:12-62   (OBJ positions only, constant material id)
//
// Written from scratch (no glm, no tinyobjloader, no Vulkan): a few lines of vector maths and a minimal OBJ
// reader are all this path needs.  Presentation (window, swapchain blit, ImGui, debug raster) is out of scope;
// read_frame() replaces the blit.  Every GPU call goes through the C ABI; there is no CPU fallback.
#pragma once

#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/rvpt_hip.h"

namespace rvpt {

struct vec3 {
    float x = 0, y = 0, z = 0;
};

// geometry.h:76-111
struct Triangle {
    Triangle() = default;
    Triangle(const vec3 &v0, const vec3 &v1, const vec3 &v2, int material_id);
    float vertex0[4] = {}, vertex1[4] = {}, vertex2[4] = {}, material_id[4] = {};
};
static_assert(sizeof(Triangle) == sizeof(rvpt_triangle), "Triangle is uploaded as is");

// material.h:9-26 (albedo[3] doubles as the index of refraction, intersection.glsl:54)
struct Material {
    enum class Type { LAMBERT, MIRROR, DIELECTRIC };
    Material() = default;
    Material(const std::array<float, 4> &albedo, const std::array<float, 4> &emission, Type type);
    float albedo[4] = {}, emission[4] = {}, data[4] = {};
};
static_assert(sizeof(Material) == sizeof(rvpt_material), "Material is uploaded as is");

// camera.h:14-57
class Camera {
public:
    explicit Camera(float aspect);
    void translate(const vec3 &in_translation);  // camera.cpp:29-33: moves along the camera's own axes
    void rotate(const vec3 &in_rotation);        // degrees; .y clamped to +-90 when the clamp is on
    void set_fov(float in_fov);
    void set_scale(float in_scale);
    void set_camera_mode(int in_mode);
    void clamp_vertical_view_angle(bool clamp);
    float get_fov() const noexcept { return fov_; }
    float get_scale() const noexcept { return scale_; }
    int get_camera_mode() const noexcept { return mode_; }
    // camera.cpp:55-66: 4 matrix columns then (aspect, radians(fov), scale, 0)
    rvpt_camera_data get_data() const;
    vec3 translation{}, rotation{};

private:
    int mode_ = 0;
    float fov_ = 90.f, scale_ = 4.f, aspect_ = 1.f;
    bool vertical_view_angle_clamp_ = false;
};

// rvpt.h:77-89 — field order == the 40-byte uniform block
struct RenderSettings {
    int max_bounces = 8;
    int aa = 1;
    uint32_t current_frame = 1;
    int camera_mode = 0;
    int top_left_render_mode = 9;
    int top_right_render_mode = 9;
    int bottom_left_render_mode = 9;
    int bottom_right_render_mode = 9;
    float split_ratio[2] = {0.5f, 0.5f};
};
static_assert(sizeof(RenderSettings) == sizeof(rvpt_render_settings), "RenderSettings is uploaded as is");

// The C ABI as a table of entry points, so that the frame-counter logic can be exercised against a recording
// fake (host_selftest.cpp) without a GPU.  Backend::native() binds include/rvpt_hip.h.
struct Backend {
    int (*create)(rvpt_hip_ctx **, int, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t);
    void (*destroy)(rvpt_hip_ctx *);
    int (*upload_scene)(rvpt_hip_ctx *, const rvpt_bvh_node *, size_t, const rvpt_triangle *, size_t, const rvpt_material *, size_t);
    int (*set_frame)(rvpt_hip_ctx *, const rvpt_render_settings *, const rvpt_camera_data *);
    int (*dispatch)(rvpt_hip_ctx *);
    int (*dispatch_frames)(rvpt_hip_ctx *, uint32_t);
    int (*wait)(rvpt_hip_ctx *);
    int (*read)(rvpt_hip_ctx *, int, void *, size_t);
    const char *(*last_error)(rvpt_hip_ctx *);
    int (*bvh_build)(const rvpt_triangle *, size_t, rvpt_bvh_node *, size_t *, uint32_t *);
    static const Backend &native();
};

class RVPT {
public:
    struct Options {
        int device = 0;
        bool bvh_traversal = true;  // the reference's live path; false = LDS-staged brute force
        bool ordered_children = false;  // with bvh_traversal: nearer child first (RVPT_HIP_TRAVERSAL_BVH_ORDERED)
        uint32_t tile_rank = 0, tile_world = 1;
        uint32_t extra_flags = 0;   // RVPT_HIP_TIMING, RVPT_HIP_ACCUM_UNORM8, ...
    };
    RVPT(uint32_t width, uint32_t height);
    RVPT(uint32_t width, uint32_t height, const Options &options, const Backend &backend = Backend::native());
    ~RVPT();
    RVPT(const RVPT &) = delete;
    RVPT &operator=(const RVPT &) = delete;

    bool initialize();  // rvpt.cpp:56-94: BVH build + permute (:83-86), resource creation, scene upload
    bool update();      // rvpt.cpp:96-126: accumulate-or-reset rule (:102-111), uniform upload
    void draw();        // rvpt.cpp:346-354: asynchronous dispatch of the compute pass
    // update(); draw_frames(n) == n x (update(); draw()) with nothing changed in between: the n accumulation frames go out
    // as one launch (rvpt_hip_dispatch_frames) and the frame counter moves on by n
    void draw_frames(uint32_t n_frames);
    void wait();        // raytrace_work_fence.wait() (rvpt.cpp:115); only needed before reading results
    void shutdown();    // rvpt.cpp:407-442

    void add_material(Material material);  // rvpt.cpp:1041
    void add_triangle(Triangle triangle);  // rvpt.cpp:1043

    // RGBA32F (width*height*4 floats) or RGBA8 (width*height*4 bytes), row-major, top row first
    std::vector<float> read_frame();
    std::vector<uint8_t> read_frame_rgba8();
    const std::string &last_error() const { return error_; }
    // the backend context, e.g. to join several RVPT objects (one per GPU, tile_rank i of n) into one RCCL group with
    // rvpt_hip_comm_init_all; read_frame() on rank 0's object is then the gather of the whole image
    rvpt_hip_ctx *context() const { return ctx_; }

    Camera scene_camera;
    RenderSettings render_settings;

    // what initialize() derived (rvpt.h:175-179: top_level_bvh, sorted_triangles)
    const std::vector<rvpt_bvh_node> &bvh_nodes() const { return nodes_; }
    const std::vector<Triangle> &sorted_triangles() const { return sorted_; }
    const std::vector<Material> &materials() const { return materials_; }

private:
    bool check(int rc, const char *what);
    uint32_t width_, height_;
    Options options_;
    const Backend &backend_;
    rvpt_hip_ctx *ctx_ = nullptr;
    std::vector<Triangle> triangles_, sorted_;
    std::vector<Material> materials_;
    std::vector<rvpt_bvh_node> nodes_;
    // PreviousFrameState (rvpt.h:211-219, rvpt.cpp:21-29); empty camera data never compares equal
    struct Previous {
        bool valid = false;
        RenderSettings settings;
        rvpt_camera_data camera;
    } previous_;
    std::string error_;
};

// main.cpp:12-62: every triangular face of every shape becomes a Triangle with `material_id`; polygons are
// fan-triangulated (tinyobjloader's default), normals / uvs / materials are ignored.  Returns triangles added, -1 on error.
long load_model(RVPT &rvpt, const std::string &path, int material_id, std::string *error = nullptr);

// Scene description beyond the reference's loader (which ignores materials, main.cpp:49-59): OBJ + MTL.  `mtllib` files are
// read relative to the OBJ, `usemtl` selects the material of the faces that follow; material ids are handed out in order
// of first use (offset by the materials the RVPT already holds); faces before any `usemtl`, or naming an undefined
// material, get a white Lambert material.  MTL mapping: Kd -> albedo, Ke -> emission, Ni -> index of refraction
// (albedo.w); illum 3/8 -> MIRROR (albedo Ks when given), illum 4/6/7/9 or d < 1 -> DIELECTRIC, else LAMBERT.
// Returns triangles added, -1 on error.  Same rules as rvpt_amd.scene.load_obj_scene.
long load_scene(RVPT &rvpt, const std::string &path, std::string *error = nullptr);

// main.cpp:102-107: the demo scene's two Lambert materials
void add_default_materials(RVPT &rvpt);

// How a run of `frames` accumulation frames with a still camera goes out (draw_frames): as few launches as `batch` allows, of
// near-equal size.  20 frames at batch 8: {7, 7, 6}; at batch 64: {20} (on a small tile share one launch beats three by 19 %,
// profiles/r03_launch_shapes.txt).  `in_flight` is ignored.  Same rule as rvpt_amd.renderer.launch_sizes.
std::vector<uint32_t> launch_sizes(uint32_t frames, uint32_t batch, uint32_t in_flight = 3);

}  // namespace rvpt

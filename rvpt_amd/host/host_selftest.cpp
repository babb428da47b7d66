// host_selftest — GPU-free checks of the C++ host layer against a recording fake of the C ABI.
// Exit code 0 and a final "host_selftest ok" line on success (run by tests/test_cpp_host.py).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <vector>

#include "rvpt_host.h"

namespace {

struct Recorded {
    uint32_t frame;
    int aa, bounces, camera_mode;
    rvpt_camera_data cam;
};
std::vector<Recorded> g_frames;
size_t g_uploaded_tris = 0, g_uploaded_nodes = 0, g_uploaded_mats = 0;
uint32_t g_flags = 0;
int g_dispatches = 0;
int g_fail = 0;

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::printf("CHECK failed at line %d: %s\n", __LINE__, #cond); \
            ++g_fail;                                                      \
        }                                                                  \
    } while (0)

int f_create(rvpt_hip_ctx **out, int, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t flags)
{
    g_flags = flags;
    *out = reinterpret_cast<rvpt_hip_ctx *>(0x1);
    return 0;
}
void f_destroy(rvpt_hip_ctx *) {}
int f_upload(rvpt_hip_ctx *, const rvpt_bvh_node *nodes, size_t n_nodes, const rvpt_triangle *, size_t n_tris, const rvpt_material *, size_t n_mats)
{
    g_uploaded_nodes = nodes ? n_nodes : 0;
    g_uploaded_tris = n_tris;
    g_uploaded_mats = n_mats;
    return 0;
}
int f_set_frame(rvpt_hip_ctx *, const rvpt_render_settings *s, const rvpt_camera_data *c)
{
    g_frames.push_back({s->current_frame, s->aa, s->max_bounces, s->camera_mode, *c});
    return 0;
}
int f_dispatch(rvpt_hip_ctx *) { ++g_dispatches; return 0; }
int g_batched_frames = 0;
int f_dispatch_frames(rvpt_hip_ctx *, uint32_t n) { g_batched_frames += static_cast<int>(n); return 0; }
int f_wait(rvpt_hip_ctx *) { return 0; }
int f_read(rvpt_hip_ctx *, int, void *, size_t) { return 0; }
const char *f_err(rvpt_hip_ctx *) { return ""; }

}  // namespace

int main(int argc, char **argv)
{
    using namespace rvpt;
    // the BVH builder is host code in librvpt_hip.so and needs no GPU: use the real one
    const Backend fake{f_create, f_destroy, f_upload, f_set_frame, f_dispatch, f_dispatch_frames, f_wait, f_read, f_err, rvpt_bvh_build};

    // Triangle: face normal rides in the .w lanes, material id as a float (geometry.h:81-91)
    const Triangle t({0, 0, 0}, {1, 0, 0}, {0, 1, 0}, 3);
    CHECK(t.vertex0[3] == 0.f && t.vertex1[3] == 0.f && t.vertex2[3] == 1.f && t.material_id[0] == 3.f);
    CHECK(t.vertex1[0] == 1.f && t.vertex2[1] == 1.f);
    // Material: data.x = type, albedo.w carries the ior (material.h:17-25)
    const Material m({0.5f, 0.6f, 0.7f, 1.5f}, {1, 2, 3, 0}, Material::Type::DIELECTRIC);
    CHECK(m.data[0] == 2.f && m.albedo[3] == 1.5f && m.emission[1] == 2.f);

    // Camera: defaults (camera.h:44-46) give identity + (aspect, pi/2, 4, 0)
    Camera cam(2.0f);
    rvpt_camera_data d = cam.get_data();
    for (int i = 0; i < 16; ++i) CHECK(d.matrix[i] == ((i % 5 == 0) ? 1.f : 0.f));
    CHECK(d.params[0] == 2.f && std::fabs(d.params[1] - 1.57079633f) < 1e-6f && d.params[2] == 4.f && d.params[3] == 0.f);
    cam.translate({0, 0, 1});
    d = cam.get_data();
    CHECK(d.matrix[12] == 0.f && d.matrix[13] == 0.f && d.matrix[14] == 1.f);
    cam.rotate({90.f, 0, 0});  // about UP: forward (+Z) turns towards +X (column 2 of the matrix)
    d = cam.get_data();
    CHECK(std::fabs(d.matrix[8] - 1.f) < 1e-6f && std::fabs(d.matrix[10]) < 1e-6f);
    cam.translate({0, 0, 1});  // moves along the camera's own forward axis = world +X now
    CHECK(std::fabs(cam.translation.x - 1.f) < 1e-6f && std::fabs(cam.translation.z - 1.f) < 1e-6f);
    cam.clamp_vertical_view_angle(true);
    cam.rotate({0, 200.f, 0});
    CHECK(cam.rotation.y == 90.f);

    // construct_camera_matrix (camera.cpp:17-25) = T * Ry(rot.x) * Rx(rot.y) * Rz(rot.z), degrees, right-handed, column-major:
    // checked against the product written out by hand (a = rot.x about UP, b = rot.y about RIGHT, c = rot.z about FORWARD)
    //   [ ca*cc + sa*sb*sc   -ca*sc + sa*sb*cc   sa*cb ]
    //   [ cb*sc               cb*cc              -sb   ]
    //   [ -sa*cc + ca*sb*sc   sa*sc + ca*sb*cc   ca*cb ]
    {
        const double poses[3][6] = {{1, 2, 3, 30, -20, 10}, {-0.5, 0.25, 4, -135, 60, -75}, {0, 0, 0, 90, 90, 0}};
        for (const auto &p : poses) {
            Camera k(1.5f);
            k.translation = {static_cast<float>(p[0]), static_cast<float>(p[1]), static_cast<float>(p[2])};
            k.rotation = {static_cast<float>(p[3]), static_cast<float>(p[4]), static_cast<float>(p[5])};
            const double rad = 3.14159265358979323846 / 180.0;
            const double ca = std::cos(p[3] * rad), sa = std::sin(p[3] * rad), cb = std::cos(p[4] * rad), sb = std::sin(p[4] * rad),
                         cc = std::cos(p[5] * rad), sc = std::sin(p[5] * rad);
            const double want[16] = {ca * cc + sa * sb * sc, cb * sc, -sa * cc + ca * sb * sc, 0,   // column 0
                                     -ca * sc + sa * sb * cc, cb * cc, sa * sc + ca * sb * cc, 0,  // column 1
                                     sa * cb, -sb, ca * cb, 0,                                     // column 2
                                     p[0], p[1], p[2], 1};                                         // column 3
            const rvpt_camera_data kd = k.get_data();
            for (int i = 0; i < 16; ++i) CHECK(std::fabs(kd.matrix[i] - want[i]) < 2e-6);
        }
        // rot = (90, 90, 0): forward -> -Y, up -> +X, right -> -Z (worked by hand from the two quarter turns)
        Camera k(1.f);
        k.rotation = {90.f, 90.f, 0.f};
        const rvpt_camera_data kd = k.get_data();
        CHECK(std::fabs(kd.matrix[9] + 1.f) < 1e-6f && std::fabs(kd.matrix[4] - 1.f) < 1e-6f && std::fabs(kd.matrix[2] + 1.f) < 1e-6f);
    }

    // load_model on a small OBJ (quad + negative indices + v/vt/vn forms), then the frame-counter rule
    const std::string obj = (argc > 1 ? std::string(argv[1]) : std::string("/tmp")) + "/host_selftest.obj";
    {
        std::ofstream f(obj);
        f << "# test\nv 0 0 2\nv 1 0 2\nv 1 1 2\nv 0 1 2\nvn 0 0 1\nf 1/1/1 2/1/1 3/1/1 4/1/1\nf -4//1 -3//1 -2//1\n";
    }
    RVPT::Options opt;
    opt.bvh_traversal = true;
    RVPT r(64, 32, opt, fake);
    std::string err;
    CHECK(load_model(r, obj, 1, &err) == 3);
    CHECK(load_model(r, obj + ".missing", 1, &err) == -1 && !err.empty());
    add_default_materials(r);
    CHECK(r.render_settings.current_frame == 1 && r.render_settings.max_bounces == 8 && r.render_settings.aa == 1);  // rvpt.h:77-89
    CHECK(r.initialize());
    CHECK(g_uploaded_tris == 3 && g_uploaded_mats == 2 && g_uploaded_nodes >= 1 && (g_flags & RVPT_HIP_TRAVERSAL_BVH));
    CHECK(r.sorted_triangles().size() == 3 && r.materials()[0].emission[2] == 0.6f);

    auto step = [&]() { r.update(); r.draw(); return r.render_settings.current_frame; };
    CHECK(step() == 0 && step() == 1 && step() == 2);
    r.render_settings.aa = 4;            // not part of PreviousFrameState (rvpt.cpp:21-29)
    r.render_settings.max_bounces = 3;
    CHECK(step() == 3);
    r.render_settings.bottom_left_render_mode = 5;
    CHECK(step() == 0 && step() == 1);
    r.render_settings.split_ratio[0] = 0.25f;
    CHECK(step() == 0);
    r.scene_camera.rotate({1.f, 0, 0});
    CHECK(step() == 0 && step() == 1);
    r.scene_camera.set_camera_mode(1);
    CHECK(step() == 0);
    r.scene_camera.set_fov(60.f);
    CHECK(step() == 0 && step() == 1);
    CHECK(g_dispatches == static_cast<int>(g_frames.size()) && g_frames.size() == 12);
    CHECK(g_frames[3].frame == 3 && g_frames[3].aa == 4 && g_frames[3].bounces == 3);
    CHECK(g_frames.back().camera_mode == 1 && std::fabs(g_frames.back().cam.params[1] - 1.04719755f) < 1e-6f);
    // batches of accumulation frames: one set_frame per batch, the counter moves on by the batch size
    r.update();
    r.draw_frames(5);
    CHECK(g_frames.back().frame == 2 && r.render_settings.current_frame == 6 && g_batched_frames == 5);
    CHECK(step() == 7);

    // launch_sizes: as few launches as `batch` allows, near-equal sizes
    {
        using V = std::vector<uint32_t>;
        CHECK(launch_sizes(20, 64) == (V{20}));
        CHECK(launch_sizes(20, 8) == (V{7, 7, 6}));
        CHECK(launch_sizes(200, 8) == V(25, 8));
        CHECK(launch_sizes(5, 64) == (V{5}));
        CHECK(launch_sizes(16, 1) == V(16, 1));
        CHECK(launch_sizes(0, 8).empty());
        CHECK(launch_sizes(20, 64, 6) == (V{20}));
        CHECK(launch_sizes(65, 64) == (V{33, 32}));
    }

    // load_scene: OBJ + MTL, ids in order of first use after the materials already present (rvpt_amd.scene.load_obj_scene's rules)
    {
        const std::string dir = (argc > 1 ? std::string(argv[1]) : std::string("/tmp"));
        {
            std::ofstream m(dir + "/host_selftest_scene.mtl");
            m << "newmtl wall\nKd 0.7 0.6 0.5\nillum 2\nnewmtl chrome\nKd 0.1 0.1 0.1\nKs 0.9 0.8 0.7\nillum 3\n"
                 "newmtl glass\nKd 1 1 1\nNi 1.33\nillum 7\nnewmtl lamp\nKd 0 0 0\nKe 4 5 6\n";
            std::ofstream o(dir + "/host_selftest_scene.obj");
            o << "mtllib host_selftest_scene.mtl\nv 0 0 1\nv 1 0 1\nv 1 1 1\nv 0 1 1\nf 1 2 3\nusemtl glass\nf 1 2 3 4\n"
                 "usemtl wall\nf 1 3 4\nusemtl nosuch\nf 2 3 4\nusemtl chrome\nf 1 2 4\nusemtl lamp\nf -4 -3 -2\n";
        }
        RVPT rs(32, 32, opt, fake);
        rs.add_material(Material({1, 1, 1, 0}, {0, 0, 0, 0}, Material::Type::LAMBERT));  // already present: ids start at 1
        CHECK(load_scene(rs, dir + "/host_selftest_scene.obj", &err) == 7);
        CHECK(rs.materials().size() == 6);  // + default, glass, wall, chrome, lamp
        CHECK(rs.materials()[2].data[0] == 2.f && rs.materials()[2].albedo[3] == 1.33f);                // glass
        CHECK(rs.materials()[4].data[0] == 1.f && rs.materials()[4].albedo[0] == 0.9f);                 // chrome: Ks
        CHECK(rs.materials()[5].emission[1] == 5.f && rs.materials()[3].albedo[1] == 0.6f);             // lamp, wall
        CHECK(load_scene(rs, dir + "/nope.obj", &err) == -1);
    }

    // brute-force contexts upload no nodes
    RVPT::Options bo;
    bo.bvh_traversal = false;
    RVPT rb(32, 32, bo, fake);
    rb.add_triangle(t);
    rb.add_material(m);
    CHECK(rb.initialize() && g_uploaded_nodes == 0 && !(g_flags & RVPT_HIP_TRAVERSAL_BVH));

    if (g_fail == 0) std::printf("host_selftest ok\n");
    return g_fail == 0 ? 0 : 1;
}

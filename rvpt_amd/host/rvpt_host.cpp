// rvpt_host.cpp — see rvpt_host.h.  Host-only C++17; links against librvpt_hip.so (the C ABI).
#include "rvpt_host.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace rvpt {

namespace {

constexpr double kPi = 3.14159265358979323846;

// 4x4 matrices in double, m[row][col]
struct mat4 {
    double m[4][4];
};
mat4 identity()
{
    mat4 r{};
    for (int i = 0; i < 4; ++i) r.m[i][i] = 1.0;
    return r;
}
mat4 mul(const mat4 &a, const mat4 &b)
{
    mat4 r{};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += a.m[i][k] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}
// rotation about a unit axis, right-handed (the convention of the reference's matrix library)
mat4 rotation(double angle, double x, double y, double z)
{
    const double c = std::cos(angle), s = std::sin(angle), t = 1.0 - c;
    mat4 r = identity();
    r.m[0][0] = c + t * x * x;
    r.m[0][1] = t * x * y - s * z;
    r.m[0][2] = t * x * z + s * y;
    r.m[1][0] = t * x * y + s * z;
    r.m[1][1] = c + t * y * y;
    r.m[1][2] = t * y * z - s * x;
    r.m[2][0] = t * x * z - s * y;
    r.m[2][1] = t * y * z + s * x;
    r.m[2][2] = c + t * z * z;
    return r;
}
// camera.cpp:17-25: translate, then rotate about UP by rotation.x, RIGHT by rotation.y, FORWARD by rotation.z
mat4 camera_matrix(const vec3 &translation, const vec3 &rotation_deg)
{
    mat4 m = identity();
    m.m[0][3] = translation.x;
    m.m[1][3] = translation.y;
    m.m[2][3] = translation.z;
    m = mul(m, rotation(rotation_deg.x * kPi / 180.0, 0, 1, 0));
    m = mul(m, rotation(rotation_deg.y * kPi / 180.0, 1, 0, 0));
    m = mul(m, rotation(rotation_deg.z * kPi / 180.0, 0, 0, 1));
    return m;
}

}  // namespace

// ---- geometry.h:81-91 -----------------------------------------------------------------------------------
Triangle::Triangle(const vec3 &v0, const vec3 &v1, const vec3 &v2, int mat)
{
    const float e0[3] = {v1.x - v0.x, v1.y - v0.y, v1.z - v0.z}, e1[3] = {v2.x - v0.x, v2.y - v0.y, v2.z - v0.z};
    float n[3] = {e0[1] * e1[2] - e0[2] * e1[1], e0[2] * e1[0] - e0[0] * e1[2], e0[0] * e1[1] - e0[1] * e1[0]};
    const float len = std::sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    for (float &c : n) c /= len;  // degenerate triangles get NaN, like the reference's normalize(0)
    const float a[4] = {v0.x, v0.y, v0.z, n[0]}, b[4] = {v1.x, v1.y, v1.z, n[1]}, c[4] = {v2.x, v2.y, v2.z, n[2]};
    std::memcpy(vertex0, a, sizeof a);
    std::memcpy(vertex1, b, sizeof b);
    std::memcpy(vertex2, c, sizeof c);
    material_id[0] = static_cast<float>(mat);
}

// ---- material.h:17-25 -----------------------------------------------------------------------------------
Material::Material(const std::array<float, 4> &alb, const std::array<float, 4> &emi, Type type)
{
    std::memcpy(albedo, alb.data(), sizeof albedo);
    std::memcpy(emission, emi.data(), sizeof emission);
    data[0] = static_cast<float>(static_cast<int>(type));
}

// ---- camera.cpp ------------------------------------------------------------------------------------------
Camera::Camera(float aspect) : aspect_(aspect) {}

void Camera::translate(const vec3 &d)
{
    const mat4 m = camera_matrix(translation, rotation);
    translation.x += static_cast<float>(m.m[0][0] * d.x + m.m[0][1] * d.y + m.m[0][2] * d.z);
    translation.y += static_cast<float>(m.m[1][0] * d.x + m.m[1][1] * d.y + m.m[1][2] * d.z);
    translation.z += static_cast<float>(m.m[2][0] * d.x + m.m[2][1] * d.y + m.m[2][2] * d.z);
}
void Camera::rotate(const vec3 &r)
{
    rotation.x += r.x;
    rotation.y += r.y;
    rotation.z += r.z;
    if (vertical_view_angle_clamp_) rotation.y = std::fmin(90.f, std::fmax(-90.f, rotation.y));
}
void Camera::set_fov(float f) { fov_ = f; }
void Camera::set_scale(float s) { scale_ = s; }
void Camera::set_camera_mode(int m) { mode_ = m; }
void Camera::clamp_vertical_view_angle(bool c) { vertical_view_angle_clamp_ = c; }

rvpt_camera_data Camera::get_data() const
{
    const mat4 m = camera_matrix(translation, rotation);
    rvpt_camera_data d{};
    for (int col = 0; col < 4; ++col)
        for (int row = 0; row < 4; ++row) d.matrix[4 * col + row] = static_cast<float>(m.m[row][col]);
    d.params[0] = aspect_;
    d.params[1] = static_cast<float>(fov_ * (kPi / 180.0));
    d.params[2] = scale_;
    d.params[3] = 0.f;
    return d;
}

// ---- the C ABI as a backend table -------------------------------------------------------------------------
const Backend &Backend::native()
{
    static const Backend b{rvpt_hip_create,          rvpt_hip_destroy, rvpt_hip_upload_scene, rvpt_hip_set_frame,  rvpt_hip_dispatch,
                           rvpt_hip_dispatch_frames, rvpt_hip_wait,    rvpt_hip_read,         rvpt_hip_last_error, rvpt_bvh_build};
    return b;
}

// ---- class RVPT -----------------------------------------------------------------------------------------
RVPT::RVPT(uint32_t width, uint32_t height) : RVPT(width, height, Options{}, Backend::native()) {}

RVPT::RVPT(uint32_t width, uint32_t height, const Options &options, const Backend &backend)
    : scene_camera(static_cast<float>(width) / static_cast<float>(height)),  // Window::get_aspect_ratio, window.cpp:89-92
      width_(width), height_(height), options_(options), backend_(backend)
{
}

RVPT::~RVPT() { shutdown(); }

bool RVPT::check(int rc, const char *what)
{
    if (rc == RVPT_HIP_OK) return true;
    const char *msg = backend_.last_error(ctx_);
    error_ = std::string(what) + " failed (" + std::to_string(rc) + "): " + (msg ? msg : "");
    std::fprintf(stderr, "[rvpt] %s\n", error_.c_str());  // fmt::print(stderr, ...) upstream
    return false;
}

void RVPT::add_material(Material material) { materials_.emplace_back(material); }
void RVPT::add_triangle(Triangle triangle) { triangles_.emplace_back(triangle); }

bool RVPT::initialize()
{
    // top_level_bvh = bvh_builder.build_bvh(triangles); sorted_triangles = permute_primitives(triangles)  (rvpt.cpp:83-86)
    sorted_.clear();
    nodes_.clear();
    if (!triangles_.empty()) {
        nodes_.resize(2 * triangles_.size() - 1);
        std::vector<uint32_t> order(triangles_.size());
        size_t n_nodes = 0;
        if (!check(backend_.bvh_build(reinterpret_cast<const rvpt_triangle *>(triangles_.data()), triangles_.size(), nodes_.data(), &n_nodes,
                                      order.data()),
                   "rvpt_bvh_build"))
            return false;
        nodes_.resize(n_nodes);
        sorted_.reserve(triangles_.size());
        for (uint32_t i : order) sorted_.push_back(triangles_[i]);
    }
    const uint32_t traversal = !options_.bvh_traversal ? RVPT_HIP_TRAVERSAL_BRUTE
                               : (options_.ordered_children ? RVPT_HIP_TRAVERSAL_BVH_ORDERED : RVPT_HIP_TRAVERSAL_BVH);
    const uint32_t flags = traversal | options_.extra_flags;
    if (!check(backend_.create(&ctx_, options_.device, width_, height_, options_.tile_rank, options_.tile_world, flags), "rvpt_hip_create"))
        return false;
    return check(backend_.upload_scene(ctx_, options_.bvh_traversal ? nodes_.data() : nullptr, options_.bvh_traversal ? nodes_.size() : 0,
                                       reinterpret_cast<const rvpt_triangle *>(sorted_.data()), sorted_.size(),
                                       reinterpret_cast<const rvpt_material *>(materials_.data()), materials_.size()),
                 "rvpt_hip_upload_scene");
}

bool RVPT::update()
{
    const rvpt_camera_data camera_data = scene_camera.get_data();
    render_settings.camera_mode = scene_camera.get_camera_mode();
    // PreviousFrameState::operator== (rvpt.cpp:21-29): split ratio, the four render modes, the camera mode and the
    // camera block — not aa, not max_bounces
    bool same = previous_.valid;
    const RenderSettings &a = previous_.settings, &b = render_settings;
    same = same && a.split_ratio[0] == b.split_ratio[0] && a.split_ratio[1] == b.split_ratio[1] &&
           a.top_left_render_mode == b.top_left_render_mode && a.top_right_render_mode == b.top_right_render_mode &&
           a.bottom_left_render_mode == b.bottom_left_render_mode && a.bottom_right_render_mode == b.bottom_right_render_mode &&
           a.camera_mode == b.camera_mode;
    for (int i = 0; same && i < 16; ++i) same = previous_.camera.matrix[i] == camera_data.matrix[i];
    for (int i = 0; same && i < 4; ++i) same = previous_.camera.params[i] == camera_data.params[i];
    if (!same) {  // rvpt.cpp:102-111
        render_settings.current_frame = 0;
        previous_.valid = true;
        previous_.settings = render_settings;
        previous_.camera = camera_data;
    } else {
        render_settings.current_frame++;
    }
    return check(backend_.set_frame(ctx_, reinterpret_cast<const rvpt_render_settings *>(&render_settings), &camera_data), "rvpt_hip_set_frame");
}

void RVPT::draw() { check(backend_.dispatch(ctx_), "rvpt_hip_dispatch"); }
void RVPT::draw_frames(uint32_t n_frames)
{
    if (check(backend_.dispatch_frames(ctx_, n_frames), "rvpt_hip_dispatch_frames")) render_settings.current_frame += n_frames - 1;
}
void RVPT::wait() { check(backend_.wait(ctx_), "rvpt_hip_wait"); }

void RVPT::shutdown()
{
    if (ctx_) backend_.destroy(ctx_);
    ctx_ = nullptr;
}

std::vector<float> RVPT::read_frame()
{
    std::vector<float> out(static_cast<size_t>(width_) * height_ * 4);
    if (!check(backend_.read(ctx_, RVPT_HIP_FORMAT_RGBA32F, out.data(), out.size() * sizeof(float)), "rvpt_hip_read")) out.clear();
    return out;
}
std::vector<uint8_t> RVPT::read_frame_rgba8()
{
    std::vector<uint8_t> out(static_cast<size_t>(width_) * height_ * 4);
    if (!check(backend_.read(ctx_, RVPT_HIP_FORMAT_RGBA8_UNORM, out.data(), out.size()), "rvpt_hip_read")) out.clear();
    return out;
}

// ---- main.cpp:12-62 ---------------------------------------------------------------------------------------
long load_model(RVPT &rvpt, const std::string &path, int material_id, std::string *error)
{
    std::ifstream in(path);
    if (!in) {
        if (error) *error = "cannot open " + path;
        return -1;
    }
    std::vector<vec3> verts;
    long added = 0;
    std::string line;
    while (std::getline(in, line)) {
        if (line.size() < 2) continue;
        std::istringstream ls(line);
        std::string tag;
        ls >> tag;
        if (tag == "v") {
            vec3 v;
            ls >> v.x >> v.y >> v.z;
            verts.push_back(v);
        } else if (tag == "f") {
            std::vector<long> idx;
            std::string tok;
            while (ls >> tok) {
                const long i = std::strtol(tok.c_str(), nullptr, 10);  // "a", "a/b", "a/b/c", "a//c": position index first
                idx.push_back(i > 0 ? i - 1 : static_cast<long>(verts.size()) + i);
            }
            for (size_t k = 1; k + 1 < idx.size(); ++k) {
                const long a = idx[0], b = idx[k], c = idx[k + 1];
                if (a < 0 || b < 0 || c < 0 || a >= static_cast<long>(verts.size()) || b >= static_cast<long>(verts.size()) ||
                    c >= static_cast<long>(verts.size())) {
                    if (error) *error = "face index out of range in " + path;
                    return -1;
                }
                rvpt.add_triangle(Triangle(verts[a], verts[b], verts[c], material_id));
                ++added;
            }
        }
    }
    return added;
}

namespace {

struct MtlEntry {
    float kd[3] = {0.8f, 0.8f, 0.8f}, ke[3] = {0, 0, 0}, ks[3] = {0, 0, 0};
    bool has_ks = false;
    float ni = 1.5f, d = 1.0f;
    int illum = 2;
};

Material to_material(const MtlEntry &m)
{
    if (m.illum == 3 || m.illum == 8) {
        const float *a = m.has_ks ? m.ks : m.kd;
        return Material({a[0], a[1], a[2], 0.f}, {m.ke[0], m.ke[1], m.ke[2], 0.f}, Material::Type::MIRROR);
    }
    if (m.illum == 4 || m.illum == 6 || m.illum == 7 || m.illum == 9 || m.d < 1.0f)
        return Material({m.kd[0], m.kd[1], m.kd[2], m.ni}, {m.ke[0], m.ke[1], m.ke[2], 0.f}, Material::Type::DIELECTRIC);
    return Material({m.kd[0], m.kd[1], m.kd[2], 0.f}, {m.ke[0], m.ke[1], m.ke[2], 0.f}, Material::Type::LAMBERT);
}

std::string rest_of_line(std::istringstream &ls)
{
    std::string out, tok;
    while (ls >> tok) out += (out.empty() ? "" : " ") + tok;
    return out;
}

void read_mtl(const std::string &path, std::vector<std::pair<std::string, MtlEntry>> &library)
{
    std::ifstream in(path);
    std::string line;
    MtlEntry *cur = nullptr;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string tag;
        if (!(ls >> tag) || tag[0] == '#') continue;
        if (tag == "newmtl") {
            const std::string name = rest_of_line(ls);
            cur = nullptr;
            for (auto &e : library)
                if (e.first == name) cur = &e.second;
            if (!cur) {
                library.emplace_back(name, MtlEntry{});
                cur = &library.back().second;
            } else {
                *cur = MtlEntry{};
            }
        } else if (!cur) {
            continue;
        } else if (tag == "Kd") {
            ls >> cur->kd[0] >> cur->kd[1] >> cur->kd[2];
        } else if (tag == "Ke") {
            ls >> cur->ke[0] >> cur->ke[1] >> cur->ke[2];
        } else if (tag == "Ks") {
            ls >> cur->ks[0] >> cur->ks[1] >> cur->ks[2];
            cur->has_ks = true;
        } else if (tag == "Ni") {
            ls >> cur->ni;
        } else if (tag == "illum") {
            float v = 2;
            ls >> v;
            cur->illum = static_cast<int>(v);
        } else if (tag == "d") {
            ls >> cur->d;
        } else if (tag == "Tr") {
            float tr = 0;
            ls >> tr;
            cur->d = 1.0f - tr;
        }
    }
}

}  // namespace

long load_scene(RVPT &rvpt, const std::string &path, std::string *error)
{
    std::ifstream in(path);
    if (!in) {
        if (error) *error = "cannot open " + path;
        return -1;
    }
    const size_t slash = path.find_last_of('/');
    const std::string base = slash == std::string::npos ? std::string() : path.substr(0, slash + 1);
    std::vector<std::pair<std::string, MtlEntry>> library;
    std::vector<std::string> used;  // material names in order of first use
    const int first_id = static_cast<int>(rvpt.materials().size());
    auto material_id = [&](const std::string &name) {
        const MtlEntry *entry = nullptr;
        for (const auto &e : library)
            if (e.first == name) entry = &e.second;
        const std::string key = entry ? name : std::string("default");
        for (size_t i = 0; i < used.size(); ++i)
            if (used[i] == key) return first_id + static_cast<int>(i);
        used.push_back(key);
        rvpt.add_material(entry ? to_material(*entry) : Material({1, 1, 1, 0}, {0, 0, 0, 0}, Material::Type::LAMBERT));
        return first_id + static_cast<int>(used.size()) - 1;
    };
    std::vector<vec3> verts;
    std::string current;
    bool have_current = false;
    long added = 0;
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string tag;
        if (!(ls >> tag)) continue;
        if (tag == "v") {
            vec3 v;
            ls >> v.x >> v.y >> v.z;
            verts.push_back(v);
        } else if (tag == "mtllib") {
            std::string lib;
            while (ls >> lib) read_mtl(base + lib, library);
        } else if (tag == "usemtl") {
            current = rest_of_line(ls);
            have_current = true;
        } else if (tag == "f") {
            std::vector<long> idx;
            std::string tok;
            while (ls >> tok) {
                const long i = std::strtol(tok.c_str(), nullptr, 10);
                idx.push_back(i > 0 ? i - 1 : static_cast<long>(verts.size()) + i);
            }
            const int mid = material_id(have_current ? current : std::string("default"));
            for (size_t k = 1; k + 1 < idx.size(); ++k) {
                const long a = idx[0], b = idx[k], c = idx[k + 1], n = static_cast<long>(verts.size());
                if (a < 0 || b < 0 || c < 0 || a >= n || b >= n || c >= n) {
                    if (error) *error = "face index out of range in " + path;
                    return -1;
                }
                rvpt.add_triangle(Triangle(verts[a], verts[b], verts[c], mid));
                ++added;
            }
        }
    }
    return added;
}

void add_default_materials(RVPT &rvpt)
{
    rvpt.add_material(Material({1, 1, 1, 0}, {0.1f, 0.4f, 0.6f, 0}, Material::Type::LAMBERT));
    rvpt.add_material(Material({1, 1, 1, 0}, {0, 0, 0, 0}, Material::Type::LAMBERT));
}

std::vector<uint32_t> launch_sizes(uint32_t frames, uint32_t batch, uint32_t in_flight)
{
    std::vector<uint32_t> sizes;
    if (frames == 0) return sizes;
    batch = std::max(1u, batch);
    (void)in_flight;  // (measured: launches in flight do not overlap for the kernels that matter; fewer, larger launches win)
    const uint32_t n = (frames + batch - 1) / batch;
    for (uint32_t i = 0; i < n; ++i) sizes.push_back(frames / n + (i < frames % n ? 1u : 0u));
    return sizes;
}

}  // namespace rvpt

// rvpt_render — headless counterpart of the reference's main() (src/rvpt/main.cpp:88-159): builds the demo scene
// (load_model + two materials, main.cpp:102-107), then runs update()/draw() for a number of frames instead of the
// window's event loop, and writes the accumulated frame as a PFM image.
//
//   rvpt_render (--obj model.obj [--material-id 1] | --scene scene.obj) [--width 1024 --height 512] [--spp 1] [--bounces 8] [--frames 16] [--batch 1]
//               [--traversal bvh|bvh_ordered|brute] [--translate x y z] [--rotate x y z] [--fov 90] [--mode 9] [--camera-mode 0]
//               [--out frame.pfm] [--dump-prefix path]   (dump: camera block, sorted triangles, nodes, materials)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "rvpt_host.h"

namespace {

bool write_pfm(const std::string &path, const std::vector<float> &rgba, uint32_t w, uint32_t h)
{
    std::ofstream f(path, std::ios::binary);
    if (!f) return false;
    f << "PF\n" << w << " " << h << "\n-1.0\n";
    for (uint32_t y = h; y-- > 0;)  // PFM stores the bottom row first
        for (uint32_t x = 0; x < w; ++x) f.write(reinterpret_cast<const char *>(&rgba[(static_cast<size_t>(y) * w + x) * 4]), 12);
    return static_cast<bool>(f);
}

template <typename T>
void dump(const std::string &path, const T *data, size_t n)
{
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char *>(data), static_cast<std::streamsize>(n * sizeof(T)));
}

}  // namespace

int main(int argc, char **argv)
{
    // before the first HIP call: the context's seven streams (six for launches in flight + one main) must not share hardware queues; the library
    // itself leaves the environment alone
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    std::string obj, scene_obj, out = "frame.pfm", dump_prefix, traversal = "bvh";
    uint32_t width = 1024, height = 512;  // Window::Settings, main.cpp:95-98
    int spp = 1, bounces = 8, frames = 16, batch = 1, material_id = 1, mode = 9, camera_mode = 0, gpus = 1;
    bool force_collective = false;
    rvpt::vec3 translate{}, rotate{};
    float fov = 90.f;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * { return (i + 1 < argc) ? argv[++i] : ""; };
        if (a == "--obj") obj = next();
        else if (a == "--scene") scene_obj = next();
        else if (a == "--out") out = next();
        else if (a == "--dump-prefix") dump_prefix = next();
        else if (a == "--traversal") traversal = next();
        else if (a == "--width") width = static_cast<uint32_t>(std::atoi(next()));
        else if (a == "--height") height = static_cast<uint32_t>(std::atoi(next()));
        else if (a == "--spp") spp = std::atoi(next());
        else if (a == "--bounces") bounces = std::atoi(next());
        else if (a == "--frames") frames = std::atoi(next());
        else if (a == "--batch") batch = std::max(1, std::min(64, std::atoi(next())));
        else if (a == "--material-id") material_id = std::atoi(next());
        else if (a == "--gpus") gpus = std::max(1, std::atoi(next()));  // the image is tile-partitioned over devices 0..gpus-1
        else if (a == "--force-collective") force_collective = true;      // one GPU, but through the RCCL gather all the same
        else if (a == "--mode") mode = std::atoi(next());
        else if (a == "--camera-mode") camera_mode = std::atoi(next());
        else if (a == "--fov") fov = static_cast<float>(std::atof(next()));
        else if (a == "--translate") { translate.x = static_cast<float>(std::atof(next())); translate.y = static_cast<float>(std::atof(next())); translate.z = static_cast<float>(std::atof(next())); }
        else if (a == "--rotate") { rotate.x = static_cast<float>(std::atof(next())); rotate.y = static_cast<float>(std::atof(next())); rotate.z = static_cast<float>(std::atof(next())); }
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (obj.empty() && scene_obj.empty()) { std::fprintf(stderr, "usage: rvpt_render (--obj model.obj | --scene scene.obj) [options]\n"); return 2; }
    if (gpus > 1) {  // --gpus N means N devices: never a silent run on fewer (a figure labelled "gpus": N must be N GPUs' work)
        int visible = 0;
        if (rvpt_hip_device_count(&visible) != RVPT_HIP_OK) { std::fprintf(stderr, "no HIP device: %s\n", rvpt_hip_last_error(nullptr)); return 1; }
        if (gpus > visible) { std::fprintf(stderr, "%d GPUs requested, %d visible\n", gpus, visible); return 1; }
    }

    // One RVPT per GPU: rank i owns the 16x16 tiles t with t % gpus == i and renders them with no exchange; one RCCL group
    // over the contexts (rvpt_hip_comm_init_all) gathers the per-tile radiance when the frame is read.  gpus == 1 is the
    // reference's single-device shape.
    std::vector<std::unique_ptr<rvpt::RVPT>> ranks;
    std::string err;
    long n = 0;
    for (int g = 0; g < gpus; ++g) {
        rvpt::RVPT::Options opt;
        opt.device = g;
        opt.tile_rank = static_cast<uint32_t>(g);
        opt.tile_world = static_cast<uint32_t>(gpus);
        opt.bvh_traversal = traversal != "brute";
        opt.ordered_children = traversal == "bvh_ordered";
        ranks.emplace_back(new rvpt::RVPT(width, height, opt));
        rvpt::RVPT &rvpt = *ranks.back();
        if (!scene_obj.empty()) {  // OBJ + MTL scene description: materials come from the file
            n = rvpt::load_scene(rvpt, scene_obj, &err);
        } else {
            n = rvpt::load_model(rvpt, obj, material_id, &err);  // main.cpp:102
            if (n >= 0) rvpt::add_default_materials(rvpt);       // main.cpp:105-107
        }
        if (n < 0) { std::fprintf(stderr, "[ERROR: MODEL-LOADING] %s\n", err.c_str()); return 1; }
        rvpt.render_settings.aa = spp;
        rvpt.render_settings.max_bounces = bounces;
        rvpt.render_settings.top_left_render_mode = rvpt.render_settings.top_right_render_mode = mode;
        rvpt.render_settings.bottom_left_render_mode = rvpt.render_settings.bottom_right_render_mode = mode;
        rvpt.scene_camera.translation = translate;
        rvpt.scene_camera.rotation = rotate;
        rvpt.scene_camera.set_fov(fov);
        rvpt.scene_camera.set_camera_mode(camera_mode);
        if (!rvpt.initialize()) { std::fprintf(stderr, "failed to initialize RVPT on device %d: %s\n", g, rvpt.last_error().c_str()); return 1; }  // main.cpp:109-114
    }
    rvpt::RVPT &rvpt = *ranks[0];
    if (gpus > 1 || force_collective) {
        std::vector<rvpt_hip_ctx *> ctxs;
        for (auto &r : ranks) ctxs.push_back(r->context());
        if (rvpt_hip_comm_init_all(ctxs.data(), gpus) != RVPT_HIP_OK) {
            std::fprintf(stderr, "RCCL group: %s\n", rvpt_hip_last_error(ctxs[0]));
            return 1;
        }
    }

    const auto t0 = std::chrono::steady_clock::now();
    // the body of main.cpp:139-155 without window / ImGui.  The camera stands still: accumulation frames may go out in batches
    // (--batch: as few launches as it allows, of near-equal size, rvpt::launch_sizes); --batch 1 is update()/draw() per frame
    for (const uint32_t nb : rvpt::launch_sizes(static_cast<uint32_t>(frames), static_cast<uint32_t>(batch))) {
        for (auto &r : ranks) {
            if (!r->update()) return 1;
            if (nb > 1)
                r->draw_frames(nb);
            else
                r->draw();
        }
    }
    for (auto &r : ranks) r->wait();
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const std::vector<float> img = rvpt.read_frame();  // with a group: the gather of every rank's tiles to rank 0
    if (img.empty() || !write_pfm(out, img, width, height)) { std::fprintf(stderr, "could not write %s: %s\n", out.c_str(), rvpt.last_error().c_str()); return 1; }
    if (!dump_prefix.empty()) {
        const rvpt_camera_data cam = rvpt.scene_camera.get_data();
        dump(dump_prefix + ".camera.f32", reinterpret_cast<const float *>(&cam), 20);
        dump(dump_prefix + ".triangles.f32", reinterpret_cast<const float *>(rvpt.sorted_triangles().data()), rvpt.sorted_triangles().size() * 16);
        dump(dump_prefix + ".nodes.bin", rvpt.bvh_nodes().data(), rvpt.bvh_nodes().size());
        dump(dump_prefix + ".materials.f32", reinterpret_cast<const float *>(rvpt.materials().data()), rvpt.materials().size() * 12);
    }
    std::printf("{\"triangles\": %ld, \"frames\": %d, \"last_frame\": %u, \"seconds\": %.6f, \"Msamples_per_s\": %.1f, \"gpus\": %d, \"collective\": %s, \"out\": \"%s\"}\n", n, frames,
                rvpt.render_settings.current_frame, secs, static_cast<double>(width) * height * spp * frames / secs / 1e6, gpus,
                (gpus > 1 || force_collective) ? "true" : "false", out.c_str());
    for (auto &r : ranks) r->shutdown();
    return 0;
}

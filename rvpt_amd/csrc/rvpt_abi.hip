// rvpt_abi.hip — implementation of the C ABI declared in include/rvpt_hip.h.
//
// Host-side counterpart of the reference seam inside `class RVPT` (src/rvpt/rvpt.cpp): resource
// creation (:639-866), per-frame upload (:96-126), dispatch (:1005-1039, :352-354) and the fence
// (:115-116) become HIP runtime calls on one stream of one device.  No exception leaves this file.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rvpt_hip.h"
#include "../../include/rvpt_hip_lab.h"
#include "rvpt_kernels.h"
#include "bvh_wide.h"
#include "rvpt_packets.h"
#include "rvpt_math.h"
#include "rvpt_rect.h"
#include "rvpt_vis.h"

#ifndef RVPT_HIP_LAB
#define RVPT_HIP_LAB 0  // 1: the laboratory build (librvpt_hip_debug.so): + the selftests and host-side forms of include/rvpt_hip_lab.h, the opt-in walks, the tuning knobs
#endif
#ifndef RV_REPORT_STACK_OVERFLOW
#define RV_REPORT_STACK_OVERFLOW 0  // (rvpt_device.h) 1: the BVH kernels report a push past the stack the host sized
#endif

// The handful of RCCL (NCCL API) types the gather needs, declared here so that the library BUILDS without the RCCL headers — a
// single-GPU host needs neither header nor library; librccl.so is dlopen()ed when a communicator is first asked for.  Values as in
// rccl.h (NCCL 2.x ABI: ncclFloat32 = 7, ncclSum = 0, a 128-byte unique id).
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;  // anything else is an error, described by ncclGetErrorString
typedef enum { ncclFloat = 7 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
}

static_assert(rv::kWideChildren == rv::kWideFormChildren && rv::kWideEmpty == rv::kWideFormEmpty, "bvh_wide.h and rvpt_kernels.h describe the same node");
static_assert(sizeof(rvpt_triangle) == 64, "Triangle layout (structs.glsl:1-7)");
static_assert(sizeof(rvpt_bvh_node) == 32, "BvhNode layout (structs.glsl:9-14)");
static_assert(sizeof(rvpt_material) == 48, "Material layout (structs.glsl:22-33)");
static_assert(sizeof(rvpt_render_settings) == 40, "RenderSettings std140 block (compute_pass.comp:28-40)");
static_assert(sizeof(rvpt_camera_data) == 80, "Camera block (compute_pass.comp:44-49)");
static_assert(offsetof(rvpt_triangle, mat_id) == 48 && offsetof(rvpt_bvh_node, bounds) == 8 &&
                  offsetof(rvpt_material, data) == 32 && offsetof(rvpt_render_settings, split_ratio) == 32 &&
                  offsetof(rvpt_camera_data, params) == 64,
              "member offsets");

struct rvpt_hip_ctx {
    int device = 0;
    uint32_t width = 0, height = 0, tiles_x = 0, tiles_y = 0;
    uint32_t tile_rank = 0, tile_world = 1, flags = 0;
    uint32_t n_local_tiles = 0, n_work = 0;
    int num_cus = 0;
    hipStream_t stream = nullptr;            // uploads, blend, read-back — in order
    // frames in flight: frame kernels rotate over `n_slots` streams / sample buffers / counter sets
    // (the reference keeps MAX_FRAMES_IN_FLIGHT = 2 per-frame resource sets, rvpt.h:25)
    static constexpr int kMaxSlots = 8;
    int n_slots = 6;                 // allocated; how many of them a launch rotates over is chosen per launch (slots_for)
    bool slots_fixed = false;        // RVPT_HIP_FRAMES_IN_FLIGHT given: always all of them
    int next_slot = 0;
    bool slot_used[kMaxSlots] = {};  // a launch has gone out on it (its blend_done event means something)
    int last_slots = 0;              // what the last launch rotated over (rvpt_hip_get_launch_info)
    hipStream_t trace_stream[kMaxSlots] = {};
    // Sample buffers: TWO per stream slot, used alternately (round 6).  A launch writes the per-pixel sample means of its frames into one and the blend on the main
    // stream consumes them; with one buffer per slot the slot's NEXT launch had to wait for that blend — trace -> event -> blend -> event -> trace, two cross-stream
    // hand-offs of ~17-20 us each on a 40-us frame (rocprofv3 kernel trace of one frame per launch: profiles/r06_launch_shapes.txt).  With two, the next launch on a
    // stream follows its predecessor in stream order and only waits for the blend of the launch BEFORE that, long finished.
    static constexpr int kBufsPerSlot = 2;
    hipEvent_t trace_done[kMaxSlots] = {}, blend_done[kMaxSlots * kBufsPerSlot] = {};
    rv::SampleRGB *d_samples[kMaxSlots * kBufsPerSlot] = {};  // per-launch sample means awaiting the blend (samples_cap frames each); buffer = slot * 2 + parity
    uint32_t samples_cap[kMaxSlots * kBufsPerSlot] = {};
    bool buf_used[kMaxSlots * kBufsPerSlot] = {};  // a launch has gone out on it (its blend_done event means something)
    int slot_parity[kMaxSlots] = {};
    size_t slot_quads = 0;
    bool overlap = true;
    uint64_t seq = 0;

    float4 *d_tris = nullptr, *d_prep = nullptr, *d_mats = nullptr, *d_nodes = nullptr;
    uint32_t *d_mat_index = nullptr;
    float4 *d_unit_n = nullptr;  // (normalize(n), 0) per triangle (prepare_triangles)
    size_t cap_unit_n = 0;
    size_t n_tris = 0, n_mats = 0, n_nodes = 0;
    uint32_t bvh_height = 0;  // nodes on the longest root-to-leaf path
    // 4-wide form of the same tree (build_wide_nodes; rvpt_bvh4.hip): 128-byte nodes of up to four children, 0 when the tree has no wide form
    float4 *d_wide = nullptr;
    size_t n_wide = 0, cap_wide = 0;
    uint32_t wide_stack_levels = 0;  // most slots a depth-first walk of the wide tree can hold at once
    // the 64-byte quantised form of the wide nodes + the exact leaf boxes by first triangle (trace_bvh4q; build_quant_nodes, build_leaf_boxes): trees every
    // inner node of which contains its children
    float4 *d_wideq = nullptr, *d_leaf_box = nullptr;
    size_t cap_wideq = 0, cap_leaf_box = 0;
    bool has_quant = false;
    float slab_extent = 0.0f;
    int bvh_quant = 0;               // RVPT_HIP_BVH_QUANT=1: the 64-byte quantised nodes (bit-exact, measured SLOWER: profiles/EXPERIMENTS.md 5.16)
    float4 *d_wide8 = nullptr;       // the 8-wide form (rvpt_bvh8.hip; RVPT_HIP_BVH_WIDE8=1): 256-byte nodes
    size_t n_wide8 = 0, cap_wide8 = 0;
    uint32_t wide8_stack_levels = 0;
    int bvh_wide8 = 0;
    int bvh_wide = 1;                // policy: BVH contexts, lean configuration, reference order, HBM-resident scene: the wide kernel (RVPT_HIP_BVH_WIDE=0 / RVPT_HIP_BVH_PER_LANE: binary)
    uint32_t bvh_head_shift = 0;  // see FrameParams::head_shift
    size_t cap_tris = 0, cap_prep = 0, cap_mat_index = 0, cap_mats = 0, cap_nodes = 0;  // allocated elements
    bool have_scene = false;

    rvpt_render_settings settings{};
    rvpt_camera_data camera{};
    bool have_frame = false;

    float4 *d_accum = nullptr;
    void *d_rowmajor = nullptr;  // width*height*16 B staging for read / write_accum
    unsigned long long *d_counter = nullptr, *d_stats = nullptr;
    // multi-GPU gather of per-tile radiance: RCCL communicator of the tile_world ranks, rank == tile_rank
    ncclComm_t comm = nullptr;
    hipStream_t comm_stream = nullptr;        // collectives (+ rank 0's un-tiling) run here, never on `stream`: a collective that hangs — a peer that never
                                              // arrives, an abort that fails — leaves rendering, read-back and wait untouched
    bool comm_stream_lost = false;            // a collective timed out on comm_stream: it is abandoned (never waited for, never destroyed)
    float *d_barrier = nullptr;               // one float: the payload of rvpt_hip_comm_barrier's all-reduce
    std::vector<rvpt_hip_ctx *> local_group;  // single-process form (comm_init_all): every rank's context, index == rank
    uint32_t *d_stack_overflow[kMaxSlots] = {};  // HBM-resident BVH kernel: stack levels beyond the LDS ones, per launch in flight
    size_t stack_overflow_cap[kMaxSlots] = {};   // in words
    int brute_packets_policy = 1;             // LDS-resident brute force, lean configuration: the packet kernel (default; RVPT_HIP_BRUTE_MIXED_PACKETS or
                                              // RVPT_HIP_BRUTE_PACKETS=0 select round 2's trace_brute_resident)
    // screen rectangles of the triangles for the packet kernel's camera rounds (rvpt_rect.h), one buffer per launch slot: rewritten (camera_rects, on the
    // slot's own stream, in front of the frame kernel) only when the camera, the image size or the scene differ from what the slot's buffer was made for
    int debug_checks = 0;                     // RVPT_HIP_DEBUG=1: rvpt_hip_wait reads the kernels' error words (traversal-stack overflow) and fails loudly
    int force_stack_levels = 0;               // RVPT_HIP_BVH_FORCE_STACK_LEVELS (tests of the above): lie to the kernels about the stack the tree needs
    int packets_cull = 1;                     // RVPT_HIP_PACKETS_CULL=0: no rectangles (A/B)
    // the bounce cull's table (bounce_visibility, once per upload of a scene the packet kernel can hold): 2 n rows of ceil(n / 32) words
    int packets_bounce_cull = 1;              // RVPT_HIP_PACKETS_BOUNCE_CULL=0: off (A/B)
    uint32_t *d_vis = nullptr;
    size_t vis_cap = 0;                       // in words
    float4 *d_leaf_boxes = nullptr;           // the leaf boxes of the bounce rounds (rvpt_vis.h), made with the table; two float4 per kLeafTris triangles
    size_t leaf_boxes_cap = 0;                // in float4
    int packets_box_cull = 1;                 // RVPT_HIP_PACKETS_BOX_CULL=0: off (A/B)
    int packets_lean_instance = 1;            // (laboratory build: RVPT_HIP_PACKETS_LEAN_INSTANCE=0 keeps the general instances — A/B)
    int packets_interleave = 1;               // RVPT_HIP_PACKETS_INTERLEAVE=g: launches of fewer than four frames deal groups of g (1, 2, 4, 8) blocks from all over the frame; 0 = tile-linear order (A/B)
    int packets_interleave_all = 0;           // ... =-g: launches of any size (measured slower for the batched ones: profiles/r06_interleave.txt)
    uint32_t vis_words = 0;                   // 0: no table for this scene
    double scene_scale = 0.0;                 // largest |coordinate| + largest extent of the uploaded triangles: what float errors of positions scale with
    uint2 *d_rects[kMaxSlots] = {};
    float4 *d_cam_records[kMaxSlots] = {};   // the camera records of the packet kernel (rvpt_early_out.h), made with the slot's rectangles: 16 B per triangle
    size_t rects_cap[kMaxSlots] = {};         // in triangles
    struct RectKey {
        rvpt_camera_data camera;
        uint64_t scene_gen;
    } rects_key[kMaxSlots] = {};
    bool rects_valid[kMaxSlots] = {};
    uint64_t scene_gen = 0;                   // bumped by every upload_scene
    float4 *d_gather = nullptr;               // rank 0: tile_world slots of slot_quads
    void *d_quant = nullptr;                  // rank 0: width*height*4 B, rgba8 of a gathered frame
    unsigned long long *d_timeline = nullptr;  // RVPT_HIP_TIMELINE=<file>: per-wave timestamps of the last frame
    size_t timeline_words = 0;
    std::string timeline_path;

    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending, spare;
    float last_ms = 0.f;
    double sum_ms = 0.0;
    uint64_t n_timed = 0;
    uint32_t last_grid = 0, last_lds = 0, last_variant = 0;
    uint32_t last_cull = 0;  // rvpt_hip_get_cull_info: what the last launch rode with
    // tuning knobs, read from the environment once at create (0 = use the built-in policy)
    struct {
        int blocks_per_cu = 0, first_units = 0, claim_units = 0, bvh_refill = 0, bvh_leaf_batch = 0;
        int bvh_top_nodes = -1;  // -1 = the built-in 256
        int bvh_stack_lds = 0;   // stack levels kept in LDS by the HBM-resident BVH kernel (0 = built-in 8)
        int bvh_wide_resident = 1;  // RVPT_HIP_BVH_WIDE_RESIDENT=0: LDS-resident scenes on the binary camera-packet kernel instead of the wide tree (trace_bvh4_resident)
        int bvh_no_resident = 0; // RVPT_HIP_BVH_NO_RESIDENT: never the LDS-resident BVH instances
        int bvh_cam_min = 0;     // camera packets: lanes that must start a camera ray together (0 = built-in)
        int bvh_detach = -1;     // camera packets: the lanes of a node leave the packet at this many or fewer (-1 = built-in)
    } tune;
    const void *occ_kernel = nullptr;  // cached occupancy query (kernel, lds) -> work-groups per CU
    size_t occ_lds = 0;
    int occ_per_cu = 0;

    std::string err;
};

namespace {

// NOTE: the library does not touch the process environment.  Frames in flight want GPU_MAX_HW_QUEUES >= 8 set by the
// HOST before HIP initialises (include/rvpt_hip.h); rvpt_render, the Python package and bench.py do so.

thread_local std::string g_err;  // for calls that fail before a context exists

// Knobs of the laboratory build (librvpt_hip_debug.so, -DRVPT_HIP_LAB=1: rvpt_amd/build.py): the sweeps of rounds 1-5 found each of them flat around its
// default (profiles/r05_bvh_thresholds.txt, EXPERIMENTS.md), so the release library reads none of them — there they are the constants of the built-in policy.
inline const char *lab_env(const char *name)
{
#if RVPT_HIP_LAB
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}
void drop_comm(rvpt_hip_ctx *ctx);  // defined with the collective, used by destroy

// RCCL entry points, resolved on first use: a single-GPU host needs no RCCL at all, and inside a torch process the
// already-loaded librccl (same SONAME) is the one that answers
struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;     // optional (rvpt_hip_comm_info)
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;  // optional
    ncclResult_t (*GetVersion)(int *) = nullptr;                      // optional
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::string why;
};
const Rccl &rccl()
{
    static const Rccl api = [] {
        Rccl r;
        void *h = nullptr;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) {
            r.why = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : "");
            return r;
        }
        auto sym = [&](const char *n) { return dlsym(h, n); };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
        r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
        r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
        r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(sym("ncclCommAbort"));
        r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
        r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
        r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(sym("ncclGetVersion"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
        r.ok = r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.Send && r.Recv && r.AllReduce && r.GetErrorString;
        if (!r.ok) r.why = "librccl.so lacks an expected entry point";
        return r;
    }();
    return api;
}

int fail(rvpt_hip_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    (ctx ? ctx->err : g_err) = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                         \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(ctx, RVPT_HIP_ERR_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define RCCL_TRY(ctx, expr)                                                                                       \
    do {                                                                                                          \
        ncclResult_t r_ = (expr);                                                                                 \
        if (r_ != ncclSuccess) return fail(ctx, RVPT_HIP_ERR_COMM, "%s -> %s", #expr, rccl().GetErrorString(r_)); \
    } while (0)

template <typename T>
int grow(rvpt_hip_ctx *ctx, T *&ptr, size_t &cap, size_t need, size_t elem_bytes)
{
    if (need <= cap && ptr) return 0;
    if (ptr) HIP_TRY(ctx, hipFree(ptr));
    ptr = nullptr;
    cap = 0;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ptr), std::max<size_t>(need, 1) * elem_bytes));
    cap = std::max<size_t>(need, 1);
    return 0;
}

int ensure_rowmajor(rvpt_hip_ctx *ctx)
{
    if (!ctx->d_rowmajor) HIP_TRY(ctx, hipMalloc(&ctx->d_rowmajor, static_cast<size_t>(ctx->width) * ctx->height * 16));
    return 0;
}

// fold finished event pairs into the running totals (blocks until they are complete)
int drain_timing(rvpt_hip_ctx *ctx)
{
    for (auto &pr : ctx->pending) {
        HIP_TRY(ctx, hipEventSynchronize(pr.second));
        float ms = 0.f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, pr.first, pr.second));
        ctx->last_ms = ms;
        ctx->sum_ms += ms;
        ctx->n_timed += 1;
        ctx->spare.push_back(pr);
    }
    ctx->pending.clear();
    return 0;
}

// every frame kernel in flight, then everything queued behind them on the main stream
int sync_all(rvpt_hip_ctx *ctx)
{
    for (int i = 0; i < ctx->n_slots; ++i) HIP_TRY(ctx, hipStreamSynchronize(ctx->trace_stream[i]));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

uint32_t owned_tiles(uint32_t n_tiles, uint32_t rank, uint32_t world) { return (n_tiles > rank) ? (n_tiles - rank + world - 1) / world : 0; }

using Kernel = void (*)(const rv::FrameParams);
struct Launch {
    Kernel kernel;
    size_t lds;        // dynamic LDS bytes per work-group
    uint32_t grid;     // work-groups
    uint32_t variant;  // 0 brute/LDS-resident (mixed packets), 1 brute/LDS-streamed, 2 bvh, 3 bvh/LDS-resident, 6 brute/LDS-resident packet kernel,
                       // 10 bvh over the 4-wide tree, 11 the same with the scene in LDS and camera packets, 12 the 8-wide walk (RVPT_HIP_BVH_WIDE8=1),
                       // 13 the 4-wide walk over 64-byte quantised nodes (RVPT_HIP_BVH_QUANT=1) (9: the streamed packet kernel of round 4,
                       // measured no faster and retired: profiles/r04_exp_stream_packets.patch)
                       // (4, 5: the wavefront pipelines of round 3, retired in ABI 5 — profiles/r04_exp_wavefront_pipelines.patch)
    bool regen;
    int slots = 3;     // launches in flight this launch rotates over (slots_for)
    bool lone = false; // no launch of this context is in flight when this one goes out
    bool cull = false; // packet kernel: the screen rectangles of the triangles ride along (FrameParams::rects)
};

// scene pointers, image geometry, the settings/camera blocks of this frame (compute_pass.comp:28-54)
void fill_frame_params(const rvpt_hip_ctx *ctx, int slot, rv::FrameParams &p, int buf = -1)
{
    const rvpt_render_settings &s = ctx->settings;
    p.tris = ctx->d_tris;
    p.prep = ctx->d_prep;
    p.mat_index = ctx->d_mat_index;
    p.unit_n = ctx->d_unit_n;
    p.mats = ctx->d_mats;
    p.nodes = ctx->d_nodes;
    p.accum = ctx->d_accum;
    p.counter = ctx->d_counter + (ctx->overlap ? slot * rv::kCounterWords : 0);
    p.sample_out = ctx->overlap ? ctx->d_samples[buf >= 0 ? buf : slot * rvpt_hip_ctx::kBufsPerSlot] : nullptr;
    p.stats = (ctx->flags & RVPT_HIP_COUNT_SEGMENTS) ? ctx->d_stats : nullptr;
    p.timeline = nullptr;
    p.n_tris = static_cast<uint32_t>(ctx->n_tris);
    p.n_mats = static_cast<uint32_t>(ctx->n_mats);
    p.n_nodes = static_cast<uint32_t>(ctx->n_nodes);
    p.n_work = ctx->n_work;
    p.n_work_frame = ctx->n_work;
    p.div_work_frame = rv::fast_div_make(ctx->n_work);
    p.div_tiles_x = rv::fast_div_make(ctx->tiles_x);
    p.width = ctx->width;
    p.height = ctx->height;
    p.tiles_x = ctx->tiles_x;
    p.tile_rank = ctx->tile_rank;
    p.tile_world = ctx->tile_world;
    p.frame = s.current_frame;
    p.quantize = (ctx->flags & RVPT_HIP_ACCUM_UNORM8) ? 1u : 0u;
    p.camera_mode = s.camera_mode;
    p.modes[0] = s.top_left_render_mode;
    p.modes[1] = s.top_right_render_mode;
    p.modes[2] = s.bottom_left_render_mode;
    p.modes[3] = s.bottom_right_render_mode;
    p.split_x = s.split_ratio[0];
    p.split_y = s.split_ratio[1];
    p.max_bounces = s.max_bounces;
    p.aa = s.aa;
    p.inv_w = 1.0f / static_cast<float>(ctx->width);  // compute_pass.comp:51
    p.inv_h = 1.0f / static_cast<float>(ctx->height);
    p.cf = static_cast<float>(s.current_frame);                  // :53
    p.inv_cf = 1.0f / static_cast<float>(s.current_frame + 1u);  // :54
    p.aspect = ctx->camera.params[0];
    p.cam_w = 1.0f / rv::tan_det(0.5f * ctx->camera.params[1]);  // camera.glsl:42
    p.ortho_scale = ctx->camera.params[2];
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) p.cam[3 * c + r] = ctx->camera.matrix[4 * c + r];
    for (int r = 0; r < 3; ++r) p.cam[9 + r] = ctx->camera.matrix[12 + r];
}

// LDS-resident BVH scenes (nodes + triangles + materials + the whole stack within 64 KiB) run their own kernel instance
bool bvh_scene_fits_lds(const rvpt_hip_ctx *ctx, uint32_t stack_levels)
{
    if (ctx->tune.bvh_no_resident) return false;  // experiments (RVPT_HIP_BVH_NO_RESIDENT): small scenes through the HBM-resident kernels
    const size_t index_bytes = ((ctx->n_tris + 3) & ~size_t(3)) * 4;
    const size_t bvh_scene_bytes = ctx->n_nodes * 32 + ctx->n_tris * 64 + index_bytes + ctx->n_mats * 48;
    const size_t full_stack_bytes = static_cast<size_t>(stack_levels) * rv::kBlock * 2 * sizeof(uint32_t);  // two words per slot
    return bvh_scene_bytes <= rv::kBvhResidentBytes && bvh_scene_bytes + full_stack_bytes <= 64 * 1024;
}

// How many launches in flight.  Three, except for SHORT launches of the HBM-resident BVH kernel (one frame of a large scene,
// the interactive case: a moving camera leaves nothing to batch): such a launch is mostly ramp-up and tail, and six of them at two
// work-groups per CU overlap those better (tools/archive/sweep_bvh_b1.sh: 1 M-triangle terrain 1080p x 1 spp 4 400 -> 6 180 Msamples/s,
// Cornell 1080p x 4 spp 2 470 -> 2 600; long launches and the LDS-resident kernels: three is as good or better).
int slots_for(const rvpt_hip_ctx *ctx, uint32_t n_frames)
{
    if (!ctx->overlap) return 1;
    if (ctx->slots_fixed) return ctx->n_slots;
    const bool bvh = (ctx->flags & RVPT_HIP_TRAVERSAL_MASK) != RVPT_HIP_TRAVERSAL_BRUTE && ctx->n_nodes > 0;
    const uint32_t levels = std::max<uint32_t>(1, std::min<uint32_t>(rv::kBvhStackDepth, ctx->bvh_height));
    const bool hbm_bvh = bvh && !bvh_scene_fits_lds(ctx, levels);
    const uint64_t samples = static_cast<uint64_t>(ctx->n_work) * n_frames * static_cast<uint64_t>(std::max(1, ctx->settings.aa));
    const bool short_launch = samples <= 10ull * 1000 * 1000;
    return std::min(ctx->n_slots, (hbm_bvh && short_launch) ? 6 : 3);
}

// which kernel instance, how much LDS, how many work-groups
int choose_launch(rvpt_hip_ctx *ctx, rv::FrameParams &p, Launch &l)
{
    const bool bvh = (ctx->flags & RVPT_HIP_TRAVERSAL_MASK) != RVPT_HIP_TRAVERSAL_BRUTE && ctx->n_nodes > 0;  // no tree = empty scene
    const bool ordered = bvh && (ctx->flags & RVPT_HIP_TRAVERSAL_MASK) == RVPT_HIP_TRAVERSAL_BVH_ORDERED;
    l.regen = (ctx->flags & RVPT_HIP_KERNEL_SIMPLE) == 0;
    // the lean kernels cover the default configuration (Kajiya everywhere, pinhole); anything else runs the GENERIC ones
    const bool generic = p.camera_mode != 0 || p.modes[0] != 9 || p.modes[1] != 9 || p.modes[2] != 9 || p.modes[3] != 9;
    const size_t index_bytes = ((ctx->n_tris + 3) & ~size_t(3)) * 4;
    // brute force keeps the whole scene in LDS when it fits the 64 KiB a work-group gets without opting in to more
    const size_t resident_bytes = ctx->n_tris * 64 + index_bytes + (ctx->n_mats <= rv::kResidentMaxMats ? ctx->n_mats * 48 : 0) +
                                  (rv::kBlock / 64) * 64 * sizeof(uint32_t);
    const bool resident = !bvh && ctx->n_tris <= rv::kResidentMaxTris && resident_bytes <= 64 * 1024;
    // BVH: traversal stack sized from the tree; nodes + triangles + materials in LDS too when everything fits 64 KiB
    // at most one push per inner level of the path from the root
    p.stack_levels = std::max<uint32_t>(1, std::min<uint32_t>(rv::kBvhStackDepth, ctx->bvh_height));
    p.head_shift = lab_env("RVPT_HIP_BVH_NO_PACKED_HEADS") ? 0u : ctx->bvh_head_shift;
    const size_t bvh_scene_bytes = ctx->n_nodes * 32 + ctx->n_tris * 64 + index_bytes + ctx->n_mats * 48;
    const bool bvh_resident = bvh && bvh_scene_fits_lds(ctx, p.stack_levels);
    // HBM-resident scenes keep only the first stack levels in LDS (the rest overflows to global memory, rarely touched) so
    // that the top of the tree fits beside them at full occupancy; LDS-resident scenes keep the whole stack
    const uint32_t lds_levels_want = ctx->tune.bvh_stack_lds > 0 ? static_cast<uint32_t>(ctx->tune.bvh_stack_lds) : 8u;
    p.stack_lds_levels = bvh_resident ? p.stack_levels : std::min(p.stack_levels, lds_levels_want);
    const size_t stack_bytes = static_cast<size_t>(p.stack_lds_levels) * rv::kBlock * 2 * sizeof(uint32_t);
    // short LDS-resident traversals: let the whole packet finish before refilling (64); long HBM traversals: refill once
    // half of the packet waits, run the parked leaves in batches of 16 lanes, 3 work-groups per CU (swept on the Cornell and
    // 1M-triangle scenes, both traversal orders, frames dispatched in batches: profiles/r01_bvh_knob_sweeps.txt)
    p.bvh_refill = ctx->tune.bvh_refill ? static_cast<uint32_t>(ctx->tune.bvh_refill) : (bvh_resident ? 64u : 32u);
    p.bvh_leaf_batch = ctx->tune.bvh_leaf_batch ? static_cast<uint32_t>(ctx->tune.bvh_leaf_batch) : 16u;
    // top of the tree in LDS (HBM-resident scenes): 256 nodes = 8 KiB by default (with 8 two-word stack levels in LDS: 24 KiB per
    // work-group, six per CU — what the registers allow anyway; swept: tools/archive/sweep_bvh_top.sh, profiles/r02_sweeps.txt), never more than the tree has (even count: sibling pairs)
    const uint32_t top_want = ctx->tune.bvh_top_nodes >= 0 ? static_cast<uint32_t>(ctx->tune.bvh_top_nodes) : 256u;
    p.bvh_top_nodes = (bvh && !bvh_resident) ? std::max(2u, std::min<uint32_t>(top_want, static_cast<uint32_t>(ctx->n_nodes)) & ~1u) : 0u;  // at least the root's line

    const int bvh_per_cu = ctx->tune.blocks_per_cu ? ctx->tune.blocks_per_cu : 3;
    l.lds = bvh ? stack_bytes + (bvh_resident ? bvh_scene_bytes : static_cast<size_t>(p.bvh_top_nodes) * 32) : (resident ? resident_bytes : static_cast<size_t>(rv::kBlock / 64) * (rv::kStreamDepth * rv::kWaveChunk * 64 + 64 * sizeof(uint32_t)));
    l.variant = bvh ? (bvh_resident ? 3u : 2u) : (resident ? 0u : 1u);
    const int sel = (l.regen ? 0 : 1) | (generic ? 2 : 0);
    static const Kernel table[4][4] = {
        {rv::trace_brute_resident<true, false>, rv::trace_brute_resident<false, false>, rv::trace_brute_resident<true, true>,
         rv::trace_brute_resident<false, true>},
        {rv::trace_brute_stream<true, false>, rv::trace_brute_stream<false, false>, rv::trace_brute_stream<true, true>,
         rv::trace_brute_stream<false, true>},
        {rv::trace_bvh<true, false, false, false>, rv::trace_bvh<false, false, false, false>, rv::trace_bvh<true, false, true, false>,
         rv::trace_bvh<false, false, true, false>},
        {rv::trace_bvh<true, true, false, false>, rv::trace_bvh<false, true, false, false>, rv::trace_bvh<true, true, true, false>,
         rv::trace_bvh<false, true, true, false>},
    };
    static const Kernel ordered_table[2][4] = {
        {rv::trace_bvh<true, false, false, true>, rv::trace_bvh<false, false, false, true>, rv::trace_bvh<true, false, true, true>,
         rv::trace_bvh<false, false, true, true>},
        {rv::trace_bvh<true, true, false, true>, rv::trace_bvh<false, true, false, true>, rv::trace_bvh<true, true, true, true>,
         rv::trace_bvh<false, true, true, true>},
    };
    l.kernel = ordered ? ordered_table[bvh_resident ? 1 : 0][sel] : table[l.variant][sel];
    // camera packets (trace_bvh4_resident: lanes that start camera rays together walk the tree as one wave-uniform packet in the reference's fixed child
    // order): at least bvh_cam_min lanes must start a camera ray at once; the lanes of a node leave the packet when at most bvh_detach of them are in it
    // (LDS-resident scenes: nobody leaves measured best; the binary-tree form of round 4, +5 % resident / +-0 HBM-resident, is profiles/r04_exp_campack_binary.patch)
    p.bvh_cam_min = ctx->tune.bvh_cam_min ? static_cast<uint32_t>(ctx->tune.bvh_cam_min) : 24u;
    p.bvh_detach = ctx->tune.bvh_detach >= 0 ? static_cast<uint32_t>(ctx->tune.bvh_detach) : 0u;
    // the 4-wide form of the tree (rvpt_bvh4.hip): scenes that do not fit LDS, lean configuration, reference order; half the dependent steps per ray
    const bool wide = bvh && !bvh_resident && !ordered && l.regen && ctx->n_wide > 0 && ctx->bvh_wide == 1 && p.head_shift != 0;
    if (wide) {
        l.variant = 10u;
#if RVPT_HIP_LAB
        const bool quant = !generic && ctx->has_quant && ctx->bvh_quant == 1;
        if (quant) l.variant = 13u;
        l.kernel = generic ? rv::trace_bvh4_generic : (quant ? rv::trace_bvh4q : rv::trace_bvh4);
#else
        const bool quant = false;
        l.kernel = generic ? rv::trace_bvh4_generic : rv::trace_bvh4;
#endif
        p.leaf_box = quant ? ctx->d_leaf_box : nullptr;
        p.slab_extent = ctx->slab_extent;
        p.wide = quant ? ctx->d_wideq : ctx->d_wide;
        p.n_wide = static_cast<uint32_t>(ctx->n_wide);
        p.stack_levels = std::max<uint32_t>(1, ctx->wide_stack_levels);
        p.stack_lds_levels = std::min(p.stack_levels, lds_levels_want);
        const uint32_t node_bytes = quant ? 64u : rv::kWideTopQuads * 16u;  // one node in the LDS copy of the tree top
        const uint32_t wide_top_want = ctx->tune.bvh_top_nodes >= 0 ? static_cast<uint32_t>(ctx->tune.bvh_top_nodes) : (quant ? 128u : 64u);  // 8 KiB, as the binary kernel's 256 nodes
        // (the knob counts NODES, and a wide node is four binary ones: whatever it asks for, the stack levels + the root record + the top nodes stay within
        // the 64 KiB a work-group can have — ADVICE r4: RVPT_HIP_BVH_TOP_NODES=2048 used to ask for 256 KiB and fail at launch)
        const size_t wide_fixed = static_cast<size_t>(p.stack_lds_levels) * rv::kBlock * 2 * sizeof(uint32_t) + 2 * sizeof(float4);
        const uint32_t wide_top_fit = static_cast<uint32_t>((64 * 1024 - std::min<size_t>(wide_fixed, 64 * 1024)) / node_bytes);
        p.wide_top_nodes = std::min<uint32_t>({wide_top_want, p.n_wide, wide_top_fit});
        l.lds = static_cast<size_t>(p.stack_lds_levels) * rv::kBlock * 2 * sizeof(uint32_t) + 2 * sizeof(float4) + static_cast<size_t>(p.wide_top_nodes) * node_bytes;
    }
#if RVPT_HIP_LAB
    // the 8-wide form (rvpt_bvh8.hip): half the steps of the 4-wide walk again on scenes whose rays see few boxes per level
    if (wide && !generic && ctx->bvh_wide8 == 1 && ctx->n_wide8 > 0) {
        l.variant = 12u;
        l.kernel = rv::trace_bvh8;
        p.wide = ctx->d_wide8;
        p.n_wide = static_cast<uint32_t>(ctx->n_wide8);
        p.stack_levels = std::max<uint32_t>(1, ctx->wide8_stack_levels);
        p.stack_lds_levels = std::min(p.stack_levels, lds_levels_want);
        const uint32_t top8_want = ctx->tune.bvh_top_nodes >= 0 ? static_cast<uint32_t>(ctx->tune.bvh_top_nodes) : 32u;  // 8 KiB
        const size_t fixed8 = static_cast<size_t>(p.stack_lds_levels + 1u) * rv::kBlock * 2 * sizeof(uint32_t) + 2 * sizeof(float4);
        p.wide_top_nodes = std::min<uint32_t>({top8_want, p.n_wide, static_cast<uint32_t>((64 * 1024 - std::min<size_t>(fixed8, 64 * 1024)) / 256)});
        l.lds = static_cast<size_t>(p.stack_lds_levels + 1u) * rv::kBlock * 2 * sizeof(uint32_t) + 2 * sizeof(float4) + static_cast<size_t>(p.wide_top_nodes) * 256;
    }
#endif
    // ... and its LDS-resident instance, with camera packets over the wide nodes: every wide node, the prepared triangles, material indices and materials
    // beside FOUR stack levels (the rest of a lane's stack in its global column: a work-group then takes 27 KB for the default scene and five fit a CU;
    // with eight levels 22 700, with four 25 100 Msamples/s; with camera packets 26 100-26 350 against the binary camera-packet kernel's 23 100:
    // tools/archive/ab_wide_resident.sh, profiles/r04_ab_wide_resident.txt)
    const uint32_t wr_levels_want = ctx->tune.bvh_stack_lds > 0 ? static_cast<uint32_t>(ctx->tune.bvh_stack_lds) : 4u;
    const size_t wide_resident_bytes = static_cast<size_t>(wr_levels_want) * rv::kBlock * 2 * sizeof(uint32_t) + 2 * sizeof(float4) + ctx->n_wide * rv::kWideTopQuads * 16 +
                                       ctx->n_tris * 64 + index_bytes + ctx->n_mats * 48;
    const bool wide_resident = bvh && bvh_resident && !ordered && l.regen && ctx->n_wide > 0 && ctx->bvh_wide == 1 && p.head_shift != 0 &&
                               ctx->tune.bvh_wide_resident == 1 && wide_resident_bytes <= 64 * 1024;
    if (wide_resident) {
        l.variant = 11u;
        l.kernel = generic ? rv::trace_bvh4_resident_generic : rv::trace_bvh4_resident;
        p.wide = ctx->d_wide;
        p.n_wide = static_cast<uint32_t>(ctx->n_wide);
        p.wide_top_nodes = p.n_wide;
        p.stack_levels = std::max<uint32_t>(1, ctx->wide_stack_levels);
        p.stack_lds_levels = std::min(p.stack_levels, wr_levels_want);
        l.lds = static_cast<size_t>(p.stack_lds_levels) * rv::kBlock * 2 * sizeof(uint32_t) + 2 * sizeof(float4) + ctx->n_wide * rv::kWideTopQuads * 16 + ctx->n_tris * 64 +
                index_bytes + ctx->n_mats * 48;
    }
    // the packet form of the resident brute-force kernel (rvpt_packets.hip): full packets of one kind per round, camera rays with the
    // packet-uniform early-out; the lean configuration only
    // (one sample per pixel — the headline's configuration — has an instance of its own: 15-word queue entries, six work-groups per CU instead of five)
    const uint32_t queue_words = (p.aa == 1) ? rv::kPacketQueueWordsAA1 : rv::kPacketQueueWords;
    const size_t packets_bytes = ctx->n_tris * 64 + index_bytes + ctx->n_mats * 48 + (rv::kBlock / 64) * queue_words * 64 * sizeof(uint32_t);
    const bool packets = !bvh && resident && ctx->n_tris > 0 && !generic && l.regen && p.max_bounces >= 1 && p.max_bounces <= 1023 &&
                         p.aa <= 1023 && ctx->n_mats <= rv::kResidentMaxMats && ctx->brute_packets_policy == 1 && packets_bytes <= 64 * 1024;
    if (packets) {
        l.variant = 6u;
        l.kernel = (p.aa == 1) ? rv::trace_brute_packets_aa1 : rv::trace_brute_packets;
        l.lds = packets_bytes;
        // + the screen rectangles (8 B per triangle, padded to 16) when they fit beside the rest
        const size_t rect_bytes = ((ctx->n_tris + 1) & ~size_t(1)) * 8;
        l.cull = ctx->packets_cull == 1 && l.lds + rect_bytes <= 64 * 1024;
        if (l.cull) l.lds += rect_bytes;
        // the bounce cull's premise: positions carry float errors of at most 2^-13 of the scene's scale — true while the camera (the origin of the first
        // segment) is no further than 64 scene scales from the world origin (rvpt_packets.hip: bounce_visibility)
        const float *o = ctx->camera.matrix + 12;
        const double far = rv::kBounceCameraScales * ctx->scene_scale;
        if (ctx->packets_bounce_cull == 1 && ctx->vis_words > 0 && std::fabs(o[0]) <= far && std::fabs(o[1]) <= far && std::fabs(o[2]) <= far) {
            p.vis = ctx->d_vis;
            p.vis_words = ctx->vis_words;
            p.vis_stride = (ctx->vis_words + 3u) & ~3u;
            if (ctx->packets_box_cull == 1 && ctx->d_leaf_boxes) p.leaf_boxes = ctx->d_leaf_boxes;
        }
    }
    if (bvh && ctx->force_stack_levels > 0) {  // tests: a stack smaller than the tree needs — the kernels clamp and report (RVPT_HIP_DEBUG)
        p.stack_levels = static_cast<uint32_t>(ctx->force_stack_levels);
        p.stack_lds_levels = std::min(p.stack_lds_levels, p.stack_levels);
    }
    const uint32_t blocks_needed = (p.n_work + rv::kBlock - 1) / rv::kBlock;
    l.grid = blocks_needed;  // one-pixel-per-lane kernel: one wave per 64 pixels
    if (l.regen) {           // persistent work-groups
        if (ctx->occ_kernel != reinterpret_cast<const void *>(l.kernel) || ctx->occ_lds != l.lds) {
            int q = 0;
            HIP_TRY(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&q, reinterpret_cast<const void *>(l.kernel), rv::kBlock, l.lds));
            ctx->occ_kernel = reinterpret_cast<const void *>(l.kernel);
            ctx->occ_lds = l.lds;
            ctx->occ_per_cu = q;
        }
        int per_cu = std::max(1, std::min(ctx->occ_per_cu, 8));
        // With frames overlapped in flight a frame kernel takes only 2 work-groups per CU: the kernels of consecutive
        // frames then co-reside (2 + 2 waves per SIMD) and a frame's tail hides under the next frame's body (swept on
        // MI355X: profiles/README.md).  The HBM-resident BVH kernels take 3-4 per CU: the register file holds 7 waves
        // per SIMD in all, so the kernels of the frames in flight fill the CU between them either way, and larger
        // per-group shares balance better (swept: profiles/r01_bvh_knob_sweeps.txt).
        // (LDS-resident BVH with several frames per launch: 3 measured 6 % better than 2; one frame per launch: 2.)
        // Brute force with several frames per launch (>= 4): a launch is long against its own ramp-up and drain, so it takes
        // the register file's five work-groups per CU and consecutive launches overlap only at their ends (swept on MI355X,
        // tools/archive/sweep_batch_bpc.sh: 8 frames per launch x 5 per CU = 6 080 / 6 500 Msamples/s over 20 / 200 frames against
        // 5 670 / 6 410 for one frame per launch x 2 per CU x 3 launches in flight).
        const bool batched = p.n_work >= 4 * p.n_work_frame;
        const int small_per_cu = batched ? (bvh ? 3 : 8) : 2;  // (brute force, batched: what LDS and registers allow — five work-groups per CU, six for the aa == 1 instance)
        if (ctx->overlap) per_cu = std::min(per_cu, (bvh && !bvh_resident) ? (ctx->tune.blocks_per_cu ? bvh_per_cu : (l.slots > 3 ? 2 : bvh_per_cu)) : small_per_cu);
        // ... which assumes that launches overlap.  A LONE launch of the HBM-resident BVH kernel — nothing of this context in flight when it goes
        // out: a rank's 20-step share sent as one launch, the first launch of a burst — has nobody to share the CU with and takes what the registers
        // allow (six per CU): rank 2's share of an 8-way partition as one 20-frame launch, C4 geometry 0.132 -> 0.079 ms per frame, C3 0.675 -> 0.519
        // (tools/archive/sweep_share_shapes.sh, profiles/r04_share_shapes.txt); launches that follow while it runs keep the overlapping shape
        // Only launches of >= 16 frames: a caller that sends such a launch has batched what it had; the first of a stream of SMALLER launches must
        // leave room for the next (measured: C3 at 8 frames per launch 3 240 -> 3 110 Msamples/s when the first launch took the whole CU).
        if (ctx->overlap && bvh && !bvh_resident && !ctx->tune.blocks_per_cu && l.lone && p.n_work >= 16u * p.n_work_frame)
            per_cu = std::max(1, std::min(ctx->occ_per_cu, 8));
        if (ctx->tune.blocks_per_cu) per_cu = ctx->tune.blocks_per_cu;
        l.grid = std::min<uint32_t>(blocks_needed, static_cast<uint32_t>(ctx->num_cus) * static_cast<uint32_t>(per_cu));
    }
    p.n_waves = l.grid * (rv::kBlock / 64);
    return 0;
}

// work distribution of the launch (kernels: WavePool): a static first chunk per wave, then sharded claims
// align_units > 1 (the packet kernel: 4 units = the 64 work items of a camera round): chunks start at multiples of it wherever the launch is large enough for
// full-size chunks, so that a camera round is ONE 16 x 4 pixel block (the kernel checks, and skips its rectangle cull for a round that is not)
void plan_work(const rvpt_hip_ctx *ctx, bool regen, rv::FrameParams &p, uint32_t align_units = 1)
{
    p.n_units = p.n_work / rv::kUnit;
    if (!regen) {
        p.first_units = 64 / rv::kUnit;  // 64 pixels per wave, nothing dynamic
        p.claim_units = 1;
    } else {
        // 128-pixel static chunk per wave (less if there is not that much work), 128-pixel claims after that
        p.first_units = std::max(1u, std::min(rv::kMaxClaimUnits, (p.n_units + p.n_waves - 1) / p.n_waves));
        p.claim_units = rv::kMaxClaimUnits;
        // the packet kernel with its culls (round 5) finishes a 64-pixel block of sky in about a microsecond: 128-pixel claims then ask the eight counters for
        // more than they can hand out (one L2 word sustains ~90 atomics per microsecond), and the CLAIMS bound the kernel — 512-pixel claims: 23 979 -> 37 966
        // Msamples/s at the driver's command, 36 169 -> 46 338 over 200 steps; an eighth of the image 0.0164 -> 0.0131 ms per frame; 16 / 32 / 64 counters
        // instead of 8 help only the small claims (tools/archive/r05_claims.sh, profiles/r05_claims.txt)
        if (align_units == 4u) p.claim_units = 32u;
        // a static chunk of 5-7 units (images of 131-229 k pixels) would start most camera rounds off a block boundary and lose them the rectangles
        if (align_units > 1 && p.first_units > align_units) p.first_units = p.first_units / align_units * align_units;
        if (ctx->tune.first_units) p.first_units = static_cast<uint32_t>(ctx->tune.first_units);
        if (ctx->tune.claim_units) p.claim_units = static_cast<uint32_t>(ctx->tune.claim_units);
    }
    p.dyn_base = static_cast<uint32_t>(std::min<uint64_t>(p.n_units, static_cast<uint64_t>(p.first_units) * p.n_waves));
    p.shard_len = (p.n_units - p.dyn_base + rv::kClaimShards - 1) / rv::kClaimShards;
    if (align_units > 1) p.shard_len = (p.shard_len + align_units - 1) / align_units * align_units;  // (the kernel clips a shard at n_units)
}

// The packet kernel's claim order over a frame's blocks (FrameParams::perm_*): only when every chunk of the plan is whole blocks (so that a camera round is one
// block of the order), the frame is whole groups, and the products of the map stay inside 32 bits.  Launches of fewer than four frames only: a wave of a one-frame
// launch holds its static chunk and ONE or two 512-item claims, which in tile-linear order are all sky or all model — the waves on the sky leave at 45 us, those on
// the model at 80-115 (profiles/r06_interleave.txt); dealt from all over the frame every claim costs about the same and the launch is 14 % shorter (one frame per
// launch 38 300 -> 42 300 Msamples/s).  A 20-frame launch gives every wave thirteen claims, which even out by themselves: there the order measured 2-4 % slower.
// (the ctx-free core: rvpt_claim_order of the laboratory build walks it without a GPU)
bool plan_claim_order(const uint32_t n_work_frame, const uint32_t g, uint32_t &groups_out, uint32_t &stride_out, uint32_t &shift_out)
{
    if (g == 0 || (g & (g - 1u)) != 0u || g > 8u) return false;
    const uint32_t blocks = n_work_frame / 64u;
    if (n_work_frame % (64u * g) || blocks / g < 16u) return false;
    const uint32_t groups = blocks / g;
    // the stride: nearest below the golden section of `groups` that is coprime to it (consecutive multiples of it mod `groups` are a low-discrepancy sequence:
    // any run of the order samples the frame evenly)
    auto gcd = [](uint32_t a, uint32_t b) {
        while (b) {
            const uint32_t t = a % b;
            a = b;
            b = t;
        }
        return a;
    };
    // ... = groups / (k + 0.618...) for the smallest k = 1, 2, ... that keeps (group index) * stride inside 32 bits: k = 1 up to ~82 000 groups (1920 x 1080 has
    // 32 640 blocks), a 3840 x 2160 frame takes k = 4; beyond k = 16 (frames of more than ~17 M pixels) the order stays tile-linear
    const uint64_t room = 0xFFFFFFFFull / groups;
    uint32_t stride = 0;
    for (uint32_t k = 1; k <= 16u && stride == 0u; ++k) {
        const uint32_t want = static_cast<uint32_t>(groups / (k + 0.6180339887498949));
        if (want <= room) stride = want;
    }
    while (stride > 1u && gcd(stride, groups) != 1u) stride -= 1u;
    if (stride <= 1u) return false;
    uint32_t shift = 0;
    while ((1u << shift) < g) shift += 1;
    groups_out = groups;
    stride_out = stride;
    shift_out = shift;
    return true;
}

void plan_interleave(const rvpt_hip_ctx *ctx, rv::FrameParams &p)
{
    p.perm_groups = 0;
    p.perm_stride = 1;
    p.perm_shift = 0;
    p.div_perm_groups = rv::fast_div_make(1);
    const uint32_t g = static_cast<uint32_t>(ctx->packets_interleave);
    if (g == 0 || p.first_units % 4u || p.claim_units % 4u || p.dyn_base % 4u || p.shard_len % 4u) return;
    if (!ctx->packets_interleave_all && p.n_work >= 4u * p.n_work_frame) return;
    uint32_t groups, stride, shift;
    if (!plan_claim_order(p.n_work_frame, g, groups, stride, shift)) return;
    p.perm_groups = groups;
    p.perm_stride = stride;
    p.perm_shift = shift;
    p.div_perm_groups = rv::fast_div_make(groups);
}

}  // namespace

extern "C" {

int rvpt_hip_abi_version(void) { return RVPT_HIP_ABI_VERSION; }

uint32_t rvpt_hip_build_flags(void) { return (RVPT_HIP_LAB ? RVPT_HIP_BUILD_LAB : 0u) | (RV_REPORT_STACK_OVERFLOW ? RVPT_HIP_BUILD_DEBUG_CHECKS : 0u); }

int rvpt_hip_device_count(int *count)
{
    if (!count) return fail(nullptr, RVPT_HIP_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(nullptr, RVPT_HIP_ERR_NO_DEVICE, "hipGetDeviceCount -> %s", hipGetErrorString(e));
    }
    *count = n;
    return RVPT_HIP_OK;
}

int rvpt_hip_create(rvpt_hip_ctx **out, int device_id, uint32_t width, uint32_t height, uint32_t tile_rank,
                    uint32_t tile_world, uint32_t flags)
{
    if (!out) return fail(nullptr, RVPT_HIP_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (width == 0 || height == 0 || tile_world == 0 || tile_rank >= tile_world)
        return fail(nullptr, RVPT_HIP_ERR_INVALID, "bad geometry %ux%u rank %u/%u", width, height, tile_rank, tile_world);
    if (static_cast<uint64_t>(width) * height > 0x7FFFFFFFull) return fail(nullptr, RVPT_HIP_ERR_INVALID, "image too large");
    if ((flags & RVPT_HIP_TRAVERSAL_MASK) == RVPT_HIP_TRAVERSAL_MASK) return fail(nullptr, RVPT_HIP_ERR_INVALID, "unknown traversal mode in flags");
    if (flags & ~static_cast<uint32_t>(RVPT_HIP_FLAGS_KNOWN))
        return fail(nullptr, RVPT_HIP_ERR_INVALID, "unknown bits 0x%x in flags (0x40 / 0x80 / 0x100 were the wavefront pipelines of ABI 3-4, retired in ABI 5)", flags & ~static_cast<uint32_t>(RVPT_HIP_FLAGS_KNOWN));
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) return fail(nullptr, RVPT_HIP_ERR_NO_DEVICE, "no HIP device visible");
    if (device_id < 0 || device_id >= n_dev) return fail(nullptr, RVPT_HIP_ERR_INVALID, "device %d out of range (%d devices)", device_id, n_dev);

    {  // one line, once per process: the seven streams of a context want eight hardware queues (include/rvpt_hip.h, PROCESS ENVIRONMENT)
        static bool noted = false;
        const char *q = getenv("GPU_MAX_HW_QUEUES");
        if (!noted && !getenv("RVPT_HIP_QUIET") && (!q || atoi(q) < 8)) {
            noted = true;
            std::fprintf(stderr, "[rvpt_hip] note: GPU_MAX_HW_QUEUES is %s; launches in flight share hardware queues below 8 (measured -10 %%). "
                                 "Set GPU_MAX_HW_QUEUES=8 before the first HIP call of the process (RVPT_HIP_QUIET=1 silences this note).\n", q ? q : "unset");
        }
    }
    rvpt_hip_ctx *ctx = new (std::nothrow) rvpt_hip_ctx;
    if (!ctx) return fail(nullptr, RVPT_HIP_ERR_HIP, "out of host memory");
    ctx->device = device_id;
    ctx->width = width;
    ctx->height = height;
    ctx->tiles_x = (width + RVPT_HIP_TILE - 1) / RVPT_HIP_TILE;
    ctx->tiles_y = (height + RVPT_HIP_TILE - 1) / RVPT_HIP_TILE;
    ctx->tile_rank = tile_rank;
    ctx->tile_world = tile_world;
    ctx->flags = flags;
    ctx->timing = (flags & RVPT_HIP_TIMING) != 0;
    ctx->n_local_tiles = owned_tiles(ctx->tiles_x * ctx->tiles_y, tile_rank, tile_world);
    ctx->n_work = ctx->n_local_tiles * 256u;

    auto bail = [&](int code) {
        std::string keep = ctx->err;
        rvpt_hip_destroy(ctx);
        g_err = keep;
        return code;
    };
#define CREATE_TRY(expr)                                                                                      \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) {                                                                               \
            fail(ctx, RVPT_HIP_ERR_HIP, "%s -> %s", #expr, hipGetErrorString(e_));                            \
            return bail(RVPT_HIP_ERR_HIP);                                                                    \
        }                                                                                                     \
    } while (0)
    CREATE_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    CREATE_TRY(hipGetDeviceProperties(&prop, device_id));
    ctx->num_cus = prop.multiProcessorCount;
    CREATE_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    // every rank allocates the largest slot (rank 0's) so that the gather payload has one size
    const size_t slot_quads = std::max<size_t>(static_cast<size_t>(owned_tiles(ctx->tiles_x * ctx->tiles_y, 0, tile_world)) * 256u, 1);
    ctx->slot_quads = slot_quads;
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_accum), slot_quads * sizeof(float4)));
    CREATE_TRY(hipMemsetAsync(ctx->d_accum, 0, slot_quads * sizeof(float4), ctx->stream));
    if (const char *e = getenv("RVPT_HIP_FRAMES_IN_FLIGHT")) {
        ctx->n_slots = std::max(1, std::min(atoi(e), int(rvpt_hip_ctx::kMaxSlots)));
        ctx->slots_fixed = true;
    }
    ctx->overlap = ctx->n_slots > 1 && getenv("RVPT_HIP_NO_OVERLAP") == nullptr;
    if (!ctx->overlap) ctx->n_slots = 1;
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_counter), ctx->n_slots * rv::kCounterWords * sizeof(unsigned long long)));
    CREATE_TRY(hipMemsetAsync(ctx->d_counter, 0, ctx->n_slots * rv::kCounterWords * sizeof(unsigned long long), ctx->stream));
    for (int i = 0; i < ctx->n_slots; ++i) {
        CREATE_TRY(hipStreamCreateWithFlags(&ctx->trace_stream[i], hipStreamNonBlocking));
        CREATE_TRY(hipEventCreateWithFlags(&ctx->trace_done[i], hipEventDisableTiming));
        for (int b = i * rvpt_hip_ctx::kBufsPerSlot; b < (i + 1) * rvpt_hip_ctx::kBufsPerSlot; ++b) {
            CREATE_TRY(hipEventCreateWithFlags(&ctx->blend_done[b], hipEventDisableTiming));
            if (ctx->overlap) {
                CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_samples[b]), slot_quads * sizeof(rv::SampleRGB)));
                CREATE_TRY(hipMemsetAsync(ctx->d_samples[b], 0, slot_quads * sizeof(rv::SampleRGB), ctx->stream));
                ctx->samples_cap[b] = 1;
            }
        }
    }
    if (const char *tl = lab_env("RVPT_HIP_TIMELINE")) ctx->timeline_path = tl;
    ctx->bvh_wide = (flags & RVPT_HIP_BVH_PER_LANE) ? 0 : 1;
    if (const char *e = lab_env("RVPT_HIP_BVH_WIDE")) ctx->bvh_wide = atoi(e) > 0 ? 1 : 0;  // experiments: A/B a whole run
    if (const char *e = lab_env("RVPT_HIP_BVH_WIDE8")) ctx->bvh_wide8 = atoi(e) > 0 ? 1 : 0;
    if (const char *e = lab_env("RVPT_HIP_BVH_QUANT")) ctx->bvh_quant = atoi(e) > 0 ? 1 : 0;
    ctx->brute_packets_policy = (flags & RVPT_HIP_BRUTE_MIXED_PACKETS) ? 0 : 1;
    if (const char *e = lab_env("RVPT_HIP_BRUTE_PACKETS")) ctx->brute_packets_policy = atoi(e) > 0 ? 1 : 0;
    if (const char *e = getenv("RVPT_HIP_DEBUG")) ctx->debug_checks = atoi(e) > 0 ? 1 : 0;
    if (ctx->debug_checks && !RV_REPORT_STACK_OVERFLOW) {  // never a silent no-op: the release kernels only clamp, nothing would ever set the word
        fail(ctx, RVPT_HIP_ERR_UNSUPPORTED, "RVPT_HIP_DEBUG=1 needs the kernels' internal checks, which this build of the library does not carry "
                                            "(rvpt_hip_build_flags() bit 1): load librvpt_hip_debug.so");
        return bail(RVPT_HIP_ERR_UNSUPPORTED);
    }
    if (const char *e = lab_env("RVPT_HIP_BVH_FORCE_STACK_LEVELS")) ctx->force_stack_levels = std::max(0, atoi(e));
    if (const char *e = getenv("RVPT_HIP_PACKETS_CULL")) ctx->packets_cull = atoi(e) > 0 ? 1 : 0;
    if (const char *e = getenv("RVPT_HIP_PACKETS_BOUNCE_CULL")) ctx->packets_bounce_cull = atoi(e) > 0 ? 1 : 0;
    if (const char *e = getenv("RVPT_HIP_PACKETS_BOX_CULL")) ctx->packets_box_cull = atoi(e) > 0 ? 1 : 0;
    if (const char *e = lab_env("RVPT_HIP_PACKETS_LEAN_INSTANCE")) ctx->packets_lean_instance = atoi(e) > 0 ? 1 : 0;
    if (const char *e = getenv("RVPT_HIP_PACKETS_INTERLEAVE")) {
        const int g = atoi(e);
        ctx->packets_interleave_all = g < 0 ? 1 : 0;
        ctx->packets_interleave = (std::abs(g) == 1 || std::abs(g) == 2 || std::abs(g) == 4 || std::abs(g) == 8) ? std::abs(g) : 0;
    }
    auto env_int = [](const char *name, int lo, int hi) {
        const char *e = lab_env(name);
        return e ? std::max(lo, std::min(hi, atoi(e))) : 0;
    };
    ctx->tune.blocks_per_cu = env_int("RVPT_HIP_BLOCKS_PER_CU", 1, 8);
    ctx->tune.first_units = env_int("RVPT_HIP_FIRST_UNITS", 1, 1 << 20);
    ctx->tune.claim_units = env_int("RVPT_HIP_CLAIM_UNITS", 1, 1 << 20);
    ctx->tune.bvh_refill = env_int("RVPT_HIP_BVH_REFILL", 1, 64);
    ctx->tune.bvh_leaf_batch = env_int("RVPT_HIP_BVH_LEAF_BATCH", 1, 64);
    ctx->tune.bvh_stack_lds = env_int("RVPT_HIP_BVH_STACK_LDS", 1, 64);
    if (const char *e = lab_env("RVPT_HIP_BVH_WIDE_RESIDENT")) ctx->tune.bvh_wide_resident = atoi(e) > 0 ? 1 : 0;
    ctx->tune.bvh_no_resident = env_int("RVPT_HIP_BVH_NO_RESIDENT", 0, 1);
    ctx->tune.bvh_cam_min = env_int("RVPT_HIP_BVH_CAM_MIN", 1, 65);  // 65 = never
    if (const char *e = lab_env("RVPT_HIP_BVH_DETACH")) ctx->tune.bvh_detach = std::max(0, std::min(64, atoi(e)));
    if (const char *e = lab_env("RVPT_HIP_BVH_TOP_NODES")) ctx->tune.bvh_top_nodes = std::max(0, std::min(2048, atoi(e)));  // 0 = no LDS copy
    CREATE_TRY(hipMalloc(reinterpret_cast<void **>(&ctx->d_stats), rv::kStatStripes * rv::kStatStride * sizeof(unsigned long long)));
    CREATE_TRY(hipMemsetAsync(ctx->d_stats, 0, rv::kStatStripes * rv::kStatStride * sizeof(unsigned long long), ctx->stream));
    CREATE_TRY(hipStreamSynchronize(ctx->stream));
#undef CREATE_TRY
    *out = ctx;
    return RVPT_HIP_OK;
}

void rvpt_hip_destroy(rvpt_hip_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    drop_comm(ctx);
    for (int i = 0; i < rvpt_hip_ctx::kMaxSlots; ++i)
        if (ctx->d_stack_overflow[i]) (void)hipFree(ctx->d_stack_overflow[i]);
    if (ctx->comm_stream && !ctx->comm_stream_lost) (void)hipStreamDestroy(ctx->comm_stream);  // (a lost one still holds a collective that never completes)
    for (int i = 0; i < rvpt_hip_ctx::kMaxSlots; ++i) {
        if (ctx->d_rects[i]) (void)hipFree(ctx->d_rects[i]);
        if (ctx->d_cam_records[i]) (void)hipFree(ctx->d_cam_records[i]);
    }
    if (ctx->d_vis) (void)hipFree(ctx->d_vis);
    if (ctx->d_leaf_boxes) (void)hipFree(ctx->d_leaf_boxes);
    if (ctx->d_wide8) (void)hipFree(ctx->d_wide8);
    if (ctx->d_leaf_box) (void)hipFree(ctx->d_leaf_box);
    if (ctx->d_wideq) (void)hipFree(ctx->d_wideq);
    if (ctx->d_gather) (void)hipFree(ctx->d_gather);
    if (ctx->d_barrier) (void)hipFree(ctx->d_barrier);
    if (ctx->d_quant) (void)hipFree(ctx->d_quant);
    for (int i = 0; i < rvpt_hip_ctx::kMaxSlots; ++i)
        if (ctx->trace_stream[i]) (void)hipStreamSynchronize(ctx->trace_stream[i]);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->d_timeline && !ctx->timeline_path.empty()) {  // debugging aid: dump the last frame's wave timeline
        std::vector<unsigned long long> h(ctx->timeline_words);
        if (hipMemcpy(h.data(), ctx->d_timeline, ctx->timeline_words * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            if (FILE *f = fopen(ctx->timeline_path.c_str(), "wb")) {
                fwrite(h.data(), 8, h.size(), f);
                fclose(f);
            }
        }
        (void)hipFree(ctx->d_timeline);
    }
    for (auto &pr : ctx->pending) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    for (auto &pr : ctx->spare) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    for (int i = 0; i < rvpt_hip_ctx::kMaxSlots; ++i) {
        if (ctx->trace_stream[i]) (void)hipStreamSynchronize(ctx->trace_stream[i]);
        if (ctx->trace_done[i]) (void)hipEventDestroy(ctx->trace_done[i]);
        for (int b = i * rvpt_hip_ctx::kBufsPerSlot; b < (i + 1) * rvpt_hip_ctx::kBufsPerSlot; ++b) {
            if (ctx->blend_done[b]) (void)hipEventDestroy(ctx->blend_done[b]);
            if (ctx->d_samples[b]) (void)hipFree(ctx->d_samples[b]);
        }
        if (ctx->trace_stream[i]) (void)hipStreamDestroy(ctx->trace_stream[i]);
    }
    void *bufs[] = {ctx->d_tris, ctx->d_prep, ctx->d_mats, ctx->d_nodes, ctx->d_wide, ctx->d_mat_index, ctx->d_unit_n, ctx->d_accum,
                    ctx->d_rowmajor, ctx->d_counter, ctx->d_stats};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int rvpt_hip_upload_scene(rvpt_hip_ctx *ctx, const rvpt_bvh_node *nodes, size_t n_nodes, const rvpt_triangle *tris,
                          size_t n_tris, const rvpt_material *mats, size_t n_mats)
{
    if (!ctx) return fail(nullptr, RVPT_HIP_ERR_INVALID, "ctx is NULL");
    if ((n_tris && !tris) || (n_mats && !mats)) return fail(ctx, RVPT_HIP_ERR_INVALID, "NULL scene array");
    if (n_tris > 0x3FFFFFFFull) return fail(ctx, RVPT_HIP_ERR_INVALID, "too many triangles");
    // an EMPTY scene has no tree (RVPT::initialize with no triangles): every ray misses whatever the traversal, and the
    // frame kernels of a BVH context then run the brute-force instance over zero triangles (choose_launch)
    const bool bvh = (ctx->flags & RVPT_HIP_TRAVERSAL_MASK) != RVPT_HIP_TRAVERSAL_BRUTE && n_tris > 0;
    uint32_t bvh_height_tmp = 0;
    // materials[int(mat_id.x)] (intersection.glsl:398) must stay inside the buffer
    for (size_t i = 0; i < n_tris; ++i) {
        const float m = tris[i].mat_id[0];
        if (!(m >= 0.0f) || static_cast<size_t>(static_cast<int>(m)) >= n_mats)
            return fail(ctx, RVPT_HIP_ERR_INVALID, "triangle %zu: material index %g outside [0,%zu)", i, static_cast<double>(m), n_mats);
    }
    if (bvh) {
        if (!nodes || n_nodes == 0) return fail(ctx, RVPT_HIP_ERR_INVALID, "BVH context needs nodes");
        for (size_t i = 0; i < n_nodes; ++i) {
            const rvpt_bvh_node &nd = nodes[i];
            if (nd.primitive_count > 0) {
                if (static_cast<uint64_t>(nd.first_child_or_primitive) + nd.primitive_count > n_tris)
                    return fail(ctx, RVPT_HIP_ERR_INVALID, "node %zu: leaf range outside the triangle buffer", i);
            } else if (static_cast<uint64_t>(nd.first_child_or_primitive) + 1 >= n_nodes) {
                return fail(ctx, RVPT_HIP_ERR_INVALID, "node %zu: child index outside the node buffer", i);
            }
        }
        // height of the tree = what the traversal stack must hold; also rejects cycles
        std::vector<std::pair<uint32_t, uint32_t>> work{{0u, 1u}};
        uint32_t height = 0;
        size_t visited = 0;
        while (!work.empty()) {
            const auto [idx, depth] = work.back();
            work.pop_back();
            if (++visited > n_nodes) return fail(ctx, RVPT_HIP_ERR_INVALID, "BVH is not a tree (node reachable twice)");
            height = std::max(height, depth);
            if (nodes[idx].primitive_count == 0) {
                work.emplace_back(nodes[idx].first_child_or_primitive, depth + 1);
                work.emplace_back(nodes[idx].first_child_or_primitive + 1, depth + 1);
            }
        }
        // the reference's uint stack[64] holds its sentinel + one push per inner node of a root-to-leaf path: a leaf may sit
        // on level 64 (root = level 1); deeper trees overflow it there (undefined) and are rejected here
        if (height > rv::kBvhStackDepth)
            return fail(ctx, RVPT_HIP_ERR_INVALID, "BVH height %u exceeds the %u levels the reference's 64-entry traversal stack can walk (intersection.glsl:363-372)", height, rv::kBvhStackDepth);
        bvh_height_tmp = height;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc0 = sync_all(ctx)) return rc0;  // frames in flight still read the old scene
    int rc;
    if ((rc = grow(ctx, ctx->d_tris, ctx->cap_tris, n_tris, sizeof(rvpt_triangle)))) return rc;
    if ((rc = grow(ctx, ctx->d_prep, ctx->cap_prep, n_tris, sizeof(rvpt_triangle)))) return rc;
    if ((rc = grow(ctx, ctx->d_mat_index, ctx->cap_mat_index, n_tris, sizeof(uint32_t)))) return rc;
    if ((rc = grow(ctx, ctx->d_unit_n, ctx->cap_unit_n, n_tris, sizeof(float4)))) return rc;
    if ((rc = grow(ctx, ctx->d_mats, ctx->cap_mats, n_mats, sizeof(rvpt_material)))) return rc;
    if (n_tris) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_tris, tris, n_tris * sizeof(rvpt_triangle), hipMemcpyHostToDevice, ctx->stream));
    if (n_mats) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_mats, mats, n_mats * sizeof(rvpt_material), hipMemcpyHostToDevice, ctx->stream));
    if (n_mats) {  // the device copy's data.w = 1 / ior ("Unused" in the reference: structs.glsl:31)
        hipLaunchKernelGGL(rv::prepare_materials, dim3((static_cast<uint32_t>(n_mats) + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_mats, static_cast<uint32_t>(n_mats));
        HIP_TRY(ctx, hipGetLastError());
    }
    std::vector<rvpt_bvh_node> device_nodes;  // must outlive the async copy below (stream is synchronised before return)
    size_t n_device_nodes = 0;
    if (bvh) {
        // Device layout of the tree (traversal order and results are unchanged): breadth-first, root at 0, slot 1
        // unused, every sibling pair on an even index = one 64-byte line, upper levels first.
        if (lab_env("RVPT_HIP_BVH_CALLER_LAYOUT")) {
            device_nodes.assign(nodes, nodes + n_nodes);
            if (device_nodes.size() < 2) device_nodes.resize(2);  // the kernels copy at least the root's 64-byte line into LDS
        } else {
            device_nodes.resize(n_nodes + 1);
            std::memset(device_nodes.data(), 0, device_nodes.size() * sizeof(rvpt_bvh_node));
            std::vector<std::pair<uint32_t, uint32_t>> queue;  // (caller index, device index), FIFO
            queue.reserve(n_nodes);
            queue.emplace_back(0u, 0u);
            uint32_t next_pair = 2;
            for (size_t head = 0; head < queue.size(); ++head) {
                const auto [src, dst] = queue[head];
                rvpt_bvh_node nd = nodes[src];
                if (nd.primitive_count == 0) {
                    const uint32_t child = nd.first_child_or_primitive;
                    nd.first_child_or_primitive = next_pair;
                    queue.emplace_back(child, next_pair);
                    queue.emplace_back(child + 1, next_pair + 1);
                    next_pair += 2;
                }
                device_nodes[dst] = nd;
            }
        }
        n_device_nodes = device_nodes.size();
        if ((rc = grow(ctx, ctx->d_nodes, ctx->cap_nodes, n_device_nodes, sizeof(rvpt_bvh_node)))) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_nodes, device_nodes.data(), n_device_nodes * sizeof(rvpt_bvh_node), hipMemcpyHostToDevice, ctx->stream));
    }
    if (n_tris) {
        const uint32_t n = static_cast<uint32_t>(n_tris);
        hipLaunchKernelGGL(rv::prepare_triangles, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_tris, n, ctx->d_prep, ctx->d_mat_index, ctx->d_unit_n);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // caller may free its arrays on return
    ctx->n_tris = n_tris;
    ctx->n_mats = n_mats;
    ctx->n_nodes = bvh ? n_device_nodes : 0;
    ctx->bvh_height = bvh_height_tmp;
    ctx->bvh_head_shift = 0;
    if (bvh) {  // can a node's (first, count) pair ride in one stack word?  indices below 2^shift, leaf sizes below 2^(32 - shift)
        uint32_t shift = 1, max_count = 0;
        while (shift < 31 && (1ull << shift) <= std::max(n_device_nodes, n_tris)) shift += 1;
        for (size_t i = 0; i < n_nodes; ++i) max_count = std::max(max_count, nodes[i].primitive_count);
        if (static_cast<uint64_t>(max_count) < (1ull << (32 - shift))) ctx->bvh_head_shift = shift;
    }
    ctx->n_wide = 0;
    ctx->wide_stack_levels = 0;
    if (bvh && !lab_env("RVPT_HIP_BVH_CALLER_LAYOUT")) {  // the 4-wide form of the tree (breadth-first device layout: children of node i at first, first + 1)
        uint32_t need = 0;
        const std::vector<float> wide = rv::build_wide_nodes(device_nodes.data(), device_nodes.size(), ctx->bvh_head_shift, need);
        if (!wide.empty() && need <= 4096u && wide.size() / 32 < rv::kWideMaxNodes) {
            const size_t n_wide = wide.size() / 32;
            if ((rc = grow(ctx, ctx->d_wide, ctx->cap_wide, n_wide * 8, sizeof(float4)))) return rc;
            HIP_TRY(ctx, hipMemcpy(ctx->d_wide, wide.data(), wide.size() * sizeof(float), hipMemcpyHostToDevice));
            ctx->n_wide = n_wide;
            ctx->wide_stack_levels = need;
        }
        ctx->has_quant = false;
#if RVPT_HIP_LAB
        if (ctx->n_wide > 0 && ctx->bvh_quant == 1) {
            float extent = 0.0f, box_extent = 0.0f;
            const std::vector<uint32_t> quant = rv::build_quant_nodes(wide, extent);
            const std::vector<float> boxes = quant.empty() ? std::vector<float>() : rv::build_leaf_boxes(device_nodes.data(), device_nodes.size(), n_tris, box_extent);
            if (!quant.empty() && !boxes.empty()) {
                if ((rc = grow(ctx, ctx->d_wideq, ctx->cap_wideq, ctx->n_wide * 4, sizeof(float4)))) return rc;
                if ((rc = grow(ctx, ctx->d_leaf_box, ctx->cap_leaf_box, n_tris * 2, sizeof(float4)))) return rc;
                HIP_TRY(ctx, hipMemcpy(ctx->d_wideq, quant.data(), quant.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
                HIP_TRY(ctx, hipMemcpy(ctx->d_leaf_box, boxes.data(), boxes.size() * sizeof(float), hipMemcpyHostToDevice));
                ctx->has_quant = true;
                ctx->slab_extent = std::max(extent, box_extent);
            }
        }
        ctx->n_wide8 = 0;
        if (ctx->bvh_wide8 == 1) {
            uint32_t need8 = 0;
            const std::vector<float> wide8 = rv::build_wide8_nodes(device_nodes.data(), device_nodes.size(), ctx->bvh_head_shift, need8);
            if (!wide8.empty() && need8 <= 4096u && wide8.size() / 64 < (rv::kWideMaxNodes >> 1)) {
                const size_t n8 = wide8.size() / 64;
                if ((rc = grow(ctx, ctx->d_wide8, ctx->cap_wide8, n8 * 16, sizeof(float4)))) return rc;
                HIP_TRY(ctx, hipMemcpy(ctx->d_wide8, wide8.data(), wide8.size() * sizeof(float), hipMemcpyHostToDevice));
                ctx->n_wide8 = n8;
                ctx->wide8_stack_levels = need8;
            }
        }
#endif
    }
    // the bounce cull's table, for scenes the packet kernel can hold (brute-force contexts, <= kResidentMaxTris triangles)
    ctx->vis_words = 0;
    ctx->scene_scale = 0.0;
    if (!bvh && n_tris > 0 && n_tris <= rv::kResidentMaxTris) {
        const double scale = rv::bounce_scene_scale(reinterpret_cast<const float *>(tris), n_tris);
        if (scale > 0.0) {
            const uint32_t n = static_cast<uint32_t>(n_tris), words = (n + 31u) / 32u;
            const uint32_t stride = (words + 3u) & ~3u;  // rows 16-byte aligned and a multiple of four words apart: a bounce round loads four words of a lane's row at once
            const size_t total = static_cast<size_t>(2) * n * stride;
            if ((rc = grow(ctx, ctx->d_vis, ctx->vis_cap, total, sizeof(uint32_t)))) return rc;
            hipLaunchKernelGGL(rv::bounce_visibility, dim3(static_cast<uint32_t>((total + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_prep, n, rv::kBounceMarginScales * scale, words,
                               stride, ctx->d_vis);
            HIP_TRY(ctx, hipGetLastError());
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            ctx->vis_words = words;
            ctx->scene_scale = scale;
            // ... and the leaf boxes of the same triangles (host arithmetic on the caller's array; rvpt_vis.h)
            // (a whole word's boxes — 32 / kLeafTris of them — are requested together: the array is padded to whole words, the padding's triangles do not exist)
            const size_t n_leaves = static_cast<size_t>(words) * (32u / rv::kLeafTris);
            std::vector<float> boxes(8 * n_leaves, 0.0f);
            rv::bounce_leaf_boxes(reinterpret_cast<const float *>(tris), n_tris, scale, boxes.data());
            if ((rc = grow(ctx, ctx->d_leaf_boxes, ctx->leaf_boxes_cap, 2 * n_leaves, sizeof(float4)))) return rc;
            HIP_TRY(ctx, hipMemcpy(ctx->d_leaf_boxes, boxes.data(), boxes.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    ctx->scene_gen += 1;  // the slots' screen rectangles belong to the old scene
    ctx->have_scene = true;
    return RVPT_HIP_OK;
}

int rvpt_hip_set_frame(rvpt_hip_ctx *ctx, const rvpt_render_settings *s, const rvpt_camera_data *cam)
{
    if (!ctx) return fail(nullptr, RVPT_HIP_ERR_INVALID, "ctx is NULL");
    if (!s || !cam) return fail(ctx, RVPT_HIP_ERR_INVALID, "NULL settings/camera");
    if (s->aa < 1) return fail(ctx, RVPT_HIP_ERR_INVALID, "aa must be >= 1 (got %d)", s->aa);
    // every render mode (0..9, anything else = integrator_Hart) and camera mode (0, 1, anything else = spherical)
    // of compute_pass.comp:68-118 is implemented
    ctx->settings = *s;
    ctx->camera = *cam;
    ctx->have_frame = true;
    return RVPT_HIP_OK;
}

namespace {

// one launch covering n_frames consecutive frames starting at settings.current_frame
static int dispatch_launch(rvpt_hip_ctx *ctx, uint32_t n_frames)
{
    const int slots = slots_for(ctx, n_frames);  // launches in flight for this kind of launch; no drain when it changes: a slot's
    if (ctx->next_slot >= slots) ctx->next_slot = 0;  // reuse waits for its own last blend, and the blends are enqueued in dispatch order
    const int slot = ctx->next_slot;
    ctx->next_slot = (slot + 1) % slots;
    ctx->last_slots = slots;
    hipStream_t tstream = ctx->overlap ? ctx->trace_stream[slot] : ctx->stream;
    const int buf = slot * rvpt_hip_ctx::kBufsPerSlot + ctx->slot_parity[slot];
    ctx->slot_parity[slot] ^= 1;
    if (ctx->overlap && n_frames > ctx->samples_cap[buf]) {
        // first batch of this size: grow the sample buffers of all the slots this kind of launch rotates over now (one drain),
        // not one buffer per later launch
        if (int rc = sync_all(ctx)) return rc;
        for (int i = 0; i < slots * rvpt_hip_ctx::kBufsPerSlot; ++i) {
            if (ctx->samples_cap[i] >= n_frames) continue;
            HIP_TRY(ctx, hipFree(ctx->d_samples[i]));
            ctx->d_samples[i] = nullptr;
            ctx->samples_cap[i] = 0;
            const size_t bytes = static_cast<size_t>(n_frames) * ctx->slot_quads * sizeof(rv::SampleRGB);
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_samples[i]), bytes));
            // lanes of partial edge tiles outside the image never store; blend_accumulate folds the padding into the
            // accumulator padding, which is part of the tile buffer handed to gathers: keep it defined (as create does).
            // ON THE SLOT'S OWN STREAM: the streams are non-blocking, a null-stream hipMemset is not ordered against them
            // and would race with the frame kernel launched next (it did: 1 case in 1 500 of tools/fuzz_parity.py)
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_samples[i], 0, bytes, ctx->trace_stream[i / rvpt_hip_ctx::kBufsPerSlot]));
            ctx->samples_cap[i] = n_frames;
        }
    }
    rv::FrameParams p{};
    fill_frame_params(ctx, slot, p, buf);
    p.n_work = n_frames * ctx->n_work;
    Launch launch{};
    launch.slots = slots;
    // is anything of this context still running?  (a finished or never used stream answers hipSuccess)  Only the HBM-resident BVH launches of >= 16 frames act on the
    // answer (choose_launch); a one-frame brute-force launch does not pay six stream queries for it
    const bool lone_matters = (ctx->flags & RVPT_HIP_TRAVERSAL_MASK) != RVPT_HIP_TRAVERSAL_BRUTE && ctx->n_nodes > 0 && n_frames >= 16u;
    if (ctx->overlap && lone_matters) {
        launch.lone = true;
        for (int i = 0; i < ctx->n_slots && launch.lone; ++i)
            if (ctx->slot_used[i] && hipStreamQuery(ctx->trace_stream[i]) != hipSuccess) launch.lone = false;
        (void)hipGetLastError();  // hipErrorNotReady is an answer, not an error
    }
    if (int rc = choose_launch(ctx, p, launch)) return rc;
    plan_work(ctx, launch.regen, p, launch.variant == 6u ? 4u : 1u);
    if (launch.variant == 6u) plan_interleave(ctx, p);
    bool lean = false;  // packet kernel: the instances for launches with all three culls (below)
    if (launch.variant == 6u) {  // the camera records and the rectangles for this camera: made on the slot's stream, in front of the frame kernel, when the slot's buffers hold another camera's
        if (ctx->n_tris > ctx->rects_cap[slot]) {
            HIP_TRY(ctx, hipStreamSynchronize(tstream));
            if (ctx->d_rects[slot]) HIP_TRY(ctx, hipFree(ctx->d_rects[slot]));
            if (ctx->d_cam_records[slot]) HIP_TRY(ctx, hipFree(ctx->d_cam_records[slot]));
            ctx->d_rects[slot] = nullptr;
            ctx->d_cam_records[slot] = nullptr;
            ctx->rects_cap[slot] = 0;
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_rects[slot]), ctx->n_tris * sizeof(uint2)));
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_cam_records[slot]), ctx->n_tris * sizeof(float4)));
            ctx->rects_cap[slot] = ctx->n_tris;
            ctx->rects_valid[slot] = false;
        }
        rvpt_hip_ctx::RectKey key{};
        key.camera = ctx->camera;
        key.scene_gen = ctx->scene_gen;
        if (!ctx->rects_valid[slot] || std::memcmp(&key, &ctx->rects_key[slot], sizeof(key)) != 0) {
            const uint32_t n = static_cast<uint32_t>(ctx->n_tris);
            hipLaunchKernelGGL(rv::camera_rects, dim3((n + 63) / 64), dim3(64), 0, tstream, p, ctx->d_rects[slot], ctx->d_cam_records[slot]);
            HIP_TRY(ctx, hipGetLastError());
            ctx->rects_key[slot] = key;
            ctx->rects_valid[slot] = true;
        }
        if (launch.cull) p.rects = ctx->d_rects[slot];
        p.cam_records = ctx->d_cam_records[slot];
        // all three culls on and every chunk of the plan whole blocks (the default for a scene with a table): the instances without the uncull'd walks — fewer
        // registers to keep alive (SGPR spills 54 -> 19 for one sample per pixel; every spill is a v_readlane / v_writelane the VALU issues), same LDS, same occupancy
        const bool whole_blocks = p.first_units % 4u == 0u && p.claim_units % 4u == 0u && p.dyn_base % 4u == 0u && p.shard_len % 4u == 0u;
        lean = p.rects != nullptr && p.vis != nullptr && p.leaf_boxes != nullptr && p.sample_out != nullptr && whole_blocks && ctx->packets_lean_instance;
        if (lean)  // ... and with or without the interleaved claim order of short launches (its scalars cost the batched launches' kernel eleven spills more)
            launch.kernel = p.perm_groups != 0u ? ((p.aa == 1) ? rv::trace_brute_packets_aa1_culls_order : rv::trace_brute_packets_culls_order)
                                                : ((p.aa == 1) ? rv::trace_brute_packets_aa1_culls : rv::trace_brute_packets_culls);
    }
    if ((launch.variant == 2 || launch.variant == 10 || launch.variant == 11 || launch.variant == 12 || launch.variant == 13) && p.stack_levels > p.stack_lds_levels) {  // global part of the traversal stack, one column per thread and level
        const size_t words = static_cast<size_t>(2) * (p.stack_levels - p.stack_lds_levels) * launch.grid * rv::kBlock;
        if (words > ctx->stack_overflow_cap[slot]) {
            HIP_TRY(ctx, hipStreamSynchronize(tstream));
            if (ctx->d_stack_overflow[slot]) HIP_TRY(ctx, hipFree(ctx->d_stack_overflow[slot]));
            ctx->d_stack_overflow[slot] = nullptr;
            ctx->stack_overflow_cap[slot] = 0;
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_stack_overflow[slot]), words * sizeof(uint32_t)));
            ctx->stack_overflow_cap[slot] = words;
        }
        p.stack_overflow = ctx->d_stack_overflow[slot];
    }
    if (!ctx->timeline_path.empty()) {
        const size_t words = static_cast<size_t>(p.n_waves) * 16;  // 8 per wave + 8 more for the instrumented BVH build's load counts
        if (words > ctx->timeline_words) {
            if (ctx->d_timeline) HIP_TRY(ctx, hipFree(ctx->d_timeline));
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->d_timeline), words * 8));
            ctx->timeline_words = words;
        }
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_timeline, 0, words * 8, tstream));
        p.timeline = ctx->d_timeline;
    }
    ctx->last_grid = launch.grid;
    ctx->last_lds = static_cast<uint32_t>(launch.lds);
    ctx->last_variant = launch.variant;
    ctx->last_cull = (p.rects != nullptr ? 1u : 0u) | (p.vis != nullptr ? 2u : 0u) | ((launch.variant == 6u && p.first_units % 4u == 0u) ? 4u : 0u) | (p.leaf_boxes != nullptr ? 16u : 0u) | (p.perm_groups != 0u ? 32u : 0u) | (lean ? 64u : 0u);

    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    if (ctx->timing) {
        if (ctx->pending.size() >= 4096) {
            int rc = drain_timing(ctx);
            if (rc) return rc;
        }
        if (!ctx->spare.empty()) {
            ev0 = ctx->spare.back().first;
            ev1 = ctx->spare.back().second;
            ctx->spare.pop_back();
        } else {
            HIP_TRY(ctx, hipEventCreate(&ev0));
            HIP_TRY(ctx, hipEventCreate(&ev1));
        }
    }
    // this sample buffer is free once the blend of the launch that last wrote it — two launches back on this stream — has consumed it
    if (ctx->overlap && ctx->buf_used[buf]) HIP_TRY(ctx, hipStreamWaitEvent(tstream, ctx->blend_done[buf], 0));
    ctx->buf_used[buf] = true;
    ctx->slot_used[slot] = true;
    if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ev0, tstream));
    hipLaunchKernelGGL(launch.kernel, dim3(launch.grid), dim3(rv::kBlock), launch.lds, tstream, p);
    HIP_TRY(ctx, hipGetLastError());
    if (ctx->timing) {
        HIP_TRY(ctx, hipEventRecord(ev1, tstream));
        ctx->pending.emplace_back(ev0, ev1);
    }
    if (ctx->overlap) {  // the temporal blend of this frame, after its samples and after every earlier blend
        HIP_TRY(ctx, hipEventRecord(ctx->trace_done[slot], tstream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->trace_done[slot], 0));
        hipLaunchKernelGGL(rv::blend_accumulate, dim3((ctx->n_work + 255) / 256), dim3(256), 0, ctx->stream, ctx->d_samples[buf],
                           ctx->d_accum, ctx->n_work, n_frames, p.frame, p.quantize);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipEventRecord(ctx->blend_done[buf], ctx->stream));
    }
    ctx->seq += 1;
    return RVPT_HIP_OK;
}

// The work / exit counters of a slot are reset by the last wave of each launch.  If a launch failed or the device reported
// an error, that may not have happened and every later frame on the slot would silently skip claimed work: zero them
// (best effort — the context may be beyond repair, the caller gets the original error either way).
static void reset_counters_after_error(rvpt_hip_ctx *ctx)
{
    if (!ctx || !ctx->d_counter) return;
    (void)hipDeviceSynchronize();
    (void)hipGetLastError();
    (void)hipMemsetAsync(ctx->d_counter, 0, ctx->n_slots * rv::kCounterWords * sizeof(unsigned long long), ctx->stream);
    (void)hipStreamSynchronize(ctx->stream);  // the context's streams are non-blocking: nothing may start before the counters are zero
}

static int dispatch_checked(rvpt_hip_ctx *ctx, uint32_t n_frames)
{
    if (!ctx) return fail(nullptr, RVPT_HIP_ERR_INVALID, "ctx is NULL");
    if (!ctx->have_scene) return fail(ctx, RVPT_HIP_ERR_INVALID, "dispatch before upload_scene");
    if (!ctx->have_frame) return fail(ctx, RVPT_HIP_ERR_INVALID, "dispatch before set_frame");
    if (n_frames == 0 || n_frames > RVPT_HIP_MAX_FRAMES_PER_DISPATCH)
        return fail(ctx, RVPT_HIP_ERR_INVALID, "n_frames = %u (1..%u)", n_frames, RVPT_HIP_MAX_FRAMES_PER_DISPATCH);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->n_work == 0) return RVPT_HIP_OK;  // this rank owns no tile
    // one launch covers as many frames as fit 2^31 work items; without frames in flight there are no sample buffers
    // to batch into and the frames go one by one (same result, dispatch order)
    const uint64_t max_items = 0x7FFFFFFFull;
    const uint32_t per_launch = ctx->overlap ? std::max<uint32_t>(1, std::min<uint32_t>(n_frames, static_cast<uint32_t>(max_items / ctx->n_work))) : 1u;
    const uint32_t base = ctx->settings.current_frame;
    int rc = RVPT_HIP_OK;
    for (uint32_t done = 0; done < n_frames && rc == RVPT_HIP_OK; done += per_launch) {
        ctx->settings.current_frame = base + done;
        rc = dispatch_launch(ctx, std::min(per_launch, n_frames - done));
    }
    ctx->settings.current_frame = base;
    if (rc == RVPT_HIP_ERR_HIP) reset_counters_after_error(ctx);
    return rc;
}

}  // namespace

int rvpt_hip_dispatch(rvpt_hip_ctx *ctx) { return dispatch_checked(ctx, 1); }

int rvpt_hip_dispatch_frames(rvpt_hip_ctx *ctx, uint32_t n_frames) { return dispatch_checked(ctx, n_frames); }

int rvpt_hip_wait(rvpt_hip_ctx *ctx)
{
    if (!ctx) return fail(nullptr, RVPT_HIP_ERR_INVALID, "ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int rc = sync_all(ctx);
    if (rc == RVPT_HIP_ERR_HIP) reset_counters_after_error(ctx);
    if (rc == RVPT_HIP_OK && ctx->debug_checks && ctx->d_counter) {
        // the word after each slot's exited-wave counter: set by a wave that pushed past the top of the traversal stack the host sized (rvpt_device.h:
        // report_stack_overflow) — the clamp kept the kernel inside its memory, the image is wrong; never silently
        bool overflow = false;
        for (int i = 0; i < ctx->n_slots; ++i) {
            unsigned long long word = 0;
            unsigned long long *at = ctx->d_counter + static_cast<size_t>(i) * rv::kCounterWords + rv::kShardStride * rv::kClaimShards + 1;
            HIP_TRY(ctx, hipMemcpy(&word, at, sizeof(word), hipMemcpyDeviceToHost));
            if (word != 0) {
                overflow = true;
                HIP_TRY(ctx, hipMemset(at, 0, sizeof(word)));
            }
        }
        if (overflow) return fail(ctx, RVPT_HIP_ERR_INVALID, "BVH traversal stack overflow: a lane pushed past the %u levels the host sized for this tree — the frame is wrong", ctx->wide_stack_levels ? ctx->wide_stack_levels : ctx->bvh_height);
    }
    return rc;
}

int rvpt_hip_query(rvpt_hip_ctx *ctx)
{
    if (!ctx) return fail(nullptr, RVPT_HIP_ERR_INVALID, "ctx is NULL");
    for (int i = 0; i <= ctx->n_slots; ++i) {
        hipStream_t st = (i < ctx->n_slots) ? ctx->trace_stream[i] : ctx->stream;
        hipError_t e = hipStreamQuery(st);
        if (e == hipErrorNotReady) return 1;
        if (e != hipSuccess) return fail(ctx, RVPT_HIP_ERR_HIP, "hipStreamQuery -> %s", hipGetErrorString(e));
    }
    return 0;
}

int rvpt_hip_wait_for(rvpt_hip_ctx *ctx, uint64_t timeout_ns)
{
    if (!ctx) return fail(nullptr, RVPT_HIP_ERR_INVALID, "ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const auto deadline = std::chrono::steady_clock::now() + std::chrono::nanoseconds(timeout_ns);
    for (;;) {
        const int q = rvpt_hip_query(ctx);
        if (q < 0) reset_counters_after_error(ctx);
        if (q <= 0) return q == 0 ? sync_all(ctx) : q;  // done (collect timing events) or error
        if (std::chrono::steady_clock::now() >= deadline) return 1;
        std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}


namespace {

// How long a collective (or the communicator's bootstrap) may take before it is reported as failed: RVPT_HIP_COMM_TIMEOUT_S, default
// 120 s.  A peer that never enters a collective would otherwise leave this rank waiting forever inside a stream.
static double comm_timeout_s()
{
    const char *e = getenv("RVPT_HIP_COMM_TIMEOUT_S");
    const double v = e ? atof(e) : 0.0;
    return v > 0.0 ? v : 120.0;
}

// Wait for `stream` — which carries a collective — at most comm_timeout_s().  On a timeout the communicator is aborted (a
// collective that did not complete leaves it unusable) and every context of the group loses it: later collectives report "no
// communicator" instead of waiting again.
static int ensure_comm_stream(rvpt_hip_ctx *ctx, rvpt_hip_ctx *m)
{
    if (m->comm_stream && !m->comm_stream_lost) return RVPT_HIP_OK;
    m->comm_stream = nullptr;  // (a lost stream is leaked on purpose: destroying it would wait for the collective that never completes)
    m->comm_stream_lost = false;
    HIP_TRY(ctx, hipStreamCreateWithFlags(&m->comm_stream, hipStreamNonBlocking));
    return RVPT_HIP_OK;
}

static int sync_collective(rvpt_hip_ctx *ctx, rvpt_hip_ctx *member, const char *what)
{
    const auto t0 = std::chrono::steady_clock::now();
    const auto deadline = t0 + std::chrono::duration<double>(comm_timeout_s());
    for (;;) {
        const hipError_t e = hipStreamQuery(member->comm_stream);
        if (e == hipSuccess) return RVPT_HIP_OK;
        if (e != hipErrorNotReady) return fail(ctx, RVPT_HIP_ERR_HIP, "%s: hipStreamQuery -> %s", what, hipGetErrorString(e));
        const auto now = std::chrono::steady_clock::now();
        if (now >= deadline) break;
        if (now - t0 > std::chrono::milliseconds(2)) std::this_thread::sleep_for(std::chrono::microseconds(50));  // spin first: a warm gather takes ~0.1 ms
    }
    const std::vector<rvpt_hip_ctx *> group = ctx->local_group.empty() ? std::vector<rvpt_hip_ctx *>{ctx} : ctx->local_group;
    bool aborted = true;
    for (rvpt_hip_ctx *m : group) {
        if (m->comm && !(rccl().CommAbort && rccl().CommAbort(m->comm) == ncclSuccess)) aborted = false;
        m->comm = nullptr;
        m->comm_stream_lost = true;  // whatever the abort achieved, nothing of this context ever waits on that stream again
    }
    for (rvpt_hip_ctx *m : group) m->local_group.clear();
    return fail(ctx, RVPT_HIP_ERR_COMM, "%s timed out after %.0f s (rank %u of %u: did every rank enter the collective?); the communicator was %s and its stream "
                "abandoned — rendering, read-back of this rank's tiles and wait are unaffected (they never share a stream with a collective)",
                what, comm_timeout_s(), ctx->tile_rank, ctx->tile_world, aborted ? "aborted" : "dropped (ncclCommAbort unavailable or failed)");
}

// Gather of per-tile radiance to rank 0: every rank sends its tile-linear accumulator (one slot of slot_quads
// pixels, the payload the renderer already keeps resident), rank 0 receives tile_world slots — grouped ncclSend / ncclRecv,
// so over xGMI each peer uses its own direct link to the root — and un-tiles them into a row-major frame.
// `frame_dev` (rank 0): width*height float4 on ctx's device, or NULL to receive without un-tiling (rank 0 found its own arguments
// invalid but still takes part: a rank that returned early would leave its peers waiting).  Single-process groups are driven from
// rank 0's context.  Everything that can fail LOCALLY (arguments, allocations, the frames in flight) is dealt with before the group
// is opened; a failure after that point aborts the communicator (sync_collective).
static int gather_to_root(rvpt_hip_ctx *ctx, float4 *frame_dev)
{
    const Rccl &n = rccl();
    if (!ctx->comm) return fail(ctx, RVPT_HIP_ERR_COMM, "context has no communicator (rvpt_hip_comm_init / rvpt_hip_comm_init_all)");
    const bool single_process = !ctx->local_group.empty();
    if (single_process && ctx->tile_rank != 0)
        return fail(ctx, RVPT_HIP_ERR_INVALID, "in a single-process group the collective is driven through rank 0's context");
    const size_t floats = ctx->slot_quads * 4;
    std::vector<rvpt_hip_ctx *> members = single_process ? ctx->local_group : std::vector<rvpt_hip_ctx *>{ctx};
    // local work first.  A local failure here is reported AFTER the exchange: this rank still posts its send / receives (of
    // whatever the accumulator holds), because its peers are already on their way into the group call.
    int local_rc = RVPT_HIP_OK;
    std::string local_err;
    for (rvpt_hip_ctx *m : members) {  // everything rendered and blended before the accumulator leaves
        if (hipSetDevice(m->device) != hipSuccess || sync_all(m) != RVPT_HIP_OK) {
            if (local_rc == RVPT_HIP_OK) local_rc = RVPT_HIP_ERR_HIP, local_err = m->err.empty() ? "hipSetDevice failed" : m->err;
        }
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->tile_rank == 0 && !ctx->d_gather) {
        if (hipMalloc(reinterpret_cast<void **>(&ctx->d_gather), static_cast<size_t>(ctx->tile_world) * floats * sizeof(float)) != hipSuccess) {
            // without a receive buffer this rank cannot take part; the peers' sends time out on their side (sync_collective)
            ctx->d_gather = nullptr;
            return fail(ctx, RVPT_HIP_ERR_HIP, "gather buffer of %zu bytes could not be allocated", static_cast<size_t>(ctx->tile_world) * floats * sizeof(float));
        }
    }
    for (rvpt_hip_ctx *m : members) {
        HIP_TRY(ctx, hipSetDevice(m->device));
        if (int rc = ensure_comm_stream(ctx, m)) return rc;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    RCCL_TRY(ctx, n.GroupStart());
    ncclResult_t first_error = ncclSuccess;  // a group that was opened is always closed, whatever a call inside it returned
    auto in_group = [&](ncclResult_t r) {
        if (r != ncclSuccess && first_error == ncclSuccess) first_error = r;
    };
    for (rvpt_hip_ctx *m : members) {
        if (m->tile_rank == 0)
            for (uint32_t r = 0; r < m->tile_world && first_error == ncclSuccess; ++r)
                in_group(n.Recv(reinterpret_cast<float *>(m->d_gather) + static_cast<size_t>(r) * floats, floats, ncclFloat, static_cast<int>(r), m->comm, m->comm_stream));
        if (first_error == ncclSuccess) in_group(n.Send(m->d_accum, floats, ncclFloat, 0, m->comm, m->comm_stream));
    }
    in_group(n.GroupEnd());
    if (first_error != ncclSuccess) return fail(ctx, RVPT_HIP_ERR_COMM, "gather of per-tile radiance -> %s", n.GetErrorString(first_error));
    if (ctx->tile_rank == 0 && frame_dev != nullptr) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        const dim3 blk(64, 4), grd((ctx->width + 63) / 64, (ctx->height + 3) / 4);
        hipLaunchKernelGGL(rv::untile_rgba32f, grd, blk, 0, ctx->comm_stream, ctx->d_gather, ctx->slot_quads, ctx->tile_world, ctx->width, ctx->height,
                           ctx->tiles_x, frame_dev);
        HIP_TRY(ctx, hipGetLastError());
    }
    for (rvpt_hip_ctx *m : members) {
        HIP_TRY(ctx, hipSetDevice(m->device));
        if (int rc = sync_collective(ctx, m, "gather of per-tile radiance")) return rc;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (local_rc != RVPT_HIP_OK) return fail(ctx, local_rc, "%s (the gather itself completed)", local_err.c_str());
    return RVPT_HIP_OK;
}

void drop_comm(rvpt_hip_ctx *ctx)
{
    // A single-process group dissolves as a whole with its first member to go: every member's communicator is destroyed
    // here, so that a later collective on a surviving context reports "no communicator" instead of waiting for a peer
    // that no longer exists.
    const std::vector<rvpt_hip_ctx *> group = ctx->local_group.empty() ? std::vector<rvpt_hip_ctx *>{ctx} : ctx->local_group;
    for (rvpt_hip_ctx *m : group) {
        if (m->comm && rccl().ok) {
            (void)hipSetDevice(m->device);
            (void)rccl().CommDestroy(m->comm);
        }
        m->comm = nullptr;
        m->local_group.clear();
    }
    (void)hipSetDevice(ctx->device);
}

}  // namespace

int rvpt_hip_comm_unique_id(void *id_out, size_t id_bytes)
{
    if (!id_out || id_bytes < RVPT_HIP_COMM_ID_BYTES) return fail(nullptr, RVPT_HIP_ERR_INVALID, "id buffer must hold %d bytes", RVPT_HIP_COMM_ID_BYTES);
    static_assert(sizeof(ncclUniqueId) == RVPT_HIP_COMM_ID_BYTES, "RVPT_HIP_COMM_ID_BYTES");
    if (!rccl().ok) return fail(nullptr, RVPT_HIP_ERR_COMM, "%s", rccl().why.c_str());
    ncclUniqueId id;
    RCCL_TRY(nullptr, rccl().GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof id);
    return RVPT_HIP_OK;
}

int rvpt_hip_comm_init(rvpt_hip_ctx *ctx, const void *unique_id, size_t id_bytes)
{
    if (!ctx) return fail(nullptr, RVPT_HIP_ERR_INVALID, "ctx is NULL");
    if (!unique_id || id_bytes < RVPT_HIP_COMM_ID_BYTES) return fail(ctx, RVPT_HIP_ERR_INVALID, "unique id must be %d bytes", RVPT_HIP_COMM_ID_BYTES);
    if (ctx->comm) return fail(ctx, RVPT_HIP_ERR_INVALID, "context already has a communicator");
    if (!rccl().ok) return fail(ctx, RVPT_HIP_ERR_COMM, "%s", rccl().why.c_str());
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    // ncclCommInitRank blocks until every rank of the group has joined; a rank that never comes (it failed before this call, it was
    // handed another id) must not hang the others: the bootstrap runs on a helper thread and is given comm_timeout_s()
    struct Boot {
        std::mutex m;
        std::condition_variable cv;
        bool done = false, abandoned = false;  // abandoned: the caller gave up waiting; a communicator that arrives late is the helper's to destroy
        ncclResult_t result = ncclSuccess;
        ncclComm_t comm = nullptr;
    };
    auto boot = std::make_shared<Boot>();
    const int device = ctx->device, world = static_cast<int>(ctx->tile_world), rank = static_cast<int>(ctx->tile_rank);
    std::thread([boot, id, device, world, rank] {
        ncclComm_t c = nullptr;
        ncclResult_t r = (hipSetDevice(device) == hipSuccess) ? rccl().CommInitRank(&c, world, id, rank) : static_cast<ncclResult_t>(1);
        bool late = false;
        {
            std::lock_guard<std::mutex> lock(boot->m);
            boot->result = r;
            boot->comm = c;
            boot->done = true;
            late = boot->abandoned;
            boot->cv.notify_all();
        }
        // the bootstrap completed after rvpt_hip_comm_init had timed out: nobody will ever use this communicator, and its peers believe this rank
        // joined — abort it (their next collective then fails at once instead of waiting out the full timeout) or at least destroy it
        if (late && r == ncclSuccess && c != nullptr) {
            if (!(rccl().CommAbort && rccl().CommAbort(c) == ncclSuccess)) (void)rccl().CommDestroy(c);
        }
    }).detach();
    {
        std::unique_lock<std::mutex> lock(boot->m);
        if (!boot->cv.wait_for(lock, std::chrono::duration<double>(comm_timeout_s()), [&] { return boot->done; })) {
            boot->abandoned = true;
            return fail(ctx, RVPT_HIP_ERR_COMM, "ncclCommInitRank (rank %d of %d) did not complete within %.0f s: not every rank joined with this id", rank, world, comm_timeout_s());
        }
        if (boot->result != ncclSuccess) return fail(ctx, RVPT_HIP_ERR_COMM, "ncclCommInitRank (rank %d of %d) -> %s", rank, world, rccl().GetErrorString(boot->result));
        ctx->comm = boot->comm;
    }
    return RVPT_HIP_OK;
}

int rvpt_hip_comm_init_all(rvpt_hip_ctx *const *ctxs, int n)
{
    if (!ctxs || n < 1) return fail(nullptr, RVPT_HIP_ERR_INVALID, "no contexts");
    std::vector<int> devices(static_cast<size_t>(n));
    for (int i = 0; i < n; ++i) {
        rvpt_hip_ctx *c = ctxs[i];
        if (!c) return fail(nullptr, RVPT_HIP_ERR_INVALID, "ctxs[%d] is NULL", i);
        if (c->tile_world != static_cast<uint32_t>(n) || c->tile_rank != static_cast<uint32_t>(i))
            return fail(c, RVPT_HIP_ERR_INVALID, "ctxs[%d] is tile %u of %u: the group must list ranks 0..%d of %d in order", i, c->tile_rank, c->tile_world, n - 1, n);
        if (c->comm) return fail(c, RVPT_HIP_ERR_INVALID, "ctxs[%d] already has a communicator", i);
        if (c->width != ctxs[0]->width || c->height != ctxs[0]->height) return fail(c, RVPT_HIP_ERR_INVALID, "ctxs[%d]: image size differs", i);
        devices[static_cast<size_t>(i)] = c->device;
    }
    if (!rccl().ok) return fail(ctxs[0], RVPT_HIP_ERR_COMM, "%s", rccl().why.c_str());
    std::vector<ncclComm_t> comms(static_cast<size_t>(n), nullptr);
    RCCL_TRY(ctxs[0], rccl().CommInitAll(comms.data(), n, devices.data()));
    std::vector<rvpt_hip_ctx *> group(ctxs, ctxs + n);
    for (int i = 0; i < n; ++i) {
        ctxs[i]->comm = comms[static_cast<size_t>(i)];
        ctxs[i]->local_group = group;
    }
    return RVPT_HIP_OK;
}

int rvpt_hip_comm_info(rvpt_hip_ctx *ctx, int *n_ranks, int *rank, int *rccl_version)
{
    if (!ctx) return fail(nullptr, RVPT_HIP_ERR_INVALID, "ctx is NULL");
    if (n_ranks) *n_ranks = 0;
    if (rank) *rank = -1;
    if (rccl_version) *rccl_version = 0;
    if (!ctx->comm) return fail(ctx, RVPT_HIP_ERR_COMM, "context has no communicator (rvpt_hip_comm_init / rvpt_hip_comm_init_all)");
    const Rccl &n = rccl();
    // asked of RCCL itself, not echoed from the context: "did RCCL see N ranks?" is what a scaling record must be able to answer
    if (n_ranks && n.CommCount) RCCL_TRY(ctx, n.CommCount(ctx->comm, n_ranks));
    if (rank && n.CommUserRank) RCCL_TRY(ctx, n.CommUserRank(ctx->comm, rank));
    if (rccl_version && n.GetVersion) RCCL_TRY(ctx, n.GetVersion(rccl_version));
    return RVPT_HIP_OK;
}

int rvpt_hip_comm_destroy(rvpt_hip_ctx *ctx)
{
    if (!ctx) return fail(nullptr, RVPT_HIP_ERR_INVALID, "ctx is NULL");
    drop_comm(ctx);  // a single-process group dissolves as a whole
    return RVPT_HIP_OK;
}

int rvpt_hip_comm_barrier(rvpt_hip_ctx *ctx)
{
    if (!ctx) return fail(nullptr, RVPT_HIP_ERR_INVALID, "ctx is NULL");
    if (!ctx->comm) return fail(ctx, RVPT_HIP_ERR_COMM, "context has no communicator (rvpt_hip_comm_init / rvpt_hip_comm_init_all)");
    const bool single_process = !ctx->local_group.empty();
    if (single_process && ctx->tile_rank != 0)
        return fail(ctx, RVPT_HIP_ERR_INVALID, "in a single-process group the collective is driven through rank 0's context");
    const Rccl &n = rccl();
    std::vector<rvpt_hip_ctx *> members = single_process ? ctx->local_group : std::vector<rvpt_hip_ctx *>{ctx};
    for (rvpt_hip_ctx *m : members) {  // this rank's own work first, and the one float the all-reduce carries
        HIP_TRY(ctx, hipSetDevice(m->device));
        if (int rc = sync_all(m)) return rc;
        if (int rc = ensure_comm_stream(ctx, m)) return rc;
        if (!m->d_barrier) {
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&m->d_barrier), 256));
            HIP_TRY(ctx, hipMemsetAsync(m->d_barrier, 0, 256, m->comm_stream));
        }
    }
    RCCL_TRY(ctx, n.GroupStart());
    ncclResult_t first_error = ncclSuccess;
    for (rvpt_hip_ctx *m : members) {
        const ncclResult_t r = n.AllReduce(m->d_barrier, m->d_barrier, 1, ncclFloat, ncclSum, m->comm, m->comm_stream);
        if (r != ncclSuccess && first_error == ncclSuccess) first_error = r;
    }
    {
        const ncclResult_t r = n.GroupEnd();
        if (r != ncclSuccess && first_error == ncclSuccess) first_error = r;
    }
    if (first_error != ncclSuccess) return fail(ctx, RVPT_HIP_ERR_COMM, "barrier -> %s", n.GetErrorString(first_error));
    for (rvpt_hip_ctx *m : members) {
        HIP_TRY(ctx, hipSetDevice(m->device));
        if (int rc = sync_collective(ctx, m, "barrier")) return rc;
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return RVPT_HIP_OK;
}

int rvpt_hip_gather(rvpt_hip_ctx *ctx, void *dst_dev_rgba32f)
{
    if (!ctx) return fail(nullptr, RVPT_HIP_ERR_INVALID, "ctx is NULL");
    if (ctx->tile_rank == 0 && !dst_dev_rgba32f && ctx->comm) {  // rank 0's own mistake must not leave the peers waiting: take part, then report
        (void)gather_to_root(ctx, nullptr);
        return fail(ctx, RVPT_HIP_ERR_INVALID, "rank 0 needs a destination");
    }
    return gather_to_root(ctx, static_cast<float4 *>(dst_dev_rgba32f));
}

int rvpt_hip_read(rvpt_hip_ctx *ctx, int format, void *dst, size_t dst_bytes)
{
    if (!ctx) return fail(nullptr, RVPT_HIP_ERR_INVALID, "ctx is NULL");
    const bool collective = ctx->comm != nullptr;  // partitioned image with a communicator: gather to rank 0, which gets the frame
    if (collective && ctx->tile_rank != 0 && ctx->local_group.empty()) return gather_to_root(ctx, nullptr);  // peers only send
    const size_t px = static_cast<size_t>(ctx->width) * ctx->height;
    const size_t need = px * (format == RVPT_HIP_FORMAT_RGBA32F ? 16 : 4);
    // rank 0's arguments.  In a collective read its peers are already sending: a bad argument here is reported AFTER this rank
    // has taken part in the exchange (receiving without un-tiling), never by leaving them in the group call.
    int arg_rc = RVPT_HIP_OK;
    if (!dst)
        arg_rc = fail(ctx, RVPT_HIP_ERR_INVALID, "dst is NULL");
    else if (format != RVPT_HIP_FORMAT_RGBA32F && format != RVPT_HIP_FORMAT_RGBA8_UNORM)
        arg_rc = fail(ctx, RVPT_HIP_ERR_INVALID, "unknown format %d", format);
    else if (dst_bytes < need)
        arg_rc = fail(ctx, RVPT_HIP_ERR_SIZE, "dst holds %zu bytes, frame needs %zu", dst_bytes, need);
    if (arg_rc == RVPT_HIP_OK && hipSetDevice(ctx->device) != hipSuccess) arg_rc = fail(ctx, RVPT_HIP_ERR_HIP, "hipSetDevice failed");
    if (arg_rc == RVPT_HIP_OK) arg_rc = ensure_rowmajor(ctx);
    if (arg_rc != RVPT_HIP_OK) {
        if (collective) {
            const std::string keep = ctx->err;
            (void)gather_to_root(ctx, nullptr);
            ctx->err = keep;
        }
        return arg_rc;
    }
    int rc = RVPT_HIP_OK;
    if (collective) {
        if ((rc = gather_to_root(ctx, static_cast<float4 *>(ctx->d_rowmajor)))) return rc;
        const void *src = ctx->d_rowmajor;
        if (format == RVPT_HIP_FORMAT_RGBA8_UNORM) {
            if (!ctx->d_quant) HIP_TRY(ctx, hipMalloc(&ctx->d_quant, px * 4));
            hipLaunchKernelGGL(rv::quantize_rowmajor, dim3(static_cast<uint32_t>((px + 255) / 256)), dim3(256), 0, ctx->stream,
                               static_cast<const float4 *>(ctx->d_rowmajor), static_cast<uint32_t>(px), static_cast<uint32_t *>(ctx->d_quant));
            HIP_TRY(ctx, hipGetLastError());
            src = ctx->d_quant;
        }
        HIP_TRY(ctx, hipMemcpyAsync(dst, src, need, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return RVPT_HIP_OK;
    }
    const dim3 blk(64, 4), grd((ctx->width + 63) / 64, (ctx->height + 3) / 4);
    hipLaunchKernelGGL(rv::read_rowmajor, grd, blk, 0, ctx->stream, ctx->d_accum, ctx->width, ctx->height, ctx->tiles_x,
                       ctx->tile_rank, ctx->tile_world, format == RVPT_HIP_FORMAT_RGBA8_UNORM ? 1 : 0, ctx->d_rowmajor);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(dst, ctx->d_rowmajor, need, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RVPT_HIP_OK;
}

int rvpt_hip_tile_buffer(rvpt_hip_ctx *ctx, void **device_ptr, size_t *bytes, size_t *max_tile_bytes)
{
    if (!ctx) return fail(nullptr, RVPT_HIP_ERR_INVALID, "ctx is NULL");
    if (device_ptr) *device_ptr = ctx->d_accum;
    if (bytes) *bytes = static_cast<size_t>(ctx->n_work) * sizeof(float4);
    if (max_tile_bytes) *max_tile_bytes = static_cast<size_t>(owned_tiles(ctx->tiles_x * ctx->tiles_y, 0, ctx->tile_world)) * 256u * sizeof(float4);
    return RVPT_HIP_OK;
}

#if RVPT_HIP_LAB  // ---- include/rvpt_hip_lab.h: diagnostics of the arithmetic specification and of the exact culls; the laboratory build only
int rvpt_hip_selftest_div(int device_id, const float *a, const float *b, float *out, size_t n)
{
    if (!a || !b || !out) return fail(nullptr, RVPT_HIP_ERR_INVALID, "NULL array");
    if (n == 0) return RVPT_HIP_OK;
    if (n > (1u << 30)) return fail(nullptr, RVPT_HIP_ERR_INVALID, "n too large");
    HIP_TRY(nullptr, hipSetDevice(device_id));
    float *d = nullptr;
    HIP_TRY(nullptr, hipMalloc(reinterpret_cast<void **>(&d), 3 * n * sizeof(float)));
    hipError_t e = hipMemcpy(d, a, n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + n, b, n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rv::selftest_div_dots, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, nullptr, d, d + n, d + 2 * n, static_cast<uint32_t>(n));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, d + 2 * n, n * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(nullptr, RVPT_HIP_ERR_HIP, "selftest_div -> %s", hipGetErrorString(e));
    return RVPT_HIP_OK;
}

int rvpt_camera_rects(const float *prepared, size_t n_tris, const rvpt_camera_data *cam, uint32_t width, uint32_t height, uint32_t *rects_out)
{
    if ((n_tris && !prepared) || !cam || !rects_out) return RVPT_HIP_ERR_INVALID;
    // FrameParams' view of the camera block (fill_frame_params): columns 0..2, the origin, aspect, 1 / tan(vfov / 2)
    float c[12];
    for (int col = 0; col < 3; ++col)
        for (int r = 0; r < 3; ++r) c[3 * col + r] = cam->matrix[4 * col + r];
    for (int r = 0; r < 3; ++r) c[9 + r] = cam->matrix[12 + r];
    const float cam_w = 1.0f / rv::tan_det(0.5f * cam->params[1]);
    const rv::RectCamera rc = rv::rect_camera(c, cam->params[0], cam_w, width, height);
    for (size_t i = 0; i < n_tris; ++i) {
        const float *q = prepared + 16 * i;
        bool neg;
        const float a = rv::camera_numerator(rv::mk(q[0], q[1], q[2]), rv::mk(q[3], q[4], q[5]), rv::mk(c[9], c[10], c[11]), neg);
        rv::camera_rect(rc, q, q + 4, q + 8, a, rects_out[2 * i], rects_out[2 * i + 1]);
    }
    return RVPT_HIP_OK;
}

int rvpt_bounce_rows(const float *tris, const float *prepared, size_t n_tris, uint32_t *rows_out, double *scale_out)
{
    if ((n_tris && (!tris || !prepared)) || !rows_out) return RVPT_HIP_ERR_INVALID;
    if (n_tris > rv::kResidentMaxTris) return RVPT_HIP_ERR_SIZE;
    const double scale = rv::bounce_scene_scale(tris, n_tris);
    if (scale_out) *scale_out = scale;
    if (scale <= 0.0) return RVPT_HIP_OK;  // no table for this scene
    const uint32_t n = static_cast<uint32_t>(n_tris), words = (n + 31u) / 32u;
    for (uint32_t row = 0; row < 2u * n; ++row)
        for (uint32_t w = 0; w < words; ++w) rows_out[static_cast<size_t>(row) * words + w] = rv::bounce_row_word(prepared, n, row, w, rv::kBounceMarginScales * scale);
    return RVPT_HIP_OK;
}

int rvpt_claim_order(uint32_t n_work_frame, uint32_t group_blocks, uint32_t *order_out, uint32_t params_out[3])
{
    uint32_t groups = 0, stride = 1, shift = 0;
    const bool on = plan_claim_order(n_work_frame, group_blocks, groups, stride, shift);
    if (params_out) params_out[0] = on ? groups : 0u, params_out[1] = stride, params_out[2] = shift;
    if (order_out) {
        const rv::FastDiv d = rv::fast_div_make(on ? groups : 1u);
        for (uint32_t b = 0; b < n_work_frame / 64u; ++b) order_out[b] = on ? rv::claim_order_block(b, groups, stride, shift, d) : b;
    }
    return RVPT_HIP_OK;
}

int rvpt_bounce_leaf_boxes(const float *tris, size_t n_tris, float *boxes_out, uint32_t *leaf_tris_out, float *tri_boxes_out)
{
    if ((n_tris && !tris) || !boxes_out) return RVPT_HIP_ERR_INVALID;
    if (n_tris > rv::kResidentMaxTris) return RVPT_HIP_ERR_SIZE;
    if (leaf_tris_out) *leaf_tris_out = rv::kLeafTris;
    const double scale = rv::bounce_scene_scale(tris, n_tris);
    if (scale <= 0.0) return RVPT_HIP_OK;  // no table, no boxes for this scene
    rv::bounce_leaf_boxes(tris, n_tris, scale, boxes_out);
    if (tri_boxes_out) rv::bounce_group_boxes(tris, n_tris, scale, 1, tri_boxes_out);
    return RVPT_HIP_OK;
}

int rvpt_hip_selftest_camera_rects(rvpt_hip_ctx *ctx, uint32_t n_samples, uint64_t out[4], float *prepared_out, uint32_t *rects_out)
{
    if (!ctx) return fail(nullptr, RVPT_HIP_ERR_INVALID, "ctx is NULL");
    if (!out) return fail(ctx, RVPT_HIP_ERR_INVALID, "out is NULL");
    if (!ctx->have_scene || !ctx->have_frame) return fail(ctx, RVPT_HIP_ERR_INVALID, "selftest_camera_rects needs upload_scene and set_frame");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = sync_all(ctx)) return rc;
    std::memset(out, 0, 4 * sizeof(uint64_t));
    if (ctx->n_tris == 0) return RVPT_HIP_OK;
    rv::FrameParams p{};
    fill_frame_params(ctx, 0, p);
    uint2 *d_rects = nullptr;
    unsigned long long *d_out = nullptr;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&d_rects), ctx->n_tris * sizeof(uint2)));
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_out), 4 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemsetAsync(d_out, 0, 4 * sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess) {
        const uint32_t n = static_cast<uint32_t>(ctx->n_tris);
        hipLaunchKernelGGL(rv::camera_rects, dim3((n + 63) / 64), dim3(64), 0, ctx->stream, p, d_rects, static_cast<float4 *>(nullptr));
        hipLaunchKernelGGL(rv::selftest_camera_rects, dim3(static_cast<uint32_t>(ctx->num_cus) * 8u), dim3(256), 0, ctx->stream, p, d_rects, n_samples, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && prepared_out) e = hipMemcpyAsync(prepared_out, ctx->d_prep, ctx->n_tris * 64, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && rects_out) e = hipMemcpyAsync(rects_out, d_rects, ctx->n_tris * sizeof(uint2), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_rects);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return fail(ctx, RVPT_HIP_ERR_HIP, "selftest_camera_rects -> %s", hipGetErrorString(e));
    return RVPT_HIP_OK;
}

int rvpt_hip_selftest_fast_div(uint32_t divisor, const uint32_t *x, uint32_t *q, size_t n)
{
    if (divisor == 0 || (n && (!x || !q))) return RVPT_HIP_ERR_INVALID;
    const rv::FastDiv f = rv::fast_div_make(divisor);
    for (size_t i = 0; i < n; ++i) q[i] = rv::fast_div(x[i], f);
    return RVPT_HIP_OK;
}

int rvpt_hip_selftest_bounce_cull(rvpt_hip_ctx *ctx, uint32_t n_samples, uint64_t out[8])
{
    if (!ctx) return fail(nullptr, RVPT_HIP_ERR_INVALID, "ctx is NULL");
    if (!out) return fail(ctx, RVPT_HIP_ERR_INVALID, "out is NULL");
    if (!ctx->have_scene || !ctx->have_frame) return fail(ctx, RVPT_HIP_ERR_INVALID, "selftest_bounce_cull needs upload_scene and set_frame");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (int rc = sync_all(ctx)) return rc;
    std::memset(out, 0, 8 * sizeof(uint64_t));
    if (ctx->vis_words == 0 || ctx->n_tris == 0) return RVPT_HIP_OK;  // no table for this scene (a BVH context, too many triangles, absurd coordinates)
    rv::FrameParams p{};
    fill_frame_params(ctx, 0, p);
    p.vis = ctx->d_vis;
    p.vis_words = ctx->vis_words;
    p.vis_stride = (ctx->vis_words + 3u) & ~3u;
    p.leaf_boxes = ctx->d_leaf_boxes;
    unsigned long long *d_out = nullptr;
    unsigned long long h_out[5] = {};
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void **>(&d_out), 5 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d_out, 0, 5 * sizeof(unsigned long long), ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rv::selftest_bounce_cull, dim3(static_cast<uint32_t>(ctx->num_cus) * 8u), dim3(256), 0, ctx->stream, p, n_samples, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(h_out, d_out, 5 * sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream);
    std::vector<uint32_t> table(static_cast<size_t>(2) * ctx->n_tris * ((ctx->vis_words + 3u) & ~3u));  // (the padding words are zero)
    if (e == hipSuccess) e = hipMemcpyAsync(table.data(), ctx->d_vis, table.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(ctx, RVPT_HIP_ERR_HIP, "selftest_bounce_cull -> %s", hipGetErrorString(e));
    out[0] = h_out[0], out[1] = h_out[1], out[4] = h_out[2], out[5] = h_out[3], out[6] = h_out[4];
    for (uint32_t w : table) out[2] += static_cast<uint64_t>(__builtin_popcount(w));
    out[3] = static_cast<uint64_t>(2) * ctx->n_tris * ctx->n_tris;
    return RVPT_HIP_OK;
}

int rvpt_hip_selftest_pretest(int device_id, const float *a, const float *den, const float *closest, unsigned char *out, size_t n)
{
    if (!a || !den || !closest || !out) return fail(nullptr, RVPT_HIP_ERR_INVALID, "NULL array");
    if (n == 0) return RVPT_HIP_OK;
    if (n > (1u << 30)) return fail(nullptr, RVPT_HIP_ERR_INVALID, "n too large");
    HIP_TRY(nullptr, hipSetDevice(device_id));
    float *d = nullptr;
    HIP_TRY(nullptr, hipMalloc(reinterpret_cast<void **>(&d), 4 * n * sizeof(float)));
    unsigned char *d_out = reinterpret_cast<unsigned char *>(d + 3 * n);
    hipError_t e = hipMemcpy(d, a, n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + n, den, n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + 2 * n, closest, n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rv::selftest_camera_pretest, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, nullptr, d, d + n, d + 2 * n, d_out, static_cast<uint32_t>(n));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, d_out, n, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(nullptr, RVPT_HIP_ERR_HIP, "selftest_pretest -> %s", hipGetErrorString(e));
    return RVPT_HIP_OK;
}

int rvpt_hip_selftest_rcp(int device_id, uint64_t mismatches_per_exponent[256])
{
    if (!mismatches_per_exponent) return fail(nullptr, RVPT_HIP_ERR_INVALID, "NULL array");
    HIP_TRY(nullptr, hipSetDevice(device_id));
    unsigned long long *d = nullptr;
    HIP_TRY(nullptr, hipMalloc(reinterpret_cast<void **>(&d), 256 * sizeof(unsigned long long)));
    hipError_t e = hipMemset(d, 0, 256 * sizeof(unsigned long long));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rv::selftest_rcp_sweep, dim3((1u << 23) / 256, 254), dim3(256), 0, nullptr, d);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(mismatches_per_exponent, d, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(nullptr, RVPT_HIP_ERR_HIP, "selftest_rcp -> %s", hipGetErrorString(e));
    return RVPT_HIP_OK;
}

#endif  // RVPT_HIP_LAB

int rvpt_hip_untile(rvpt_hip_ctx *ctx, const void *gathered_dev, size_t slot_bytes, uint32_t n_ranks, void *dst_dev_rgba32f)
{
    if (!ctx) return fail(nullptr, RVPT_HIP_ERR_INVALID, "ctx is NULL");
    if (!gathered_dev || !dst_dev_rgba32f || n_ranks == 0 || slot_bytes % sizeof(float4)) return fail(ctx, RVPT_HIP_ERR_INVALID, "bad untile arguments");
    const size_t need = static_cast<size_t>(owned_tiles(ctx->tiles_x * ctx->tiles_y, 0, n_ranks)) * 256u * sizeof(float4);
    if (slot_bytes < need) return fail(ctx, RVPT_HIP_ERR_SIZE, "slot holds %zu bytes, rank 0 of %u needs %zu", slot_bytes, n_ranks, need);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const dim3 blk(64, 4), grd((ctx->width + 63) / 64, (ctx->height + 3) / 4);
    hipLaunchKernelGGL(rv::untile_rgba32f, grd, blk, 0, ctx->stream, static_cast<const float4 *>(gathered_dev), slot_bytes / sizeof(float4),
                       n_ranks, ctx->width, ctx->height, ctx->tiles_x, static_cast<float4 *>(dst_dev_rgba32f));
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RVPT_HIP_OK;
}

int rvpt_hip_write_accum(rvpt_hip_ctx *ctx, const void *src_rgba32f, size_t src_bytes)
{
    if (!ctx) return fail(nullptr, RVPT_HIP_ERR_INVALID, "ctx is NULL");
    const size_t need = static_cast<size_t>(ctx->width) * ctx->height * 16;
    if (!src_rgba32f) return fail(ctx, RVPT_HIP_ERR_INVALID, "src is NULL");
    if (src_bytes < need) return fail(ctx, RVPT_HIP_ERR_SIZE, "src holds %zu bytes, frame needs %zu", src_bytes, need);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int rc = ensure_rowmajor(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_rowmajor, src_rgba32f, need, hipMemcpyHostToDevice, ctx->stream));
    if (ctx->n_work) {
        hipLaunchKernelGGL(rv::tile_rgba32f, dim3((ctx->n_work + 255) / 256), dim3(256), 0, ctx->stream, static_cast<const float4 *>(ctx->d_rowmajor),
                           ctx->width, ctx->height, ctx->tiles_x, ctx->tile_rank, ctx->tile_world, ctx->n_work, ctx->d_accum);
        HIP_TRY(ctx, hipGetLastError());
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RVPT_HIP_OK;
}

int rvpt_hip_get_timing(rvpt_hip_ctx *ctx, float *kernel_ms_last, double *kernel_ms_sum, uint64_t *n_dispatches)
{
    if (!ctx) return fail(nullptr, RVPT_HIP_ERR_INVALID, "ctx is NULL");
    if (!ctx->timing) return fail(ctx, RVPT_HIP_ERR_INVALID, "context created without RVPT_HIP_TIMING");
    int rc = drain_timing(ctx);
    if (rc) return rc;
    if (kernel_ms_last) *kernel_ms_last = ctx->last_ms;
    if (kernel_ms_sum) *kernel_ms_sum = ctx->sum_ms;
    if (n_dispatches) *n_dispatches = ctx->n_timed;
    return RVPT_HIP_OK;
}

int rvpt_hip_reset_timing(rvpt_hip_ctx *ctx)
{
    if (!ctx) return fail(nullptr, RVPT_HIP_ERR_INVALID, "ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->timing) {
        int rc = drain_timing(ctx);
        if (rc) return rc;
    }
    ctx->last_ms = 0.f;
    ctx->sum_ms = 0.0;
    ctx->n_timed = 0;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_stats, 0, rv::kStatStripes * rv::kStatStride * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RVPT_HIP_OK;
}

int rvpt_hip_get_stats(rvpt_hip_ctx *ctx, uint64_t stats[2])
{
    if (!ctx) return fail(nullptr, RVPT_HIP_ERR_INVALID, "ctx is NULL");
    if (!stats) return fail(ctx, RVPT_HIP_ERR_INVALID, "stats is NULL");
    if (!(ctx->flags & RVPT_HIP_COUNT_SEGMENTS)) return fail(ctx, RVPT_HIP_ERR_INVALID, "context created without RVPT_HIP_COUNT_SEGMENTS");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    unsigned long long h[rv::kStatStripes * rv::kStatStride] = {};  // (64 pairs, one cache line each: rvpt_kernels.h)
    HIP_TRY(ctx, hipMemcpyAsync(h, ctx->d_stats, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    stats[0] = stats[1] = 0;
    for (uint32_t s = 0; s < rv::kStatStripes; ++s) {
        stats[0] += h[rv::kStatStride * s + 0];
        stats[1] += h[rv::kStatStride * s + 1];
    }
    return RVPT_HIP_OK;
}

int rvpt_hip_get_launch_info(rvpt_hip_ctx *ctx, uint32_t *grid_blocks, uint32_t *lds_bytes, uint32_t *kernel_variant,
                             uint32_t *frames_in_flight)
{
    if (!ctx) return fail(nullptr, RVPT_HIP_ERR_INVALID, "ctx is NULL");
    if (ctx->last_grid == 0) return fail(ctx, RVPT_HIP_ERR_INVALID, "no frame dispatched yet");
    if (grid_blocks) *grid_blocks = ctx->last_grid;
    if (lds_bytes) *lds_bytes = ctx->last_lds;
    if (kernel_variant) *kernel_variant = ctx->last_variant;
    if (frames_in_flight) *frames_in_flight = static_cast<uint32_t>(ctx->last_slots ? ctx->last_slots : slots_for(ctx, 1));
    return RVPT_HIP_OK;
}

int rvpt_hip_get_cull_info(rvpt_hip_ctx *ctx, uint32_t *flags)
{
    if (!ctx) return fail(nullptr, RVPT_HIP_ERR_INVALID, "ctx is NULL");
    if (!flags) return fail(ctx, RVPT_HIP_ERR_INVALID, "flags is NULL");
    if (ctx->last_grid == 0) return fail(ctx, RVPT_HIP_ERR_INVALID, "no frame dispatched yet");
    *flags = ctx->last_cull;
    return RVPT_HIP_OK;
}

const char *rvpt_hip_last_error(rvpt_hip_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

}  // extern "C"

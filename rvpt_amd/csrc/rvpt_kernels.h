// rvpt_kernels.h — kernel parameter block and launch constants shared by the kernels and the C-ABI
// launcher (rvpt_abi.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef RVPT_HIP_LAB
#define RVPT_HIP_LAB 0  // 1 (rvpt_amd/build.py: librvpt_hip_debug.so): + the selftest kernels and the opt-in walks that measured slower (trace_bvh8, trace_bvh4q)
#endif

namespace rv {

constexpr uint32_t kBlock = 256;            // 4 wavefronts per work-group
// Tile ownership (include/rvpt_hip.h: RVPT_HIP_TILE_SHIFT): the tile at (tx, ty) of the tile grid has SLOT s = ty * tiles_x + (tx + kTileShift * ty) % tiles_x
// — a row-major numbering in which every row is rotated by kTileShift more tiles than the row above — and belongs to rank s % world as local tile
// s / world.  Where tiles_x is a multiple of the world size (1920 px wide: 120 tiles, 8 ranks) plain row-major numbering would hand every rank whole
// tile COLUMNS; the rotation makes the pattern diagonal: rank = (tx + 3 ty) % world there.
constexpr uint32_t kTileShift = 3;
__host__ __device__ inline uint32_t tile_slot(const uint32_t tx, const uint32_t ty, const uint32_t tiles_x) { return ty * tiles_x + (tx + (kTileShift * ty) % tiles_x) % tiles_x; }
// inverse: slot -> (tx, ty)
__host__ __device__ inline void slot_tile(const uint32_t slot, const uint32_t tiles_x, uint32_t &tx, uint32_t &ty)
{
    ty = slot / tiles_x;
    const uint32_t rotated = slot - ty * tiles_x;
    tx = (rotated + tiles_x - (kTileShift * ty) % tiles_x) % tiles_x;
}
// Division of a 32-bit unsigned by a divisor that is a kernel argument, as a multiply-high, a subtract and two shifts (the round-up method of Granlund &
// Montgomery as libdivide's "branchfree" form): q = mulhi(m, x); q = (((x - q) >> 1) + q) >> s.  Exact for every x and every d >= 1 (tests/test_host_utils.py
// sweeps it).  The frame kernels turned a claimed work index into (frame, tile, pixel) with three to four real integer divisions — ~25 VALU each, a quarter of
// a camera round of the packet kernel once the culls had removed the arithmetic around them.
struct FastDiv {
    uint32_t m, s, d;
};
inline FastDiv fast_div_make(const uint32_t d)
{
    FastDiv f{0u, 0u, d};
    if (d <= 1u) return f;  // (d = 1: handled by the select in fast_div)
    uint32_t l = 0;
    while ((1ull << (l + 1)) <= d) l += 1;  // floor(log2(d))
    if ((d & (d - 1u)) == 0u) {  // a power of two: q = 0, ((x - 0) >> 1) >> (l - 1) = x >> l
        f.s = l - 1u;
        return f;
    }
    f.s = l;
    f.m = static_cast<uint32_t>((((1ull << (l + 1)) - d) << 32) / d + 1ull);  // floor(2^(33 + l) / d) - 2^32 + 1: the low 32 bits of the 33-bit multiplier
    return f;
}
__host__ __device__ inline uint32_t fast_div(const uint32_t x, const FastDiv f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t q = __umulhi(f.m, x);
#else
    const uint32_t q = static_cast<uint32_t>((static_cast<unsigned long long>(f.m) * x) >> 32);
#endif
    const uint32_t t = (((x - q) >> 1) + q) >> f.s;
    return f.d == 1u ? x : t;
}
// The packet kernel's claim order (FrameParams::perm_*): block b of a frame's claim order -> the frame's block in tile-linear order.  A bijection of [0, groups << shift)
// for a stride coprime to `groups`; (b >> shift) * stride stays inside 32 bits (rvpt_abi.hip: plan_claim_order bounds the stride).  tests/test_host_utils.py walks it.
__host__ __device__ inline uint32_t claim_order_block(const uint32_t b, const uint32_t groups, const uint32_t stride, const uint32_t shift, const FastDiv div_groups)
{
    const uint32_t prod = (b >> shift) * stride;
    const uint32_t g = prod - fast_div(prod, div_groups) * groups;
    return (g << shift) | (b & ((1u << shift) - 1u));
}
#ifndef RV_MAX_CLAIM_UNITS
#define RV_MAX_CLAIM_UNITS 8
#endif
constexpr uint32_t kUnit = 16;                             // work indices per claim unit (one tile row)
constexpr uint32_t kMaxClaimUnits = RV_MAX_CLAIM_UNITS;    // units per claim (8 = half a 16x16 tile)
#ifndef RV_CLAIM_SHARDS
#define RV_CLAIM_SHARDS 8
#endif
constexpr uint32_t kClaimShards = RV_CLAIM_SHARDS;  // dynamic work counters (one cache line each)
constexpr uint32_t kShardStride = 16;        // unsigned long long words between counters (128 B)
constexpr uint32_t kCounterWords = kShardStride * (kClaimShards + 1);  // + the exited-wave counter
// RVPT_HIP_COUNT_SEGMENTS: every wave adds its segments and samples at its exit.  ONE pair of words took 2 x 2048 same-address atomics per one-frame launch — an L2
// word sustains ~90 per microsecond: 14 us on a 44-us frame (one frame per launch 47 000 -> 35 600 Msamples/s with the flag).  64 pairs, one cache line each, picked by the
// wave's index; rvpt_hip_get_stats adds them up.
constexpr uint32_t kStatStripes = 64, kStatStride = 16;  // (words of 8 bytes)
constexpr uint32_t kWaveChunk = 32;         // triangles per LDS window of the streamed kernel: 2 KiB
#ifndef RV_STREAM_DEPTH
#define RV_STREAM_DEPTH 2
#endif
constexpr uint32_t kStreamDepth = RV_STREAM_DEPTH;  // windows per WAVE: window c + kStreamDepth - 1 is requested before window c is waited for.
                                                    // 2 = double buffering.  3 and 4 measured no different (tools/archive/ab_stream_depth.sh,
                                                    // profiles/r03_stream_depth.txt: 1 M triangles 7.1-7.4e11 tests/s at 512x288, 9 164
                                                    // triangles 1.27e12 at 1080p, any depth): the stream is not what the loop waits for
constexpr uint32_t kResidentMaxTris = 1024; // <= 64 KiB of prepared triangles stay resident in LDS
constexpr uint32_t kResidentMaxMats = 64;   // materials staged in LDS beside them (else read from HBM/L2)
constexpr uint32_t kBvhStackDepth = 64;     // intersection.glsl:363
constexpr uint32_t kBvhResidentBytes = 48 * 1024;  // nodes + triangles + materials up to this size live in LDS
constexpr uint32_t kWideChildren = 4;              // children per node of the wide form of the tree (rvpt_bvh4.hip; bvh_wide.cpp: build_wide_nodes)
constexpr uint32_t kWideEmpty = 0xFFFFFFFFu;       // head word of an unused child slot of a wide node
constexpr uint32_t kWideMaxNodes = 1u << 25;       // the wide walks address a node as a 32-bit byte offset (node << 7): larger trees keep the binary walk
#ifndef RV_BVH4_TOP_QUADS
#define RV_BVH4_TOP_QUADS 8
#endif
constexpr uint32_t kWideTopQuads = RV_BVH4_TOP_QUADS;  // float4 (16 B) between two wide nodes in the LDS copy of the tree top: 9 = 144 B, bank-conflict free (rvpt_bvh4.hip)

// One per-pixel sample mean awaiting the blend: 12 bytes (round 5; the alpha of the accumulator is 0 whatever the sample's would be, compute_pass.comp:165 —
// a 16-byte record was a quarter of the blend's read traffic and of the frame kernels' stores for nothing).  A wave's 64 records are 768 contiguous bytes.
struct SampleRGB {
    float x, y, z;
};

// Everything one frame's kernel needs, passed by value (kernarg segment -> SGPRs).
struct FrameParams {
    // scene (device pointers)
    const float4 *tris;         // n_tris x 4 float4, reference Triangle records as uploaded (integrator_Hart only)
    const float4 *prep;         // n_tris x 4 float4, prepared triangles
    const uint32_t *mat_index;  // n_tris
    const float4 *unit_n;       // n_tris: (normalize(n), 0) — intersect_scene's normalisation of the hit triangle's normal (intersection.glsl:511-513), the same function
                                // of the same n evaluated once per triangle by prepare_triangles instead of once per hit (same bits: IEEE sqrt / divide, no contraction)
    const float4 *mats;         // n_mats x 3 float4 (albedo, emission, data)
    const float4 *nodes;        // n_nodes x 2 float4 (rvpt_bvh_node), BVH contexts only
    // image
    float4 *accum;                   // this rank's tile-linear RGBA32F accumulator, n_work entries
    SampleRGB *sample_out;           // non-null: store this frame's per-pixel sample mean here and leave the
                                     // temporal blend to blend_accumulate (frames overlap in flight)
    unsigned long long *counter;     // kClaimShards work counters + exited-wave counter (all 0 between launches)
    unsigned long long *stats;       // [0] segments, [1] samples; nullptr = do not count
    unsigned long long *timeline;    // optional per-wave timestamps (RVPT_HIP_TIMELINE), 8 words per wave
    uint32_t n_tris;
    uint32_t n_work;   // work items of this launch: frames in the launch * n_work_frame
    uint32_t n_work_frame;  // owned tiles * 256
    FastDiv div_work_frame, div_tiles_x;  // fast_div by n_work_frame and by tiles_x (fill_frame_params)
    uint32_t n_waves;  // wavefronts in this launch
    uint32_t n_mats;
    uint32_t n_nodes;       // BVH contexts
    uint32_t stack_levels;  // BVH traversal stack entries per lane (the tree's height, <= kBvhStackDepth)
    uint32_t stack_lds_levels;  // ... of which this many (the bottom of the stack) live in LDS; deeper entries go to stack_overflow
    uint32_t *stack_overflow;   // [stack_levels - stack_lds_levels][threads of the launch], HBM-resident BVH kernel only
    uint32_t head_shift;    // BVH kernels: a stack slot's second word is the stacked node's (first, count) pair packed as first | count << head_shift
                            // (0: the tree's leaf sizes do not fit beside its indices — the slot holds the node index and a pop fetches the pair)
    uint32_t bvh_refill;    // BVH kernels: hand out new queries once this many lanes of a packet wait for one
    uint32_t bvh_leaf_batch;  // BVH kernels: run the parked leaves once this many lanes of a packet hold one
    uint32_t bvh_top_nodes;   // HBM-resident BVH kernel: this many nodes from the top of the (breadth-first) tree are copied into LDS
    const float4 *wide;       // wide (4-child) form of the tree, 8 float4 per node: minx[4] maxx[4] miny[4] maxy[4] minz[4] maxz[4] head[4] pad (trace_bvh4)
    uint32_t n_wide;          // ... its node count
    uint32_t wide_top_nodes;  // ... of which this many (the upper levels: the layout is breadth first) are copied into LDS
    const float4 *leaf_box;   // trace_bvh4q (p.wide = the 64-byte quantised nodes): the exact box of the leaf whose first triangle is i at [2 i], [2 i + 1]
                              // (bvh_wide.cpp: build_leaf_boxes)
    float slab_extent;        // ... and the largest |coordinate| of the quantised tree (the margin of the conservative child test, rvpt_device.h: quant_slab_setup)
    const uint2 *rects;       // packet kernel: per triangle, the screen rectangle (in 16 x 4 pixel blocks) outside which no camera ray of this launch can hit it
                              // (rvpt_rect.h; camera_rects writes it when the camera or the scene changed); nullptr = no culling
    const float4 *cam_records;  // packet kernel: (n', |dot(v0 - o, n)|) per triangle for the launch's camera (rvpt_early_out.h: camera_record), made by camera_rects; read
                                // by the camera rounds through scalar loads
    const uint32_t *vis;      // packet kernel: the bounce cull — row 2 A + s (vis_words words, bit B) = may a ray that leaves triangle A on side s hit triangle B
                              // (rvpt_packets.hip: bounce_visibility, once per scene); nullptr = no culling
    uint32_t vis_words;       // words per row: ceil(n_tris / 32)
    uint32_t vis_stride;      // ... and words from one row to the next: vis_words rounded up to a multiple of four (16-byte rows, zero padded)
    const float4 *leaf_boxes; // packet kernel (with vis): two float4 per group of kLeafTris consecutive triangles — (lo.xyz, hi.x) (hi.yz, 0, 0), widened (rvpt_vis.h:
                              // bounce_leaf_boxes); a bounce round drops a group no lane's ray can come near.  nullptr = off
    uint32_t bvh_cam_min;     // camera packets (trace_bvh4_resident): at least this many lanes must start a camera ray at once to walk as a packet
    uint32_t bvh_detach;      // ... and the lanes of a node leave the packet (go on per lane) when at most this many of them are in it
    // work distribution plan (units of kUnit work indices, see WavePool): wave w owns units
    // [w*first_units, (w+1)*first_units); units from dyn_base on are dealt from kClaimShards counters,
    // shard s covering [dyn_base + s*shard_len, +shard_len), claim_units at a time
    uint32_t n_units, first_units, claim_units, dyn_base, shard_len;
    // packet kernel: the ORDER in which a frame's 16 x 4 pixel blocks are dealt (round 6).  Block b of a frame's claim order is the frame's block
    // ((b >> perm_shift) * perm_stride mod perm_groups) << perm_shift | (b & ((1 << perm_shift) - 1)): groups of 2^perm_shift consecutive blocks, consecutive
    // groups of the claim order perm_stride groups apart (a stride coprime to perm_groups, near the golden section of it: any run of the order samples the
    // frame evenly, so every 512-item claim holds its share of sky and of model).  perm_groups == 0: the identity (rvpt_abi.hip: plan_interleave)
    uint32_t perm_groups, perm_stride, perm_shift;
    FastDiv div_perm_groups;
    uint32_t width, height, tiles_x;
    uint32_t tile_rank, tile_world;
    // frame (compute_pass.comp:28-40,50-54)
    uint32_t frame;
    uint32_t quantize;  // 1: round the blended mean to UNORM8 each frame (RVPT_HIP_ACCUM_UNORM8)
    // full compute_pass.comp surface (GENERIC kernels; the Kajiya/pinhole kernels ignore these)
    int camera_mode;          // 0 pinhole, 1 ortho, else spherical (compute_pass.comp:102-118)
    int modes[4];             // top-left, top-right, bottom-left, bottom-right integrator (:134-144); 0..9, else Hart
    float split_x, split_y;
    float ortho_scale;        // cam.params.z
    int max_bounces, aa;
    float inv_w, inv_h;
    float cf, inv_cf;
    // camera (compute_pass.comp:44-49): columns 0..2 of the matrix, then the origin (column 3)
    float aspect;
    float cam_w;  // 1 / tan(vfov / 2)
    float cam[12];
};

__global__ void prepare_triangles(const float4 *__restrict__ tris, uint32_t n, float4 *__restrict__ prep,
                                  uint32_t *__restrict__ mat_index, float4 *__restrict__ unit_n);
// the library's own copy of the materials: data.w ("Unused": structs.glsl:31; the live path reads data.x alone, intersection.glsl:51) = 1 / ior, the division integrator_Kajiya does at every hit from outside (integrators.glsl:611)
__global__ void prepare_materials(float4 *__restrict__ mats, uint32_t n);
template <bool REGEN, bool GENERIC> __global__ void trace_brute_resident(const FrameParams p);
template <bool REGEN, bool GENERIC> __global__ void trace_brute_stream(const FrameParams p);
template <bool REGEN, bool RESIDENT, bool GENERIC, bool ORDERED> __global__ void trace_bvh(const FrameParams p);
// the reference's traversal over the 4-wide form of its tree (rvpt_bvh4.hip): lean configuration, reference child order, HBM-resident scenes
__global__ void trace_bvh4(const FrameParams p);
__global__ void trace_bvh4_resident(const FrameParams p);  // ... the whole scene in LDS
__global__ void trace_bvh4_generic(const FrameParams p);           // ... every render / camera mode (GENERIC)
__global__ void trace_bvh4_resident_generic(const FrameParams p);
#if RVPT_HIP_LAB
__global__ void trace_bvh4q(const FrameParams p);  // ... over the 64-byte quantised nodes, exact leaf boxes at the visit (trees whose boxes contain their children)
__global__ void trace_bvh8(const FrameParams p);  // ... over the 8-wide form (rvpt_bvh8.hip): lean configuration, HBM-resident scenes
#endif
__global__ void blend_accumulate(const SampleRGB *__restrict__ samples, float4 *__restrict__ accum, uint32_t n, uint32_t n_frames,
                                 uint32_t frame0, uint32_t quantize);
#if RVPT_HIP_LAB
__global__ void selftest_div_dots(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, uint32_t n);
__global__ void selftest_rcp_sweep(unsigned long long *__restrict__ mismatches);
#endif
__global__ void quantize_rowmajor(const float4 *__restrict__ src, uint32_t n, uint32_t *__restrict__ dst);
__global__ void untile_rgba32f(const float4 *__restrict__ slots, size_t slot_quads, uint32_t n_ranks, uint32_t width,
                               uint32_t height, uint32_t tiles_x, float4 *__restrict__ dst);
__global__ void tile_rgba32f(const float4 *__restrict__ src, uint32_t width, uint32_t height, uint32_t tiles_x,
                             uint32_t tile_rank, uint32_t tile_world, uint32_t n_work, float4 *__restrict__ accum);
__global__ void read_rowmajor(const float4 *__restrict__ accum, uint32_t width, uint32_t height, uint32_t tiles_x,
                              uint32_t tile_rank, uint32_t tile_world, int as_rgba8, void *__restrict__ dst);

}  // namespace rv

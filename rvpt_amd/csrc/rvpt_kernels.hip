// rvpt_kernels.hip — gfx950 path-trace kernels behind the rvpt_hip C ABI.
//
// What the kernels compute is compute_pass.comp::main (reference assets/shaders/compute_pass.comp:121-167): the
// Kajiya path tracer (integrators.glsl:547-677 over intersection.glsl:267-323,361-413,489-517) in the lean
// instances, every other integrator (integrators.glsl:24-543) and camera (camera.glsl:29-99) in the GENERIC ones.
// How they compute it is native to CDNA4:
//
//   * one wavefront = a packet of 64 independent paths, one per lane; a lane that finishes its pixel immediately
//     claims the next pixel of the frame ("ray regeneration") through a ballot + mbcnt prefix over the wave, so
//     lanes stay busy although paths end after 1..8 segments; work-groups are persistent and claim work from
//     sharded counters;
//   * brute force: the ray-independent half of every triangle test is precomputed once per scene upload
//     (prepare_triangles) into a 64-byte record; records are staged through LDS (resident, or streamed in
//     double-buffered windows) and read with wave-uniform ds_read_b128 (broadcast) in the intersect loop; in the
//     frame tail the few live rays are split over the whole wave, k lanes per ray;
//   * BVH: the reference's exact visiting order, "while-while" persistent traversal with per-lane refill, stack in
//     LDS, the whole scene in LDS when it is small;
//   * a frame kernel stores each pixel's sample mean (tile-linear: a wave's claims are consecutive indices, so its
//     16-byte stores coalesce); blend_accumulate folds it into the accumulator, which lets consecutive frames
//     overlap in flight.
//
// Arithmetic follows DESIGN.md "Arithmetic specification" (rvpt_math.h); compiled with -ffp-contract=off and
// -fno-slp-vectorize.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rvpt_device.h"

#ifndef RV_UNROLL
#define RV_UNROLL 4
#endif
#ifndef RV_BVH_MIN_WAVES
#define RV_BVH_MIN_WAVES 1  // HBM-resident BVH kernel: 73 VGPRs = 6-7 waves per SIMD; forcing 8 (64 VGPRs, 9 dwords spilled) measured -24 %
#endif
#ifndef RV_SPLIT_BELOW
#define RV_SPLIT_BELOW 32  // split mode when at most this many lanes of a wave still carry a ray (0 = never)
#endif
#ifndef RV_MIN_WAVES
#define RV_MIN_WAVES 6  // lean brute-force kernel: 80 VGPRs (8 dwords of scratch) for a sixth wave per SIMD, +1.4 % measured
#endif
#define RV_PRAGMA_(x) _Pragma(#x)
#define RV_PRAGMA_UNROLL(n) RV_PRAGMA_(unroll n)

namespace rv {


// ------------------------------------------------------------------------------------------------
// Scene preparation: ray-independent terms of intersect_triangle_fast (intersection.glsl:287-305)
// and the integer material index (intersection.glsl:398: materials[int(triangle.mat_id.x)]).
__global__ void prepare_triangles(const float4 *__restrict__ tris, uint32_t n, float4 *__restrict__ prep,
                                  uint32_t *__restrict__ mat_index, float4 *__restrict__ unit_n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 a = tris[4 * i + 0], b = tris[4 * i + 1], c = tris[4 * i + 2], m = tris[4 * i + 3];
    const f3 v0 = mk(a.x, a.y, a.z);
    const f3 e0 = mk(b.x, b.y, b.z) - v0;
    const f3 e1 = mk(c.x, c.y, c.z) - v0;
    const f3 nn = cross(e0, e1);
    const float a00 = dot(e1, e1);
    const float a01 = -dot(e0, e1);
    const float a11 = dot(e0, e0);
    const float inv_det = 1.0f / fma_(-a01, a01, a00 * a11);  // `x - a*b` of the shader is one fma (intersection.glsl:305)
    prep[4 * i + 0] = make_float4(v0.x, v0.y, v0.z, nn.x);
    prep[4 * i + 1] = make_float4(nn.y, nn.z, e0.x, e0.y);
    prep[4 * i + 2] = make_float4(e0.z, e1.x, e1.y, e1.z);
    prep[4 * i + 3] = make_float4(a00, a01, a11, inv_det);
    mat_index[i] = static_cast<uint32_t>(static_cast<int>(m.x));
    // the normal a hit normalises (intersection.glsl:511-513; shade: rvpt_device.h) depends on the triangle alone: normalize() of the record's own n, once
    const f3 un = normalize(nn);
    unit_n[i] = make_float4(un.x, un.y, un.z, 0.0f);
}

__global__ void prepare_materials(float4 *__restrict__ mats, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    mats[3 * i + 2].w = 1.0f / mats[3 * i + 0].w;  // eta of a ray that enters (shade: `eta = 1 / eta`), IEEE divide as there
}

// ------------------------------------------------------------------------------------------------
// Brute force.  One body, two sources of the prepared triangle records:
//   RESIDENT (n_tris * 64 B <= 64 KiB): the whole scene is copied into LDS once per work-group; after that staging
//            barrier the four waves run independently.
//   STREAM   (larger scenes): every WAVE streams the records through its own ring of kStreamDepth LDS windows of kWaveChunk
//            triangles, filled by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write pass) kStreamDepth - 1
//            windows ahead of the intersect loop (kStreamDepth = 2: one ahead; two or three ahead measured no different).  Nothing is shared between waves, so there is no barrier anywhere in the
//            loop and the waves of a work-group are as independent as in the resident kernel (ray regeneration, split
//            mode for the frame tail).  Each staged record serves the 64 rays of one wave: S * N * 64 / 64 bytes per
//            sample come from L2 (the scene itself is read from HBM once per XCD at most) — against 43 VALU per ray and
//            record the loop stays FP32-VALU-bound by two orders of magnitude.
// The windows of the streamed variant: [wave][kStreamDepth][kWaveChunk] records, then the per-wave owner tables of split mode.

// stage window `c` of the scene into `dst` (wave-uniform LDS address): lane l copies quads l, l + 64, ...
#if !defined(__gfx950__) && defined(__HIP_DEVICE_COMPILE__)
#error "rvpt_kernels.hip is written for gfx950: the streamed windows use its 16-byte global_load_lds (LDS-DMA)"
#endif
__device__ __forceinline__ void stream_issue(const FrameParams &p, float4 *dst, const uint32_t c, const uint32_t lane)
{
    if (p.n_tris == 0) return;  // (an empty scene runs the resident instance, choose_launch; nothing to stage either way)
    const uint32_t last_quad = 4u * p.n_tris - 1u;
#pragma unroll
    for (uint32_t k = 0; k < kWaveChunk * 4u / 64u; ++k) {
        const uint32_t q = min(c * (kWaveChunk * 4u) + k * 64u + lane, last_quad);  // the tail re-reads the last record; never tested
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p.prep + q),
                                         (__attribute__((address_space(3))) void *)(dst + k * 64u), 16, 0, 0);
    }
}
// Window c of this wave has landed and may be read: LDS-DMA is counted by vmcnt and completes in order, so it is enough that at
// most the requests of the `newer` windows issued after it (two instructions each) are still outstanding.
static_assert(kWaveChunk * 4u / 64u == 2u, "stream_wait_for counts two LDS-DMA instructions per window");
__device__ __forceinline__ void stream_wait_for(const uint32_t newer)
{
    if (newer == 0u)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (newer == 1u)
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (newer == 2u)
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
}
static_assert(kStreamDepth >= 2u && kStreamDepth <= 4u, "stream_wait_for covers up to three newer windows");
// The streamed scene as a sequence of windows: start() requests the first kStreamDepth - 1, next(c) requests window
// c + kStreamDepth - 1 into the slot window c - 1 has left, waits for window c and returns its records.
struct WindowStream {
    float4 *ring;
    uint32_t n_chunks;
    __device__ __forceinline__ void start(const FrameParams &p, const uint32_t lane) const
    {
        for (uint32_t w = 0; w + 1u < kStreamDepth && w < n_chunks; ++w) stream_issue(p, ring + w * (kWaveChunk * 4u), w, lane);
    }
    __device__ __forceinline__ const v4f *next(const FrameParams &p, const uint32_t c, const uint32_t lane) const
    {
        const uint32_t ahead = c + kStreamDepth - 1u;
        if (ahead < n_chunks) stream_issue(p, ring + (ahead % kStreamDepth) * (kWaveChunk * 4u), ahead, lane);
        stream_wait_for(min(kStreamDepth - 1u, n_chunks - 1u - c));
        return reinterpret_cast<const v4f *>(ring + (c % kStreamDepth) * (kWaveChunk * 4u));
    }
};

template <bool REGEN, bool GENERIC, bool STREAM>
__device__ __forceinline__ void brute_body(const FrameParams &p)
{
    extern __shared__ __attribute__((aligned(16))) float4 lds_tris[];
    // RESIDENT LDS: [prepared triangles][material index per triangle][materials][per-wave owner table]
    uint32_t *lds_mat_index = reinterpret_cast<uint32_t *>(lds_tris + 4u * p.n_tris);
    float4 *lds_mats = reinterpret_cast<float4 *>(lds_mat_index + ((p.n_tris + 3u) & ~3u));
    const bool mats_in_lds = !STREAM && p.n_mats <= kResidentMaxMats;
    if (!STREAM) {
        for (uint32_t i = threadIdx.x; i < 4u * p.n_tris; i += kBlock) lds_tris[i] = p.prep[i];
        for (uint32_t i = threadIdx.x; i < p.n_tris; i += kBlock) lds_mat_index[i] = p.mat_index[i];
        if (mats_in_lds)
            for (uint32_t i = threadIdx.x; i < 3u * p.n_mats; i += kBlock) lds_mats[i] = p.mats[i];
        __syncthreads();
    }
    const ShadeSrc shade_src = STREAM ? ShadeSrc{p.prep, p.mat_index, p.mats, p.unit_n} : ShadeSrc{lds_tris, lds_mat_index, mats_in_lds ? lds_mats : p.mats, p.unit_n};

    const uint32_t lane = lane_id();
    const uint32_t wave_in_block = uniform(threadIdx.x >> 6);
    const uint32_t wave_id = uniform(blockIdx.x * (kBlock / 64u) + wave_in_block);
    WavePool pool;
    pool.shard = wave_id % kClaimShards;
    Lane L{};
    uint32_t nsmp = 0;
    bool have_pixel = false, need_sample = false;
    // optional timeline: [0] start [1] pool dry [2] end (100 MHz wall clock) [3] rounds | split<<32 [4] regen cycles [5] lane-rounds [6] shade cycles [7] block
    unsigned long long t_start = 0, t_dry = 0;
    uint32_t rounds = 0, split_rounds = 0, lane_rounds = 0;
    unsigned long long t_regen = 0, t_shade = 0;
    if (p.timeline) t_start = wall_clock64();

    // per-wave scratch: lane id of the r-th active ray (split mode)
    uint32_t *owner_of_rank = STREAM ? reinterpret_cast<uint32_t *>(lds_tris + (kBlock / 64u) * kStreamDepth * kWaveChunk * 4u) + wave_in_block * 64u
                                     : reinterpret_cast<uint32_t *>(lds_mats + (mats_in_lds ? 3u * p.n_mats : 0u)) + wave_in_block * 64u;
    const v4f *src = reinterpret_cast<const v4f *>(lds_tris);
    const uint32_t n_chunks = (p.n_tris + kWaveChunk - 1u) / kWaveChunk;
    const WindowStream stream{lds_tris + wave_in_block * (kStreamDepth * kWaveChunk * 4u), n_chunks};  // STREAM: this wave's ring of windows

    for (;;) {
        unsigned long long t_a = 0;
        if (p.timeline) t_a = __builtin_amdgcn_s_memtime();
        regenerate<REGEN, GENERIC>(pool, p, lane, wave_id, have_pixel, need_sample, L);
        if (ballot(have_pixel) == 0) break;
        if (have_pixel && need_sample) {
            begin_sample_t<GENERIC>(L, p);
            need_sample = false;
            nsmp += 1;
        }
        if (p.timeline) t_regen += __builtin_amdgcn_s_memtime() - t_a;
        const bool tracing = have_pixel && wants_trace<GENERIC>(L, p);
        const uint64_t active = ballot(tracing);
        const uint32_t n_active = static_cast<uint32_t>(__builtin_popcountll(active));
        if (p.timeline) {
            rounds += 1;
            lane_rounds += n_active;
            if (n_active <= RV_SPLIT_BELOW) split_rounds += 1;
            if (pool.exhausted && t_dry == 0) t_dry = wall_clock64();
        }
        float closest = kInf;
        uint32_t hit = 0xFFFFFFFFu;
        if (n_active > RV_SPLIT_BELOW) {
            // ---- packet mode: one ray per lane, every lane walks all triangles (uniform LDS reads) ----
            const f3 o = L.o, d = L.d;
            if (!STREAM) {
                if (tracing) intersect_run<RV_UNROLL>(src, 0u, p.n_tris, o, d, closest, hit);
            } else {
                stream.start(p, lane);
                for (uint32_t c = 0; c < n_chunks; ++c) {
                    const v4f *buf = stream.next(p, c, lane);  // window c has landed; the next kStreamDepth - 1 fly during the loop
                    const uint32_t first = c * kWaveChunk;
                    if (tracing) intersect_run<RV_UNROLL>(buf, first, min(kWaveChunk, p.n_tris - first), o, d, closest, hit);
                }
            }
        } else if (n_active > 0) {
            // ---- split mode (frame tail): the few live rays are spread over the whole wave, k = 64/n lanes
            // per ray, lane s of a group testing triangles s, s+k, s+2k, ... (of every window, when streaming); a
            // lexicographic (t, index) min-reduction over the group reproduces the sequential closest hit exactly
            // (first index wins ties, as the strict `t < closest` does in buffer order).
            const uint32_t k = 64u / n_active;  // lanes per ray (>= 2); lanes >= n_active*k idle this round
            const uint32_t rank = prefix_rank(active);
            if (tracing) owner_of_rank[rank] = lane;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t group = lane / k, slice = lane - group * k;
            const bool helper = group < n_active;
            const uint32_t owner = helper ? owner_of_rank[group] : lane;
            const f3 o = mk(__shfl(L.o.x, owner, 64), __shfl(L.o.y, owner, 64), __shfl(L.o.z, owner, 64));
            const f3 d = mk(__shfl(L.d.x, owner, 64), __shfl(L.d.y, owner, 64), __shfl(L.d.z, owner, 64));
            float c = kInf;
            uint32_t h = 0xFFFFFFFFu;
            if (!STREAM) {
                if (helper) {
#pragma unroll 2
                    for (uint32_t i = slice; i < p.n_tris; i += k) {
                        const PrepTri t = unpack(src[4 * i + 0], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]);
                        test_triangle(t, o, d, i, c, h);
                    }
                }
            } else {
                stream.start(p, lane);
                for (uint32_t w = 0; w < n_chunks; ++w) {
                    const v4f *buf = stream.next(p, w, lane);
                    const uint32_t first = w * kWaveChunk, count = min(kWaveChunk, p.n_tris - first);
                    if (helper) {
#pragma unroll 2
                        for (uint32_t i = slice; i < count; i += k) {
                            const PrepTri t = unpack(buf[4 * i + 0], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]);
                            test_triangle(t, o, d, first + i, c, h);
                        }
                    }
                }
            }
            // tree reduction towards slice 0 of every group (k need not be a power of two)
            for (uint32_t m = 1; m < k; m <<= 1) {
                const float c2 = __shfl(c, lane + m, 64);
                const uint32_t h2 = __shfl(h, lane + m, 64);
                const bool take = (slice + m < k) & ((c2 < c) | ((c2 == c) & (h2 < h)));
                c = take ? c2 : c;
                h = take ? h2 : h;
            }
            closest = __shfl(c, rank * k, 64);
            hit = __shfl(h, rank * k, 64);
        }
        unsigned long long t_b = 0;
        if (p.timeline) t_b = __builtin_amdgcn_s_memtime();
        if (have_pixel) {
            bool done = true;
            f3 radiance = mk(0.0f, 0.0f, 0.0f);
            if (tracing) {
                L.nseg += 1;
                done = shade_t<GENERIC>(L, p, shade_src, hit, closest, radiance);
            }
            retire(L, p, done, radiance, have_pixel, need_sample);
        }
        if (p.timeline) t_shade += __builtin_amdgcn_s_memtime() - t_b;
    }
    if (p.timeline && lane == 0) {
        unsigned long long *t = p.timeline + 8ull * wave_id;
        t[0] = t_start;
        t[1] = t_dry;
        t[2] = wall_clock64();
        t[3] = rounds | (static_cast<unsigned long long>(split_rounds) << 32);
        t[4] = t_regen;
        t[5] = lane_rounds;
        t[6] = t_shade;
        t[7] = blockIdx.x;
    }
    wave_exit(p, lane, L.nseg, nsmp);
}

template <bool REGEN, bool GENERIC>
__global__ __launch_bounds__(kBlock, GENERIC ? 1 : RV_MIN_WAVES) void trace_brute_resident(const FrameParams p)
{
    brute_body<REGEN, GENERIC, false>(p);
}

template <bool REGEN, bool GENERIC>
__global__ __launch_bounds__(kBlock, GENERIC ? 1 : RV_MIN_WAVES) void trace_brute_stream(const FrameParams p)
{
    brute_body<REGEN, GENERIC, true>(p);
}

// ------------------------------------------------------------------------------------------------
// BVH traversal (intersection.glsl:361-413).  The reference walks the tree depth first, left child first, testing a
// node's box when the node comes off the stack.  The kernel tests the boxes of BOTH children when it is at their
// parent (one 64-byte fetch in the device layout) and only stacks a right child whose box passed:
//   * a box that fails against the current closest_t also fails later (closest_t only shrinks, the slab test is
//     monotone in it), so not stacking it drops a visit that would have had no effect;
//   * a stacked child passed with t_exit >= entry, so the reference's test at pop time, min(t_exit, closest_t) >=
//     entry, is exactly closest_t >= entry.  A slot is two words — the exact entry distance and the child's own
//     (first, count) pair packed into one word (FrameParams::head_shift; the LDS-resident instance and trees whose leaf
//     sizes do not fit keep the node index there and fetch the pair) — so a pop is two LDS reads and one compare:
//     no box is fetched twice and nothing is fetched at a pop (a step's dependent chain is what bounds this kernel).
// The sequence of boxes that pass, of leaves visited and of triangle tests — hence closest_t and the hit — is the
// reference's, with half the dependent fetches per ray.  ORDERED is the reference's own "TODO: Order the children on
// the stack" (intersection.glsl:405), opt-in, with its own oracle variant: the nearer child (smaller entry distance,
// left on ties) is visited first and the farther one stacked.
// The stack lives in LDS, one column per lane (stack[level][thread], conflict-free), `stack_levels` deep (the tree's
// height, validated at upload; at most the reference's 64).

template <bool REGEN, bool RESIDENT, bool GENERIC, bool ORDERED>
__global__ __launch_bounds__(kBlock, (GENERIC || RESIDENT) ? 1 : RV_BVH_MIN_WAVES) void trace_bvh(const FrameParams p)
{
    // LDS: [stack: stack_levels x kBlock u32] and, when RESIDENT (small scenes), copies of the nodes, the
    // prepared triangles, the material indices and the materials: traversal is a chain of dependent fetches, so
    // serving them at LDS latency instead of L2 latency is what this kernel is bound by.
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    float4 *lds_nodes = reinterpret_cast<float4 *>(lds_stack + 2u * p.stack_lds_levels * kBlock);  // two words per stack slot
    float4 *lds_prep = lds_nodes + 2u * p.n_nodes;
    uint32_t *lds_mat_index = reinterpret_cast<uint32_t *>(lds_prep + 4u * p.n_tris);
    float4 *lds_mats = reinterpret_cast<float4 *>(lds_mat_index + ((p.n_tris + 3u) & ~3u));
    if (RESIDENT) {
        for (uint32_t i = threadIdx.x; i < 2u * p.n_nodes; i += kBlock) lds_nodes[i] = p.nodes[i];
        for (uint32_t i = threadIdx.x; i < 4u * p.n_tris; i += kBlock) lds_prep[i] = p.prep[i];
        for (uint32_t i = threadIdx.x; i < p.n_tris; i += kBlock) lds_mat_index[i] = p.mat_index[i];
        for (uint32_t i = threadIdx.x; i < 3u * p.n_mats; i += kBlock) lds_mats[i] = p.mats[i];
        __syncthreads();
    }
    // HBM-resident scenes: the top of the tree in LDS.  The device layout is breadth-first (upload_scene), so the first
    // bvh_top_nodes records are the upper levels — where most visits happen (Cornell + model: 78 % of all node visits touch
    // the first 256 nodes) — and an access served by LDS costs 64 clocks instead of the 180-290 of a vL1D / L2 hit
    // (tools/microbench/latency_probe.hip): a step's dependent chain is what this kernel is bound by (profiles/EXPERIMENTS.md: what binds the binary walk).
    float4 *lds_top = lds_nodes;  // same place as the resident copy: right behind the stack
    const uint32_t top_nodes = RESIDENT ? 0u : p.bvh_top_nodes;
    if (!RESIDENT && top_nodes) {
        for (uint32_t i = threadIdx.x; i < 2u * top_nodes; i += kBlock) lds_top[i] = p.nodes[i];
        __syncthreads();
    }
    const float4 *nodes = RESIDENT ? lds_nodes : p.nodes;
    const v4f *prep = reinterpret_cast<const v4f *>(RESIDENT ? lds_prep : p.prep);
    const ShadeSrc shade_src = RESIDENT ? ShadeSrc{lds_prep, lds_mat_index, lds_mats, p.unit_n} : ShadeSrc{p.prep, p.mat_index, p.mats, p.unit_n};
    const uint32_t top_level = p.stack_levels - 1u;
    // LDS-resident scenes fetch a popped node's pair from LDS (measured faster there than packing it into the slot)
    const uint32_t head_shift = RESIDENT ? 0u : p.head_shift;
    // Entries [0, lds_levels) of a lane's stack live in LDS; a traversal that stacks more far children than that (rare: the
    // stack is sized for the tree's height, the typical depth is a handful) keeps the rest in global memory, one coalesced
    // column per level.  With the LDS freed, the top of the tree fits beside the stack without costing occupancy.
    const uint32_t lds_levels = p.stack_lds_levels;
    uint32_t *const ovf = p.stack_overflow + (static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x);
    const size_t ovf_stride = static_cast<size_t>(gridDim.x) * kBlock;

    const uint32_t lane = lane_id();
    const uint32_t wave_id = uniform(blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6));
    WavePool pool;
    pool.shard = wave_id % kClaimShards;
    Lane L{};
    uint32_t nsmp = 0;
    bool have_pixel = false, need_sample = false;

    // Traversal lengths differ wildly between the rays of a packet (measured: 14 % lane utilisation on the Cornell
    // scene when a round waits for its slowest ray), so the loop below is "while-while": lanes advance node by
    // node, and as soon as enough of them have finished their traversal those lanes alone shade, continue their
    // path or claim a new pixel and re-enter the traversal loop beside the lanes that are still walking.
    enum { S_IDLE = 0, S_TRAV = 1, S_HIT = 2 };
    int state = S_IDLE;
    bool walking = false;  // a traversal is in progress (state == S_TRAV and the root has been tested)
    float closest = kInf;
    uint32_t hit = 0xFFFFFFFFu, sp = 0;
    uint32_t cur = 0;                            // inner node being processed (its box passed): index of its children pair
    uint32_t leaf_first = 0, leaf_count = 0;     // leaf reached (its box passed), waiting for its triangle tests
    auto enter = [&](const float4 n0) {          // n0.xy = {first_child_or_primitive, primitive_count} (bvh.h:12-19)
        const uint32_t first = __float_as_uint(n0.x), count = __float_as_uint(n0.y);
        cur = first;
        leaf_first = first;
        leaf_count = count;  // 0 for an inner node
    };
    f3 inv = mk(0.0f, 0.0f, 0.0f);
#ifdef RV_BVH_PROFILE  // experiments only (tools/bvh_phase_profile.py): where a packet's time goes, per wave
    unsigned long long pf_refill = 0, pf_inner = 0, pf_leaf = 0, pf_iters = 0, pf_leaf_phases = 0, pf_inner_lanes = 0, pf_leaf_lanes = 0,
                       pf_refill_lanes = 0, pf_refills = 0, pf_t0 = __builtin_amdgcn_s_memtime(), pf_mark = 0, pf_hist = 0, pf_dry_iters = 0;
    uint32_t pf_ld_node = 0, pf_ld_node_lanes = 0, pf_ld_pop = 0, pf_ld_pop_lanes = 0, pf_ld_leaf = 0, pf_ld_leaf_lanes = 0;  // wave-level global load instructions by source (summed over lanes at exit)
    auto pf_count = [&](uint32_t &insts, uint32_t &lanes, const uint32_t n) {  // call in divergent code: the lowest active lane counts the instruction(s)
        const uint64_t act = ballot(true);
        if (lane == static_cast<uint32_t>(__builtin_ctzll(act))) insts += n, lanes += n * static_cast<uint32_t>(__builtin_popcountll(act));
    };
#endif

    for (;;) {
#ifdef RV_BVH_PROFILE
        pf_mark = __builtin_amdgcn_s_memtime();
        pf_refills += 1;
        pf_refill_lanes += __builtin_popcountll(ballot(state != S_TRAV));
#endif
        // ---- refill: every lane that is not traversing gets its next query, until nothing more can be handed out
        for (;;) {
            if (state == S_HIT) {  // a finished closest-hit query: one step of the integrator
                f3 radiance = mk(0.0f, 0.0f, 0.0f);
                L.nseg += 1;
                const bool done = shade_t<GENERIC>(L, p, shade_src, hit, closest, radiance);
                state = S_IDLE;
                if (done)
                    retire(L, p, true, radiance, have_pixel, need_sample);
                else
                    state = S_TRAV;
            }
            regenerate<REGEN, GENERIC>(pool, p, lane, wave_id, have_pixel, need_sample, L);
            if (have_pixel && need_sample && state == S_IDLE) {
                begin_sample_t<GENERIC>(L, p);
                need_sample = false;
                nsmp += 1;
                if (wants_trace<GENERIC>(L, p))
                    state = S_TRAV;
                else  // no bounce budget: the integrator returns black without a query
                    retire(L, p, true, mk(0.0f, 0.0f, 0.0f), have_pixel, need_sample);
            }
            if (state == S_TRAV && !walking) {  // start the traversal of L.o, L.d at the root
                closest = kInf;
                hit = 0xFFFFFFFFu;
                inv = mk(1.0f / L.d.x, 1.0f / L.d.y, 1.0f / L.d.z);
                sp = 0;
                const float4 *root = RESIDENT ? nodes : lds_top;  // (compile-time) HBM-resident scenes: the LDS copy of the tree top always holds the root
                const float4 n0 = root[0], n1 = root[1];
                float entry;
                if (slab_entry(L.o, inv, n0, n1, closest, entry)) {
                    enter(n0);
                    walking = true;
                } else {
                    state = S_HIT;  // the ray misses the root box
                }
            }
            const bool more = (have_pixel && state != S_TRAV) || (!have_pixel && !pool.exhausted);
            if (ballot(more) == 0) break;
        }
        if (ballot(state == S_TRAV) == 0) break;  // nothing in flight and nothing left to claim
#ifdef RV_BVH_PROFILE
        pf_refill += __builtin_amdgcn_s_memtime() - pf_mark;
#endif

        // ---- traverse: every iteration each walking lane handles one node; a lane that reaches a leaf parks there
        // (leaf_count > 0) until enough lanes have one, then they run their triangle tests together — inner nodes and
        // leaves cost very different amounts, mixing them in one step would leave most of the packet idle either way.
        for (uint32_t steps = 0;; ++steps) {
#ifdef RV_BVH_PROFILE
            pf_mark = __builtin_amdgcn_s_memtime();
            pf_iters += 1;
            {
                const unsigned long long nw = __builtin_popcountll(ballot(state == S_TRAV && leaf_count == 0));
                pf_inner_lanes += nw;
                pf_hist += 1ull << (16u * static_cast<uint32_t>(nw > 48 ? 3 : nw > 32 ? 2 : nw > 16 ? 1 : 0));  // 4 x 16-bit bins
                if (pool.exhausted) pf_dry_iters += 1;
            }
#endif
            bool need_pop = false;  // this lane's node is finished, take the next candidate from the stack
            if (state == S_TRAV && leaf_count == 0) {
                const uint32_t c = cur;  // sibling pair = one 64-byte line in the device layout
                // ONE address per lane — the pair's line in the LDS copy of the tree top or in global memory — and four FLAT loads off
                // it (flat_load_dwordx4 ... offset:16/32/48): a lane's aperture decides where each is served, no second set of loads
                const float4 *pair = (!RESIDENT && c + 1u < top_nodes) ? lds_top + 2 * c : nodes + 2 * c;
                const float4 a0 = pair[0], a1 = pair[1], b0 = pair[2], b1 = pair[3];
#ifdef RV_BVH_PROFILE
                if (!RESIDENT && c + 1u >= top_nodes) pf_count(pf_ld_node, pf_ld_node_lanes, 4);
#endif
                float e0, e1;
                const bool h0 = slab_entry(L.o, inv, a0, a1, closest, e0);
                const bool h1 = slab_entry(L.o, inv, b0, b1, closest, e1);
                // which child first: the reference always the left one; ORDERED the nearer one, left on ties
                const bool right_first = h1 && (!h0 || (ORDERED && e1 < e0));
                if (h0 && h1) {
                    // two words per slot: the stacked child's exact entry distance and its node index.
                    // The host sized the stack from the tree's height (upload_scene), so sp never passes top_level.
                    const uint32_t far_entry = __float_as_uint(right_first ? e0 : e1);
                    const float4 far_head = right_first ? a0 : b0;  // .xy = the stacked child's (first, count): known now, so a pop need not fetch it
                    const uint32_t far_node = head_shift ? (__float_as_uint(far_head.x) | (__float_as_uint(far_head.y) << head_shift)) : (right_first ? c : c + 1u);
                    report_stack_overflow(p, sp > top_level);
                    const uint32_t at = min(sp, top_level);
                    if (at < lds_levels) {
                        lds_stack[(2u * at + 0u) * kBlock + threadIdx.x] = far_entry;
                        lds_stack[(2u * at + 1u) * kBlock + threadIdx.x] = far_node;
                    } else {
                        ovf[(2u * (at - lds_levels) + 0u) * ovf_stride] = far_entry;
                        ovf[(2u * (at - lds_levels) + 1u) * ovf_stride] = far_node;
                    }
                    sp = at + 1u;  // (= sp + 1 unless the push was clamped: the pops then stay inside the stack)
                }
                if (h0 || h1)
                    enter(right_first ? b0 : a0);
                else
                    need_pop = true;
            }
#ifdef RV_BVH_PROFILE
            pf_inner += __builtin_amdgcn_s_memtime() - pf_mark;
            pf_mark = __builtin_amdgcn_s_memtime();
#endif
            bool run_leaves = true;  // LDS-resident scenes: traversals are short, parking does not pay (measured)
            if (!RESIDENT) {
                const uint32_t at_leaf = static_cast<uint32_t>(__builtin_popcountll(ballot(leaf_count > 0)));
                const uint32_t at_inner = static_cast<uint32_t>(__builtin_popcountll(ballot(state == S_TRAV && leaf_count == 0)));
                run_leaves = at_leaf > 0 && (at_inner == 0 || at_leaf >= p.bvh_leaf_batch);
            }
#ifdef RV_BVH_PROFILE
            if (run_leaves) {
                pf_leaf_phases += 1;
                pf_leaf_lanes += __builtin_popcountll(ballot(leaf_count > 0));
            }
#endif
            if (run_leaves && leaf_count > 0) {
                for (uint32_t i = leaf_first; i < leaf_first + leaf_count; ++i) {
                    const v4f *tp = prep + 4 * i;
                    const PrepTri t = unpack(tp[0], tp[1], tp[2], tp[3]);
                    test_triangle(t, L.o, L.d, i, closest, hit);
#ifdef RV_BVH_PROFILE
                    if (!RESIDENT) pf_count(pf_ld_leaf, pf_ld_leaf_lanes, 4);
#endif
                }
                leaf_count = 0;
                need_pop = true;
            }
#ifdef RV_BVH_PROFILE
            pf_leaf += __builtin_amdgcn_s_memtime() - pf_mark;  // leaf tests (and the decision around them)
            pf_mark = __builtin_amdgcn_s_memtime();
#endif
            if (need_pop) {
                bool found = false;
                while (sp > 0 && !found) {
                    sp -= 1;
                    uint32_t entry_bits, cand;
                    if (sp < lds_levels) {
                        entry_bits = lds_stack[(2u * sp + 0u) * kBlock + threadIdx.x];
                        cand = lds_stack[(2u * sp + 1u) * kBlock + threadIdx.x];
                    } else {
                        entry_bits = ovf[(2u * (sp - lds_levels) + 0u) * ovf_stride];
                        cand = ovf[(2u * (sp - lds_levels) + 1u) * ovf_stride];
                    }
                    // The reference tests the popped node's box against the current closest_t: min(t_exit, closest_t) >= entry.
                    // The node was stacked because t_exit >= entry held, so that test IS closest_t >= entry — with the exact
                    // entry distance on the stack no box has to be fetched again; only the node's (first, count) pair is.
                    if (closest >= __uint_as_float(entry_bits)) {
                        float2 fc;
                        if (head_shift) {  // (wave-uniform) the pair rode on the stack
                            fc.x = __uint_as_float(cand & ((1u << head_shift) - 1u));
                            fc.y = __uint_as_float(cand >> head_shift);
                        } else {
                            const float2 *head = reinterpret_cast<const float2 *>((!RESIDENT && cand < top_nodes) ? lds_top + 2 * cand : nodes + 2 * cand);
                            fc = *head;
                        }
#ifdef RV_BVH_PROFILE
                        if (!RESIDENT && !head_shift && cand >= top_nodes) pf_count(pf_ld_pop, pf_ld_pop_lanes, 1);
#endif
                        enter(make_float4(fc.x, fc.y, 0.0f, 0.0f));
                        found = true;
                    }
                }
                if (!found) {  // nothing left to visit
                    state = S_HIT;
                    walking = false;
                }
            }
#ifdef RV_BVH_PROFILE
            pf_inner += __builtin_amdgcn_s_memtime() - pf_mark;  // pops count as inner-node work
#endif
            if (ballot(state == S_TRAV) == 0) break;
            // lanes that could be given work right now: finished queries, and empty lanes while pixels remain
            const uint32_t waiting = static_cast<uint32_t>(__builtin_popcountll(ballot(state == S_HIT || (!have_pixel && !pool.exhausted))));
            if (waiting >= p.bvh_refill || (waiting > 0 && steps >= 4u * p.bvh_refill)) break;  // (parked lanes stay parked)
        }
    }
#ifdef RV_BVH_PROFILE
    if (p.timeline && lane == 0) {
        unsigned long long *t = p.timeline + 8ull * wave_id;
        t[0] = pf_refill;
        t[1] = pf_inner;
        t[2] = pf_leaf;
        t[3] = pf_iters | (pf_leaf_phases << 32);
        t[4] = pf_inner_lanes | (pf_leaf_lanes << 32);
        t[5] = pf_hist;  // inner iterations with 0-16 / 17-32 / 33-48 / 49-64 lanes walking, 16 bits each
        t[6] = pf_refill_lanes | (pf_refills << 32);
        t[7] = (__builtin_amdgcn_s_memtime() - pf_t0) | (pf_dry_iters << 40);  // iterations after the pixel pool ran dry
    }
    if (p.timeline) {  // (several launches in flight add into the same rows: read the shares, not the totals)
        unsigned long long *t = p.timeline + 8ull * p.n_waves + 8ull * wave_id;
        atomicAdd(&t[0], pf_ld_node), atomicAdd(&t[1], pf_ld_node_lanes), atomicAdd(&t[2], pf_ld_pop), atomicAdd(&t[3], pf_ld_pop_lanes), atomicAdd(&t[4], pf_ld_leaf), atomicAdd(&t[5], pf_ld_leaf_lanes);
    }
#endif
    wave_exit(p, lane, L.nseg, nsmp);
}

// ------------------------------------------------------------------------------------------------
// Temporal blend as its own pass (compute_pass.comp:146-148,162-166): out = (prev*f + sampled) * 1/(f+1), prev
// ignored at frame 0.  Same operations as the fused form in finish_pixel, so the result is bit-identical; being
// separate lets the trace kernels of consecutive frames overlap (they no longer touch the accumulator).
__global__ void blend_accumulate(const SampleRGB *__restrict__ samples, float4 *__restrict__ accum, uint32_t n, uint32_t n_frames,
                                 uint32_t frame0, uint32_t quantize)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // the frames of one launch, oldest first: samples[k * n + i] is frame frame0 + k of pixel i
    f3 prev = mk(0.0f, 0.0f, 0.0f);
    if (frame0 != 0u) {
        const float4 a = accum[i];
        prev = mk(a.x, a.y, a.z);
    }
    for (uint32_t k = 0; k < n_frames; ++k) {
        const uint32_t frame = frame0 + k;
        const SampleRGB sv = samples[static_cast<size_t>(k) * n + i];
        const float cf = static_cast<float>(frame);               // compute_pass.comp:53
        const float inv_cf = 1.0f / static_cast<float>(frame + 1u);  // :54
        prev = store_format(fma3(prev, cf, mk(sv.x, sv.y, sv.z)) * inv_cf, quantize);
    }
    accum[i] = make_float4(prev.x, prev.y, prev.z, 0.0f);
}

#if RVPT_HIP_LAB
// ------------------------------------------------------------------------------------------------
// diagnostics of the arithmetic specification (rvpt_hip_selftest_*): div_dots on operand arrays, and the refined hardware
// reciprocal against the correctly rounded 1/b for every binary32 b of one exponent (grid.y = exponent - 1)
__global__ void selftest_div_dots(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = div_dots(a[i], b[i]);
}
__global__ void selftest_rcp_sweep(unsigned long long *__restrict__ mismatches)
{
    const uint32_t exponent = blockIdx.y + 1;
    const uint32_t mantissa = blockIdx.x * blockDim.x + threadIdx.x;
    const float b = __uint_as_float((exponent << 23) | mantissa);
    float r = __builtin_amdgcn_rcpf(b);
    r = fma_(fma_(-b, r, 1.0f), r, r);
    // ... and v_rcp_f32 is odd: rcp(-b) = -rcp(b) bit for bit (the camera records flip the sign of the plane equation, rvpt_early_out.h)
    const bool odd = __float_as_uint(__builtin_amdgcn_rcpf(-b)) == (__float_as_uint(__builtin_amdgcn_rcpf(b)) ^ 0x80000000u);
    const bool bad = __float_as_uint(r) != __float_as_uint(1.0f / b) || !odd;
    const unsigned long long m = ballot(bad);
    if (m != 0 && (threadIdx.x & 63) == 0) atomicAdd(&mismatches[exponent], static_cast<unsigned long long>(__builtin_popcountll(m)));
}

#endif  // RVPT_HIP_LAB

// ------------------------------------------------------------------------------------------------
// Layout helpers: tile-linear accumulator <-> row-major images.

// row-major RGBA32F -> rgba8 UNORM (same conversion as read_rowmajor); for frames gathered from several ranks
__global__ void quantize_rowmajor(const float4 *__restrict__ src, uint32_t n, uint32_t *__restrict__ dst)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 v = src[i];
    auto q = [](float f) -> uint32_t {
        f = (f > 0.0f) ? f : 0.0f;
        f = (f > 1.0f) ? 1.0f : f;
        return static_cast<uint32_t>(__builtin_floorf(fma_(f, 255.0f, 0.5f)));
    };
    dst[i] = q(v.x) | (q(v.y) << 8) | (q(v.z) << 16) | (q(v.w) << 24);
}

__global__ void untile_rgba32f(const float4 *__restrict__ slots, size_t slot_quads, uint32_t n_ranks, uint32_t width,
                               uint32_t height, uint32_t tiles_x, float4 *__restrict__ dst)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= width || y >= height) return;
    const uint32_t tile = tile_slot(x >> 4, y >> 4, tiles_x);
    const uint32_t rank = tile % n_ranks;
    const uint32_t local_tile = tile / n_ranks;
    const size_t idx = static_cast<size_t>(rank) * slot_quads + static_cast<size_t>(local_tile) * 256u + ((y & 15u) << 4) + (x & 15u);
    dst[static_cast<size_t>(y) * width + x] = slots[idx];
}

// row-major image -> this rank's tile-linear accumulator (rvpt_hip_write_accum)
__global__ void tile_rgba32f(const float4 *__restrict__ src, uint32_t width, uint32_t height, uint32_t tiles_x,
                             uint32_t tile_rank, uint32_t tile_world, uint32_t n_work, float4 *__restrict__ accum)
{
    const uint32_t work = blockIdx.x * blockDim.x + threadIdx.x;
    if (work >= n_work) return;
    uint32_t tx, ty;
    slot_tile((work >> 8) * tile_world + tile_rank, tiles_x, tx, ty);
    const uint32_t gx = tx * 16u + (work & 15u), gy = ty * 16u + ((work & 255u) >> 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (gx < width && gy < height) v = src[static_cast<size_t>(gy) * width + gx];
    accum[work] = v;
}

// this rank's tiles -> row-major image (float or rgba8 UNORM); pixels of foreign tiles become 0
__global__ void read_rowmajor(const float4 *__restrict__ accum, uint32_t width, uint32_t height, uint32_t tiles_x,
                              uint32_t tile_rank, uint32_t tile_world, int as_rgba8, void *__restrict__ dst)
{
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= width || y >= height) return;
    const uint32_t tile = tile_slot(x >> 4, y >> 4, tiles_x);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tile % tile_world == tile_rank) v = accum[static_cast<size_t>(tile / tile_world) * 256u + ((y & 15u) << 4) + (x & 15u)];
    const size_t o = static_cast<size_t>(y) * width + x;
    if (as_rgba8) {
        // clamp to [0,1] (NaN -> 0), scale by 255, round half up (compute_pass.comp:41-42 image format)
        auto q = [](float f) -> uint32_t {
            f = (f > 0.0f) ? f : 0.0f;
            f = (f > 1.0f) ? 1.0f : f;
            return static_cast<uint32_t>(__builtin_floorf(fma_(f, 255.0f, 0.5f)));
        };
        static_cast<uint32_t *>(dst)[o] = q(v.x) | (q(v.y) << 8) | (q(v.z) << 16) | (q(v.w) << 24);
    } else {
        static_cast<float4 *>(dst)[o] = v;
    }
}

// explicit instantiations used by the launcher
#define RV_INST2(K)                                            \
    template __global__ void K<true, false>(const FrameParams);  \
    template __global__ void K<false, false>(const FrameParams); \
    template __global__ void K<true, true>(const FrameParams);   \
    template __global__ void K<false, true>(const FrameParams);
RV_INST2(trace_brute_resident)
RV_INST2(trace_brute_stream)
#define RV_INST4(R, O)                                                            \
    template __global__ void trace_bvh<true, R, false, O>(const FrameParams);  \
    template __global__ void trace_bvh<false, R, false, O>(const FrameParams); \
    template __global__ void trace_bvh<true, R, true, O>(const FrameParams);   \
    template __global__ void trace_bvh<false, R, true, O>(const FrameParams);
RV_INST4(true, false)
RV_INST4(false, false)
RV_INST4(true, true)
RV_INST4(false, true)

}  // namespace rv

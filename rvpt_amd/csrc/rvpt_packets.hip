// rvpt_packets.hip — the brute-force frame kernel for scenes resident in LDS, lean configuration (Kajiya everywhere, pinhole camera):
// every round of a wave is a FULL packet of 64 rays of one kind.
//
// trace_brute_resident (rvpt_kernels.hip) hands a lane its next pixel the moment its pixel is finished, so its packets mix camera rays
// with bounce rays of neighbouring pixels.  Here a wave alternates between two kinds of round and parks paths in a 64-entry queue in
// LDS in between (nothing leaves the chip, nothing waits for another wave):
//
//   camera round   the wave claims 64 consecutive work items = a 16 x 4 pixel block of one tile; 64 camera rays with one origin and
//                  nearly one direction walk the triangles with the packet-uniform early-out (rvpt_early_out.h: 61 % of all
//                  (packet, triangle) pairs of the headline frame skip the second half of the test) on CAMERA RECORDS — the numerator
//                  of the plane distance is one number per triangle for every camera ray of the launch, computed once per work-group;
//                  Round 5: before any of that, the SCREEN RECTANGLES (rvpt_rect.h) — every triangle carries the conservative rectangle of 16 x 4
//                  pixel blocks outside which no camera ray of this launch can hit it; the 64 rays of a round come from ONE block, so lane i tests
//                  rectangle 64 k + i against the wave's block, a ballot gives the candidate triangles and the loop walks only those (default scene,
//                  default camera: 1.7 % of all (block, triangle) pairs).  A superset test: the image cannot change.
//   bounce round   taken as soon as the paths still alive in the lanes plus the parked ones make a full packet (or no pixels are
//                  left): the lanes without a path pop parked ones — ballot + mbcnt — and 64 bounce rays walk the triangles with the
//                  plain loop.
//
// After either round the lanes shade (integrators.glsl:576-671), finished pixels store their sample mean, and what is still alive stays
// in its lane; before a camera round the survivors are parked (ballot + mbcnt compaction into the queue).  The queue never holds more
// than 63 paths at a round boundary.  Per-pixel operations and their order are trace_brute_resident's (the RNG state is keyed on the
// pixel and travels with the path; the samples of a pixel are sequential), so the image is the same, bit for bit.
// Work per pixel: 1 camera segment at ~0.7 of a full pass + (S - 1) bounce segments, against S full passes — S = 1.44 on the headline frame.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rvpt_device.h"
#include "rvpt_early_out.h"
#include "rvpt_packets.h"
#include "rvpt_vis.h"

#ifndef RV_PACKETS_MIN_WAVES
#define RV_PACKETS_MIN_WAVES 6
#endif
#ifndef RV_PACKETS_SPLIT_BELOW
#define RV_PACKETS_SPLIT_BELOW 32u  // split mode when at most this many lanes of a round carry a ray (and nothing is parked)
#endif
#ifndef RV_PACKETS_TIMELINE
#define RV_PACKETS_TIMELINE 0  // 1 (tools/packets_timeline.py builds it): per-wave timestamps and round counts for RVPT_HIP_TIMELINE — six registers the loops want otherwise
#endif
#ifndef RV_BOX_BATCH
#define RV_BOX_BATCH 1  // leaf boxes of a bounce round requested together (1, 2 or 4: a whole word's); more boxes in flight = fewer waits and more SGPRs: 2 and 4 measured slower
#endif
#ifndef RV_PACKETS_BOUNCE_EARLY
#define RV_PACKETS_BOUNCE_EARLY 0  // 1: the early-out loop in bounce rounds as well (experiment: incoherent packets rarely fail the pre-test together)
#endif

namespace rv {

namespace {

// A parked path: everything a lane needs to go on — 18 words, stored field-major ([field][entry]: lanes of a wave touch consecutive
// words, conflict-free).  AA1 (one sample per pixel, the headline's configuration: its own kernel instance since round 6): a pixel's sum of finished samples is zero
// for as long as its only path is in flight, so the queue holds 15 words per path — with the camera records out of LDS (scalar loads) a work-group then needs
// 26.3 KB instead of 31.7 and SIX fit a CU instead of five.
template <bool AA1>
__device__ __forceinline__ void park(uint32_t *q, const uint32_t at, const Lane &L, const uint32_t leave)
{
    const float f[12] = {L.o.x, L.o.y, L.o.z, L.d.x, L.d.y, L.d.z, L.thr.x, L.thr.y, L.thr.z, L.col.x, L.col.y, L.col.z};
#pragma unroll
    for (uint32_t k = 0; k < 12u; ++k) q[k * 64u + at] = __float_as_uint(f[k]);
    q[12u * 64u + at] = L.rng;
    q[13u * 64u + at] = L.work;
    // sample < aa <= 1023, bounce < max_bounces <= 1023 (rvpt_abi.hip: choose_launch), leave <= 2 * kResidentMaxTris - 1 = 2047 or all ones -> 4095
    q[14u * 64u + at] = static_cast<uint32_t>(L.sample) | (static_cast<uint32_t>(L.bounce) << 10) | (leave << 20);
    if (!AA1) {
        q[15u * 64u + at] = __float_as_uint(L.sum.x);
        q[16u * 64u + at] = __float_as_uint(L.sum.y);
        q[17u * 64u + at] = __float_as_uint(L.sum.z);
    }
}
template <bool AA1>
__device__ __forceinline__ void unpark(const uint32_t *q, const uint32_t at, Lane &L, uint32_t &leave)
{
    float f[12];
#pragma unroll
    for (uint32_t k = 0; k < 12u; ++k) f[k] = __uint_as_float(q[k * 64u + at]);
    L.o = mk(f[0], f[1], f[2]);
    L.d = mk(f[3], f[4], f[5]);
    L.thr = mk(f[6], f[7], f[8]);
    L.col = mk(f[9], f[10], f[11]);
    L.rng = q[12u * 64u + at];
    L.work = q[13u * 64u + at];
    const uint32_t packed = q[14u * 64u + at];
    L.sample = static_cast<int>(packed & 0x3FFu);
    L.bounce = static_cast<int>((packed >> 10) & 0x3FFu);
    leave = packed >> 20;
    leave = (leave == 0xFFFu) ? 0xFFFFFFFFu : leave;
    L.sum = AA1 ? mk(0.0f, 0.0f, 0.0f) : mk(__uint_as_float(q[15u * 64u + at]), __uint_as_float(q[16u * 64u + at]), __uint_as_float(q[17u * 64u + at]));
}

// the triangles base + (set bits of `todo`) — a wave-uniform list — in ascending order, four tests' arithmetic scheduled together as in intersect_run<4>
__device__ __forceinline__ void intersect_listed(const v4f *src, const uint32_t base, uint32_t todo, const f3 o, const f3 d, float &closest, uint32_t &hit)
{
    while (__builtin_popcount(todo) >= 4) {
        uint32_t j[4];
        OpenTest r[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) {
            j[k] = base + static_cast<uint32_t>(__builtin_ctz(todo));
            todo &= todo - 1u;
            r[k] = test_triangle_open(unpack(src[4 * j[k] + 0], src[4 * j[k] + 1], src[4 * j[k] + 2], src[4 * j[k] + 3]), o, d);
        }
        asm volatile("" ::"v"(r[0].tt), "v"(r[0].m), "v"(r[0].s), "v"(r[1].tt), "v"(r[1].m), "v"(r[1].s), "v"(r[2].tt), "v"(r[2].m), "v"(r[2].s), "v"(r[3].tt),
                     "v"(r[3].m), "v"(r[3].s));
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) accept_hit(r[k], j[k], closest, hit);
    }
    while (todo != 0u) {
        const uint32_t j = base + static_cast<uint32_t>(__builtin_ctz(todo));
        todo &= todo - 1u;
        accept_hit(test_triangle_open(unpack(src[4 * j + 0], src[4 * j + 1], src[4 * j + 2], src[4 * j + 3]), o, d), j, closest, hit);
    }
}

}  // namespace

// CULLS: the instance for launches that ride with all three exact culls and a work plan of whole blocks (the default for a scene with a table) — the walks
// without a cull (split mode, the packet-uniform early-out over every triangle, the plain loop) are not in it
// ORDER (CULLS only): the instance carries the interleaved claim order of short launches (FrameParams::perm_*); the batched launches' instance does not — the
// map's six scalars cost eleven SGPR spills in a kernel that never uses them
template <bool AA1, bool CULLS, bool ORDER = true>
__device__ __forceinline__ void packets_body(const FrameParams &p)
{
    constexpr uint32_t kPathWords = AA1 ? kPacketQueueWordsAA1 : kPacketQueueWords;
    // LDS: [prepared triangles][material index per triangle][materials][camera records: (n, dot(v0 - o, n)) per triangle][per wave: the queue of parked paths]
    extern __shared__ __attribute__((aligned(16))) float4 lds_tris[];
    uint32_t *lds_mat_index = reinterpret_cast<uint32_t *>(lds_tris + 4u * p.n_tris);
    float4 *lds_mats = reinterpret_cast<float4 *>(lds_mat_index + ((p.n_tris + 3u) & ~3u));
    for (uint32_t i = threadIdx.x; i < 4u * p.n_tris; i += kBlock) lds_tris[i] = p.prep[i];
    for (uint32_t i = threadIdx.x; i < p.n_tris; i += kBlock) lds_mat_index[i] = p.mat_index[i];
    for (uint32_t i = threadIdx.x; i < 3u * p.n_mats; i += kBlock) lds_mats[i] = p.mats[i];
    // the camera records — (n', |dot(v0 - o, n)|) per triangle for the launch's camera (rvpt_early_out.h) — are read through SCALAR loads since round 6 (a camera round
    // looks at record j for a wave-uniform j: s_load_dwordx4 through the constant address space); camera_rects makes them with the rectangles, once per launch camera
    typedef const __attribute__((address_space(4))) v4f *ConstRecords;
    const ConstRecords cam_records = (ConstRecords)(reinterpret_cast<uintptr_t>(p.cam_records));
    uint2 *lds_rect = reinterpret_cast<uint2 *>(lds_mats + 3u * p.n_mats);  // the screen rectangles (p.rects != nullptr)
    if (CULLS || p.rects != nullptr)
        for (uint32_t i = threadIdx.x; i < p.n_tris; i += kBlock) lds_rect[i] = p.rects[i];
    __syncthreads();
    const ShadeSrc shade_src{lds_tris, lds_mat_index, lds_mats, p.unit_n};
    const v4f *src = reinterpret_cast<const v4f *>(lds_tris);

    const uint32_t lane = lane_id();
    const uint32_t wave_in_block = uniform(threadIdx.x >> 6);
    const uint32_t wave_id = uniform(blockIdx.x * (kBlock / 64u) + wave_in_block);
    uint32_t *queue = reinterpret_cast<uint32_t *>((CULLS || p.rects != nullptr) ? lds_rect + ((p.n_tris + 1u) & ~1u) : lds_rect) + wave_in_block * (kPathWords * 64u);  // (16-byte aligned)
    uint32_t parked = 0;  // paths in the queue (wave-uniform)

    WavePool pool;
    pool.shard = wave_id % kClaimShards;
    Lane L{};
    bool has = false;  // this lane holds a live path whose next segment is to be traced
    uint32_t leave = 0xFFFFFFFFu;  // ... and where that segment leaves from: 2 * triangle + side (shade), all ones = anywhere (a camera ray)
    uint32_t nsmp = 0;
    // optional timeline (RVPT_HIP_TIMELINE): [0] start [1] pool dry [2] end (100 MHz wall clock) [3] camera rounds | bounce rounds << 32 [4] split rounds | lane-rounds << 32
    unsigned long long t_start = 0, t_dry = 0;
    uint32_t n_cam = 0, n_bounce = 0, n_split = 0, lane_rounds = 0, n_listed = 0;  // [5] triangles walked by the culled bounce rounds
    uint32_t n_after_dry = 0, lanes_after_dry = 0;  // [6] rounds | lane-rounds << 32 after the pool ran dry for this wave; [7] the wave's last camera round (clock)
    unsigned long long t_last_cam = 0;
    if (RV_PACKETS_TIMELINE && p.timeline) t_start = wall_clock64();
    // (instrumented build) where a wave's time goes, in shader clocks: [0] loop head + claims [1] camera round set-up (park, decode, begin_sample) [2] bounce round set-up (unpark)
    // [3] camera walk (rectangles + tests) [4] bounce culls (union of rows, leaf boxes) [5] bounce triangle tests [6] shade + sample store [7] the other walks (no cull)
    uint32_t ph[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // (a wave lives ~10^7 clocks: 32 bits hold it)
    uint32_t ph_last = 0;
    if (RV_PACKETS_TIMELINE && p.timeline) ph_last = uniform(static_cast<uint32_t>(__builtin_readcyclecounter()));
#define RV_PHASE(k)                                                     \
    if (RV_PACKETS_TIMELINE && p.timeline) {                            \
        const uint32_t ph_now = uniform(static_cast<uint32_t>(__builtin_readcyclecounter())); \
        ph[k] += ph_now - ph_last;                                      \
        ph_last = ph_now;                                               \
    }

    for (;;) {
        const uint64_t alive = ballot(has);
        const uint32_t n_alive = static_cast<uint32_t>(__builtin_popcountll(alive));
        // pixels left to claim?  (refills the wave's pool of claimed work indices: 128 at a time for launches of any size that matters;
        // a chunk that is not a multiple of 64 — tiny images, the end of a shard — ends in a camera round with idle lanes)
        bool pixels = pool.end != pool.next;
        if (!pixels && !pool.exhausted) pixels = next_chunk<true>(pool, p, lane, wave_id);
        RV_PHASE(0)
        bool camera_round = false;
        bool cull = false;            // this camera round's 64 work items are one 16 x 4 block of one tile and frame: the rectangles apply
        uint32_t bx = 0, by = 0;      // ... that block (wave-uniform)
        if (pixels && n_alive + parked < 64u) {
            // ---- camera round: park what is alive, then every lane starts the pixel pool.next + lane
            if (n_alive) {
                if (has) park<AA1>(queue, parked + prefix_rank(alive), L, leave);
                parked += n_alive;
                has = false;
            }
            // the block this round takes: pool.next counts in the CLAIM ORDER, which deals a frame's blocks from all over the frame (FrameParams::perm_*, round 6:
            // a claim of 512 work items then holds its share of sky and of model instead of being one or the other); wave-uniform integer arithmetic
            uint32_t first = pool.next;
            if (ORDER && p.perm_groups != 0u) {
                uint32_t f = 0, in_frame = first;
                if (p.n_work_frame != p.n_work) {
                    f = fast_div(first, p.div_work_frame);
                    in_frame = first - f * p.n_work_frame;
                }
                first = uniform(f * p.n_work_frame + ((claim_order_block(in_frame >> 6, p.perm_groups, p.perm_stride, p.perm_shift, p.div_perm_groups) << 6) | (in_frame & 63u)));
            }
            const bool in_chunk = pool.next + lane < pool.end;
            const uint32_t work = first + lane;
            pool.next += 64u;
            uint32_t frame_offset = 0, pixel = work;
            if (p.n_work_frame != p.n_work) {
                frame_offset = fast_div(work, p.div_work_frame);
                pixel = work - frame_offset * p.n_work_frame;
            }
            uint32_t gx, gy;
            const bool inside = decode_work(p, pixel, gx, gy);
            // 64 consecutive work items starting at a multiple of 64 = rows 4 k .. 4 k + 3 of one 16 x 16 tile of one frame (n_work_frame is a multiple
            // of 256); chunks start at multiples of 64 for every launch of a size that matters (rvpt_abi.hip: plan_work) — otherwise no culling this round
            cull = CULLS || (p.rects != nullptr && (uniform(work) & 63u) == 0u);
            bx = uniform(gx >> 4);
            by = uniform(gy >> 2);
            if (in_chunk && inside) {
                L.work = work;
                L.gx = gx;
                L.gy = gy;
                L.rng = wang_hash(gx + gy * p.width) + (p.frame + frame_offset);  // util.glsl:35-36
                L.sample = 0;
                L.sum = mk(0.0f, 0.0f, 0.0f);
                begin_sample(L, p);
                leave = 0xFFFFFFFFu;
                nsmp += 1;
                has = true;
            }
            pool.next = min(pool.next, pool.end);
            camera_round = true;
            RV_PHASE(1)
        } else {
            // ---- bounce round: the lanes without a path take parked ones (the last parked first)
            if (parked && n_alive < 64u) {
                const uint64_t empty = ~alive;
                const uint32_t take = min(parked, 64u - n_alive);
                const uint32_t rank = prefix_rank(empty);
                if (!has && rank < take) {
                    unpark<AA1>(queue, parked - 1u - rank, L, leave);
                    has = true;
                }
                parked -= take;
            }
            if (ballot(has) == 0) break;  // no path anywhere, no pixel left
            RV_PHASE(2)
        }

        float closest = kInf;
        uint32_t hit = 0xFFFFFFFFu;
        const uint64_t active = ballot(has);
        const uint32_t n_active = static_cast<uint32_t>(__builtin_popcountll(active));
        if (RV_PACKETS_TIMELINE && p.timeline) {
            n_cam += camera_round ? 1u : 0u;
            n_bounce += camera_round ? 0u : 1u;
            lane_rounds += n_active;
            if (!pixels && t_dry == 0) t_dry = wall_clock64();
            if (!pixels) n_after_dry += 1, lanes_after_dry += n_active;
            if (camera_round) t_last_cam = wall_clock64();
        }
        if (!CULLS && n_active > 0u && n_active <= RV_PACKETS_SPLIT_BELOW && parked == 0u && p.vis == nullptr) {
            // ---- split mode (the launch's tail: no pixels left, the last paths dying out; ONLY WITHOUT the bounce cull — a split round walks all n_tris / k triangles per
            // lane, 72 tests at 32 rays, where a culled round of the default scene walks ~10: round 6): the few rays are spread over the whole
            // wave, k = 64 / n lanes per ray, lane s of a group testing triangles s, s + k, ...; a lexicographic (t, index)
            // min-reduction over the group reproduces the sequential closest hit exactly (trace_brute_resident's split mode; the
            // owner table lives in the queue, which is empty here)
            const uint32_t k = 64u / n_active;
            const uint32_t rank = prefix_rank(active);
            if (has) queue[rank] = lane;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint32_t group = lane / k, slice = lane - group * k;
            const bool helper = group < n_active;
            const uint32_t owner = helper ? queue[group] : lane;
            const f3 o = mk(__shfl(L.o.x, owner, 64), __shfl(L.o.y, owner, 64), __shfl(L.o.z, owner, 64));
            const f3 d = mk(__shfl(L.d.x, owner, 64), __shfl(L.d.y, owner, 64), __shfl(L.d.z, owner, 64));
            float c = kInf;
            uint32_t h = 0xFFFFFFFFu;
            if (helper) {
#pragma unroll 2
                for (uint32_t i = slice; i < p.n_tris; i += k) {
                    const PrepTri t = unpack(src[4 * i + 0], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]);
                    test_triangle(t, o, d, i, c, h);
                }
            }
            for (uint32_t m = 1; m < k; m <<= 1) {  // tree reduction towards slice 0 of every group (k need not be a power of two)
                const float c2 = __shfl(c, lane + m, 64);
                const uint32_t h2 = __shfl(h, lane + m, 64);
                const bool take = (slice + m < k) & ((c2 < c) | ((c2 == c) & (h2 < h)));
                c = take ? c2 : c;
                h = take ? h2 : h;
            }
            closest = __shfl(c, rank * k, 64);
            hit = __shfl(h, rank * k, 64);
            n_split += 1;
            __builtin_amdgcn_wave_barrier();  // the table is read before anything is parked over it
        } else if (camera_round && cull) {
            // ---- the triangles whose rectangle holds this block, 64 at a time: lane i looks at rectangle base + i, the ballot is the candidate list
            for (uint32_t base = 0; base < p.n_tris; base += 64u) {
                const uint32_t idx = base + lane;
                bool candidate = false;
                if (idx < p.n_tris) {
                    const uint2 r = lds_rect[idx];
                    candidate = rect_holds(r.x, r.y, bx, by);
                }
                uint64_t todo = ballot(candidate);
                while (todo != 0) {  // ascending triangle index: the order of the sequential rule
                    const uint32_t j = base + static_cast<uint32_t>(__builtin_ctzll(todo));
                    todo &= todo - 1;
                    if (has) camera_test_one(src, cam_records, j, L.o, L.d, closest, hit);
                }
            }
            RV_PHASE(3)
        } else if (!camera_round && (CULLS || p.vis != nullptr)) {
            // ---- bounce round with the bounce cull: a ray that leaves triangle A on side s can only hit the triangles of row 2 A + s of the table (those
            // not wholly behind A's plane as seen from that side); the wave walks the UNION of its lanes' rows — a superset for every lane
            const uint32_t *row = p.vis + static_cast<size_t>(leave == 0xFFFFFFFFu ? 0u : leave) * p.vis_stride;
            // Round 6, the LEAF BOXES (rvpt_vis.h): of what the union leaves, a group of kLeafTris consecutive triangles is walked only if some lane's ray can come near
            // the group's box (conservative slab test; lanes without a provable segment vote for every box) — default scene: 29.6 -> ~10 triangles per round
            const bool boxes = CULLS || p.leaf_boxes != nullptr;
            typedef const __attribute__((address_space(4))) float *ConstFloats;
            const ConstFloats leaf_boxes_k = (ConstFloats)(reinterpret_cast<uintptr_t>(p.leaf_boxes));
            const LeafRay lr = leaf_ray(L.o, L.d);
            const bool vote_all = has && leave == 0xFFFFFFFFu;
            const bool own_row = has && leave != 0xFFFFFFFFu;
            constexpr uint32_t kPerWord = 32u / kLeafTris, kMask = (1u << kLeafTris) - 1u;
            constexpr uint32_t kBatchMask = static_cast<uint32_t>((1ull << (RV_BOX_BATCH * kLeafTris)) - 1ull);
            // What a bounce round WAITS for (the per-phase clocks of tools/packets_timeline.py: the culls took a quarter of a wave's time, more than the triangle tests they
            // leave): a row word per 32 triangles and a box per leaf, each a load the next step depended on.  So: FOUR words of the lane's row in one 16-byte load (rows
            // are 16-byte aligned, a multiple of four words apart, zero padded: vis_stride), the four wave-wide ORs back to back, and a word's boxes in ONE scalar request
            // (contiguous: 32 B x kPerWord) before any of them is tested.
            for (uint32_t w0 = 0; w0 < p.vis_words; w0 += 4u) {
                uint4 r = make_uint4(0u, 0u, 0u, 0u);
                if (own_row) r = *reinterpret_cast<const uint4 *>(row + w0);
                if (vote_all) r = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
                // (all four even where the scene has fewer words — 143 triangles are five: four and one —: four independent DPP chains overlap, three branches around
                // them measured 3 % slower)
                const uint32_t m[4] = {wave_or(r.x), wave_or(r.y), wave_or(r.z), wave_or(r.w)};
                RV_PHASE(4)
#pragma unroll
                for (uint32_t j = 0; j < 4u; ++j) {
                    const uint32_t w = w0 + j;
                    if (w >= p.vis_words) break;
                    uint32_t todo = m[j];
                    if (w + 1u == p.vis_words && (p.n_tris & 31u) != 0u) todo &= (1u << (p.n_tris & 31u)) - 1u;
                    if (todo == 0u) continue;  // (wave-uniform)
                    if (boxes) {
                        // (through the constant address space: a wave-uniform address there is a SCALAR load from the scalar cache, boxes in SGPRs; as an ordinary global
                        // pointer the compiler makes it a vector load behind the kernel's own stores)
#pragma unroll
                        for (uint32_t k0 = 0; k0 < kPerWord; k0 += RV_BOX_BATCH) {
                            if (((todo >> (kLeafTris * k0)) & kBatchMask) == 0u) continue;  // (wave-uniform)
                            const ConstFloats b = leaf_boxes_k + 8u * (kPerWord * w + k0);
                            float bb[8u * RV_BOX_BATCH];
#pragma unroll
                            for (uint32_t i = 0; i < 8u * RV_BOX_BATCH; ++i) bb[i] = ((i & 7u) < 6u) ? b[i] : 0.0f;
#pragma unroll
                            for (uint32_t k = k0; k < k0 + RV_BOX_BATCH; ++k) {
                                if (((todo >> (kLeafTris * k)) & kMask) == 0u) continue;  // (wave-uniform)
                                const uint32_t o8 = 8u * (k - k0);
                                const float4 b0 = make_float4(bb[o8 + 0u], bb[o8 + 1u], bb[o8 + 2u], bb[o8 + 3u]), b1 = make_float4(bb[o8 + 4u], bb[o8 + 5u], 0.0f, 0.0f);
                                const bool near = has && (vote_all || leaf_slab(lr, b0, b1));
                                if (ballot(near) == 0) todo &= ~(kMask << (kLeafTris * k));
                            }
                        }
                    }
                    if (RV_PACKETS_TIMELINE && p.timeline) n_listed += static_cast<uint32_t>(__builtin_popcount(todo));
                    RV_PHASE(4)
                    if (has) intersect_listed(src, 32u * w, todo, L.o, L.d, closest, hit);
                    RV_PHASE(5)
                }
            }
        } else if (!CULLS && has) {
            if (camera_round)
                intersect_run_camera(src, cam_records, 0u, p.n_tris, L.o, L.d, closest, hit);
            else if (RV_PACKETS_BOUNCE_EARLY)
                intersect_run_early(src, p.n_tris, L.o, L.d, closest, hit);
            else
                intersect_run<4>(src, 0u, p.n_tris, L.o, L.d, closest, hit);
        }
        RV_PHASE(7)
        if (has) {
            L.nseg += 1;
            f3 radiance = mk(0.0f, 0.0f, 0.0f);
            if (shade(L, p, shade_src, hit, closest, radiance, &leave)) {  // the path ended
                L.sum = L.sum + radiance;
                L.sample += 1;
                if (!AA1 && L.sample < p.aa) {  // the pixel's next sample: its camera ray joins the bounce rays (it has lost its block)
                    uint32_t frame_offset = 0, pixel = L.work;
                    if (p.n_work_frame != p.n_work) {
                        frame_offset = fast_div(L.work, p.div_work_frame);
                        pixel = L.work - frame_offset * p.n_work_frame;
                    }
                    decode_work(p, pixel, L.gx, L.gy);
                    begin_sample(L, p);
                    leave = 0xFFFFFFFFu;
                    nsmp += 1;
                } else {
                    if (CULLS) {  // (these instances only run with frames in flight: the sample leaves as 12 bytes, blend_accumulate finishes compute_pass.comp:162-166)
                        const float faa = static_cast<float>(p.aa);
                        const f3 sampled = (AA1 || p.aa == 1) ? L.sum : mk(L.sum.x / faa, L.sum.y / faa, L.sum.z / faa);  // finish_pixel's operations
                        p.sample_out[L.work] = SampleRGB{sampled.x, sampled.y, sampled.z};
                    } else {
                        finish_pixel(L, p);
                    }
                    has = false;
                }
            }
        }
        RV_PHASE(6)
    }
    if (RV_PACKETS_TIMELINE && p.timeline && lane == 0) {
        unsigned long long *w = p.timeline + 8ull * p.n_waves + 8ull * wave_id;  // (second half of the rows)
        for (int k = 0; k < 8; ++k) w[k] = ph[k];
        unsigned long long *t = p.timeline + 8ull * wave_id;
        t[0] = t_start;
        t[1] = t_dry;
        t[2] = wall_clock64();
        t[3] = n_cam | (static_cast<unsigned long long>(n_bounce) << 32);
        t[4] = n_split | (static_cast<unsigned long long>(lane_rounds) << 32);
        t[5] = n_listed;
        t[6] = n_after_dry | (static_cast<unsigned long long>(lanes_after_dry) << 32);
        t[7] = t_last_cam;
    }
    wave_exit(p, lane, L.nseg, nsmp);
}

__global__ __launch_bounds__(kBlock, RV_PACKETS_MIN_WAVES) void trace_brute_packets(const FrameParams p) { packets_body<false, false>(p); }
__global__ __launch_bounds__(kBlock, RV_PACKETS_MIN_WAVES) void trace_brute_packets_aa1(const FrameParams p) { packets_body<true, false>(p); }
__global__ __launch_bounds__(kBlock, RV_PACKETS_MIN_WAVES) void trace_brute_packets_culls(const FrameParams p) { packets_body<false, true, false>(p); }
__global__ __launch_bounds__(kBlock, RV_PACKETS_MIN_WAVES) void trace_brute_packets_aa1_culls(const FrameParams p) { packets_body<true, true, false>(p); }
__global__ __launch_bounds__(kBlock, RV_PACKETS_MIN_WAVES) void trace_brute_packets_culls_order(const FrameParams p) { packets_body<false, true, true>(p); }
__global__ __launch_bounds__(kBlock, RV_PACKETS_MIN_WAVES) void trace_brute_packets_aa1_culls_order(const FrameParams p) { packets_body<true, true, true>(p); }

__global__ void bounce_visibility(const float4 *__restrict__ prep, uint32_t n, double margin, uint32_t words, uint32_t stride, uint32_t *__restrict__ out)
{
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 2u * n * stride) return;
    const uint32_t row = id / stride, w = id - row * stride;  // rows are `stride` >= `words` words apart (a multiple of four: the frame kernel loads four words at once), zero padded
    out[id] = w >= words ? 0u : bounce_row_word(reinterpret_cast<const float *>(prep), n, row, w, margin);  // rvpt_vis.h: the host evaluates the same function (rvpt_bounce_rows)
}

__global__ void camera_rects(const FrameParams p, uint2 *__restrict__ rects, float4 *__restrict__ records)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n_tris) return;
    const RectCamera c = rect_camera(p.cam, p.aspect, p.cam_w, p.width, p.height);
    const float4 q0 = p.prep[4 * i + 0], q1 = p.prep[4 * i + 1], q2 = p.prep[4 * i + 2];
    const float a0[4] = {q0.x, q0.y, q0.z, q0.w}, a1[4] = {q1.x, q1.y, q1.z, q1.w}, a2[4] = {q2.x, q2.y, q2.z, q2.w};
    bool neg;
    const float a = camera_numerator(mk(q0.x, q0.y, q0.z), mk(q0.w, q1.x, q1.y), mk(p.cam[9], p.cam[10], p.cam[11]), neg);
    uint2 r;
    camera_rect(c, a0, a1, a2, a, r.x, r.y);
    rects[i] = r;
    if (records != nullptr) {  // ... and the triangle's camera record for the same camera (rvpt_early_out.h), read by the camera rounds through scalar loads
        v4f qa, qb;
        qa.x = q0.x, qa.y = q0.y, qa.z = q0.z, qa.w = q0.w;
        qb.x = q1.x, qb.y = q1.y, qb.z = q1.z, qb.w = q1.w;
        const v4f rec = camera_record(qa, qb, mk(p.cam[9], p.cam[10], p.cam[11]));
        records[i] = make_float4(rec.x, rec.y, rec.z, rec.w);
    }
}

#if RVPT_HIP_LAB  // ---- diagnostics of the two culls and of the pre-test (include/rvpt_hip_lab.h)
__global__ void selftest_camera_rects(const FrameParams p, const uint2 *__restrict__ rects, uint32_t n_samples, unsigned long long *__restrict__ out)
{
    unsigned long long accepted = 0, outside = 0, held = 0, pairs = 0;
    const uint32_t n_px = p.width * p.height;
    for (uint32_t px = blockIdx.x * blockDim.x + threadIdx.x; px < n_px; px += gridDim.x * blockDim.x) {
        Lane L{};
        L.gx = px % p.width;
        L.gy = px / p.width;
        const uint32_t bx = L.gx >> 4, by = L.gy >> 2;
        for (uint32_t s = 0; s < n_samples; ++s) {
            L.rng = wang_hash(px) + (p.frame + s);  // the sample the frame kernels trace for frame p.frame + s (util.glsl:35-36)
            begin_sample(L, p);
            for (uint32_t j = 0; j < p.n_tris; ++j) {
                const float4 q0 = p.prep[4 * j + 0], q1 = p.prep[4 * j + 1], q2 = p.prep[4 * j + 2], q3 = p.prep[4 * j + 3];
                v4f a, b, c, d;
                a.x = q0.x, a.y = q0.y, a.z = q0.z, a.w = q0.w;
                b.x = q1.x, b.y = q1.y, b.z = q1.z, b.w = q1.w;
                c.x = q2.x, c.y = q2.y, c.z = q2.z, c.w = q2.w;
                d.x = q3.x, d.y = q3.y, d.z = q3.z, d.w = q3.w;
                const OpenTest r = test_triangle_open(unpack(a, b, c, d), L.o, L.d);
                const bool accept = (r.m > 0.0f) & (r.s < 1.0f) & (r.tt < kInf);  // accept_hit with the interval wide open
                const uint2 rc = rects[j];
                const bool holds = rect_holds(rc.x, rc.y, bx, by);
                accepted += accept ? 1u : 0u;
                outside += (accept && !holds) ? 1u : 0u;
                if (s == 0 && (L.gx & 15u) == 0u && (L.gy & 3u) == 0u) {  // once per block
                    held += holds ? 1u : 0u;
                    pairs += 1u;
                }
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        accepted += __shfl_down(accepted, off, 64);
        outside += __shfl_down(outside, off, 64);
        held += __shfl_down(held, off, 64);
        pairs += __shfl_down(pairs, off, 64);
    }
    if (lane_id() == 0) {
        atomicAdd(&out[0], accepted);
        atomicAdd(&out[1], outside);
        atomicAdd(&out[2], held);
        atomicAdd(&out[3], pairs);
    }
}

__global__ void selftest_bounce_cull(const FrameParams p, uint32_t n_samples, unsigned long long *__restrict__ out)
{
    unsigned long long accepted = 0, outside = 0, outside_box = 0, box_tests = 0, box_hits = 0;
    const ShadeSrc shade_src{p.prep, p.mat_index, p.mats, p.unit_n};
    const uint32_t n_px = p.width * p.height;
    for (uint32_t px = blockIdx.x * blockDim.x + threadIdx.x; px < n_px; px += gridDim.x * blockDim.x) {
        Lane L{};
        L.gx = px % p.width;
        L.gy = px / p.width;
        for (uint32_t s = 0; s < n_samples; ++s) {
            L.rng = wang_hash(px) + (p.frame + s);
            begin_sample(L, p);
            uint32_t leave = 0xFFFFFFFFu;
            for (;;) {  // one path, every segment against EVERY triangle (the mixed-packet kernel's loop without its tricks)
                float closest = kInf;
                uint32_t hit = 0xFFFFFFFFu;
                const LeafRay lr = leaf_ray(L.o, L.d);
                for (uint32_t j = 0; j < p.n_tris; ++j) {
                    const float4 q0 = p.prep[4 * j + 0], q1 = p.prep[4 * j + 1], q2 = p.prep[4 * j + 2], q3 = p.prep[4 * j + 3];
                    v4f a, b, c, d;
                    a.x = q0.x, a.y = q0.y, a.z = q0.z, a.w = q0.w;
                    b.x = q1.x, b.y = q1.y, b.z = q1.z, b.w = q1.w;
                    c.x = q2.x, c.y = q2.y, c.z = q2.z, c.w = q2.w;
                    d.x = q3.x, d.y = q3.y, d.z = q3.z, d.w = q3.w;
                    const OpenTest r = test_triangle_open(unpack(a, b, c, d), L.o, L.d);
                    const bool open_accept = (r.m > 0.0f) & (r.s < 1.0f) & (r.tt < kInf);  // with the interval wide open: every t > 0 the test can accept
                    if (leave != 0xFFFFFFFFu && open_accept) {
                        accepted += 1;
                        const uint32_t word = p.vis[static_cast<size_t>(leave) * p.vis_stride + (j >> 5)];
                        outside += ((word >> (j & 31u)) & 1u) ? 0u : 1u;
                        if (p.leaf_boxes != nullptr) {  // ... and the ray passes the slab test of the triangle's leaf box, as the frame kernel evaluates it
                            const uint32_t leaf = j / kLeafTris;
                            outside_box += leaf_slab(lr, p.leaf_boxes[2u * leaf + 0u], p.leaf_boxes[2u * leaf + 1u]) ? 0u : 1u;
                        }
                    }
                    if (leave != 0xFFFFFFFFu && p.leaf_boxes != nullptr && (j % kLeafTris) == 0u) {  // how selective the boxes are, per ray
                        box_tests += 1;
                        box_hits += leaf_slab(lr, p.leaf_boxes[2u * (j / kLeafTris) + 0u], p.leaf_boxes[2u * (j / kLeafTris) + 1u]) ? 1u : 0u;
                    }
                    const bool accept = (r.m > 0.0f) & (r.s < 1.0f) & (r.tt < closest);
                    closest = accept ? r.tt : closest;
                    hit = accept ? j : hit;
                }
                f3 radiance = mk(0.0f, 0.0f, 0.0f);
                if (shade(L, p, shade_src, hit, closest, radiance, &leave)) break;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        accepted += __shfl_down(accepted, off, 64);
        outside += __shfl_down(outside, off, 64);
        outside_box += __shfl_down(outside_box, off, 64);
        box_tests += __shfl_down(box_tests, off, 64);
        box_hits += __shfl_down(box_hits, off, 64);
    }
    if (lane_id() == 0) {
        atomicAdd(&out[0], accepted);
        atomicAdd(&out[1], outside);
        atomicAdd(&out[2], outside_box);
        atomicAdd(&out[3], box_tests);
        atomicAdd(&out[4], box_hits);
    }
}

__global__ void selftest_camera_pretest(const float *__restrict__ a, const float *__restrict__ den, const float *__restrict__ closest,
                                        unsigned char *__restrict__ out, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // the record of a triangle whose plane has unit normal x and lies at distance a[i] >= 0 in front of the origin: (n', |num|) = ((1, 0, 0), a) or NaN
    v4f q0, q1;
    q0.x = a[i], q0.y = 0.0f, q0.z = 0.0f, q0.w = 1.0f;
    q1.x = 0.0f, q1.y = 0.0f, q1.z = 0.0f, q1.w = 0.0f;
    const v4f rec = camera_record(q0, q1, mk(0.0f, 0.0f, 0.0f));
    const bool through = camera_pretest(rec.w, den[i], closest[i]);
    const float t = div_dots(a[i], den[i]);
    const bool quotient = (t > 0.0f) & (t < closest[i]);
    out[i] = static_cast<unsigned char>((through ? 1u : 0u) | (quotient ? 2u : 0u));
}

#endif  // RVPT_HIP_LAB

}  // namespace rv

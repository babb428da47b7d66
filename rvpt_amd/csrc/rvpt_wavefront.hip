// rvpt_wavefront.hip — the wavefront form of the BVH path (HBM-resident scenes, Kajiya / pinhole) for gfx950.
//
// What it computes is what trace_bvh<.., RESIDENT = false, GENERIC = false, ..> computes (rvpt_kernels.hip): compute_pass.comp::main
// over intersect_bvh (intersection.glsl:361-413, 489-517) and integrator_Kajiya (integrators.glsl:547-677), bit for bit.  How:
// the megakernel keeps a path's whole state in registers (76 VGPRs: six waves per SIMD) and lets the lanes of a packet alternate
// between walking the tree and shading, at 42 % lane utilisation.  Here the two halves are separate kernels with separate register
// budgets, and the path state lives in HBM between them (288 GB: a launch of 8 x 1080p frames keeps 16.6 M paths in flight, 1 GB):
//
//   wf_begin      one thread per work item (frame, pixel): RNG seed, camera ray (compute_pass.comp:50-54, 151-156)   -> records
//   per iteration (at most aa * max_bounces of them, every live path advancing one segment per iteration):
//     wf_traverse persistent waves; a lane holds a ray, its interval and its stack — nothing else (<= 64 VGPRs: eight waves per
//                 SIMD) — and is refilled from the record stream the moment its walk ends; writes (t, hit) into the record
//     wf_shade    one thread per live record, 64 hits at a time at full lane utilisation: one loop body of integrator_Kajiya,
//                 the next sample's camera ray when a path ends, the pixel's sample mean when its last sample ends; the
//                 survivors are compacted to the front of their 256-record chunk (ballot + mbcnt + one LDS exchange)
//
// Records never leave their chunk (chunk c = work items 256 c .. 256 c + 255 = one 16 x 16 tile of one frame), so compaction needs
// no global atomics, a record's origin is eight bits, and everything a chunk touches stays within a few KiB.  The samples of a
// pixel are sequential (the reference's RNG state runs on from one sample into the next, util.glsl:35-50), the paths of different
// work items independent: the order of operations per pixel is the megakernel's, hence the same bits.
// Per segment the pipeline moves 32 B (ray read) + 8 B (hit write) + 64 B + 64 B (record read / write in wf_shade) through HBM —
// the first kernels of this renderer for which the HBM roofline means something.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rvpt_device.h"
#include "rvpt_early_out.h"
#include "rvpt_wavefront.h"

#ifndef RV_WF_MIN_WAVES
#define RV_WF_MIN_WAVES 8  // wf_traverse: 64 VGPRs
#endif
#ifndef RV_WF_CLAIM_RAYS
#define RV_WF_CLAIM_RAYS 512u  // a dynamic claim aims at this many live rays
#endif

namespace rv {

namespace {

// The path records stream through the memory system once per iteration (GBs per launch) while the tree and the triangles — a MB or
// so, fetched again and again by dependent loads — should stay in L2: every record access is NON-TEMPORAL (the nt cache policy:
// streamed lines are the first to be evicted).  Measured without it: the traverse kernel's L2 hit rate fell from the megakernel's
// 99.7 % to 84 % and 59 % of its wave time went to waiting on memory (profiles/r03_c3_wf_first_pmc.json).
__device__ __forceinline__ float4 stream_load(const float4 *p)
{
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stream_store(float4 *p, const float4 x)
{
    v4f v;
    v.x = x.x, v.y = x.y, v.z = x.z, v.w = x.w;
    __builtin_nontemporal_store(v, reinterpret_cast<v4f *>(p));
}
__device__ __forceinline__ void store_ray(float4 *rays, const uint32_t q, const f3 o, const f3 d)
{
    stream_store(rays + 2 * q + 0, make_float4(o.x, o.y, o.z, d.x));
    stream_store(rays + 2 * q + 1, make_float4(d.y, d.z, 0.0f, 0.0f));
}
__device__ __forceinline__ void store_aux(float4 *aux, const uint32_t q, const Lane &L, const uint32_t slot)
{
    stream_store(aux + 2 * q + 0, make_float4(L.thr.x, L.thr.y, L.thr.z, L.col.x));
    stream_store(aux + 2 * q + 1, make_float4(L.col.y, L.col.z, __uint_as_float(L.rng),
                                              __uint_as_float(slot | (static_cast<uint32_t>(L.bounce) << 8) | (static_cast<uint32_t>(L.sample) << 16))));
}

// work item -> (frame offset, pixel of this rank's tile-linear accumulator)
__device__ __forceinline__ void split_work(const FrameParams &p, const uint32_t work, uint32_t &frame_offset, uint32_t &pixel)
{
    frame_offset = 0;
    pixel = work;
    if (p.n_work_frame != p.n_work) {
        frame_offset = work / p.n_work_frame;
        pixel = work - frame_offset * p.n_work_frame;
    }
}

// Compaction of the surviving records of one chunk (one work-group): position of this thread's record among the survivors, and
// their number.  `wave_counts` is 4 words of LDS.  Contains the barrier that separates a chunk's reads from its in-place writes.
__device__ __forceinline__ uint32_t compact_rank(const bool alive, uint32_t *wave_counts, uint32_t &total)
{
    const uint64_t mask = ballot(alive);
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0) wave_counts[wave] = static_cast<uint32_t>(__builtin_popcountll(mask));
    __syncthreads();
    uint32_t base = 0;
    total = 0;
#pragma unroll
    for (uint32_t w = 0; w < kWfChunk / 64u; ++w) {
        const uint32_t n = wave_counts[w];
        base += (w < wave) ? n : 0u;
        total += n;
    }
    return base + prefix_rank(mask);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// Every work item of the launch starts its first sample: RNG seed util.glsl:35-36, camera ray compute_pass.comp:151-156.
// Work items of partial edge tiles that lie outside the image get no record.
__global__ __launch_bounds__(kWfChunk) void wf_begin(const FrameParams p)
{
    __shared__ uint32_t wave_counts[kWfChunk / 64u];
    uint32_t live = 0;
    for (uint32_t c = blockIdx.x; c < p.wf_chunks; c += gridDim.x) {
        const uint32_t work = c * kWfChunk + threadIdx.x;
        uint32_t frame_offset, pixel, gx, gy;
        split_work(p, work, frame_offset, pixel);
        Lane L{};
        const bool alive = decode_work(p, pixel, gx, gy);
        if (alive) {
            L.gx = gx;
            L.gy = gy;
            L.rng = wang_hash(gx + gy * p.width) + (p.frame + frame_offset);
            L.sample = 0;
            begin_sample(L, p);
        }
        uint32_t total;
        const uint32_t at = c * kWfChunk + compact_rank(alive, wave_counts, total);
        if (alive) {
            store_ray(p.wf_rays, at, L.o, L.d);
            store_aux(p.wf_aux, at, L, threadIdx.x);
        }
        if (threadIdx.x == 0) {
            p.wf_count[c] = total;
            live += total;
        }
        __syncthreads();  // wave_counts is reused by the next chunk
    }
    if (threadIdx.x == 0) {
        if (live) atomicAdd(&p.wf_live[0], live);
        if (p.stats != nullptr && live) atomicAdd(&p.stats[1], static_cast<unsigned long long>(live));  // samples begun
    }
}

// ------------------------------------------------------------------------------------------------
// One integrator step per live record (integrators.glsl:576-671), then compute_pass.comp:156-166 for a path that ended.
__global__ __launch_bounds__(kWfChunk) void wf_shade(const FrameParams p)
{
    __shared__ uint32_t wave_counts[kWfChunk / 64u];
    const ShadeSrc shade_src{p.prep, p.mat_index, p.mats};
    uint32_t live = 0, segments = 0, begun = 0;
    for (uint32_t c = blockIdx.x; c < p.wf_chunks; c += gridDim.x) {
        const uint32_t n = p.wf_count[c];
        if (n == 0) continue;  // (uniform over the work-group)
        const uint32_t q = c * kWfChunk + threadIdx.x;
        bool alive = threadIdx.x < n;
        Lane L{};
        uint32_t slot = 0;
        if (alive) {
            const float4 r0 = stream_load(p.wf_rays + 2 * q + 0), r1 = stream_load(p.wf_rays + 2 * q + 1);
            const float4 a0 = stream_load(p.wf_aux + 2 * q + 0), a1 = stream_load(p.wf_aux + 2 * q + 1);
            const v2f th = __builtin_nontemporal_load(reinterpret_cast<const v2f *>(p.wf_hits) + q);  // (t, hit index) of this record's ray
            L.o = mk(r0.x, r0.y, r0.z);
            L.d = mk(r0.w, r1.x, r1.y);
            L.thr = mk(a0.x, a0.y, a0.z);
            L.col = mk(a0.w, a1.x, a1.y);
            L.rng = __float_as_uint(a1.z);
            const uint32_t packed = __float_as_uint(a1.w);
            slot = packed & 255u;
            L.bounce = static_cast<int>((packed >> 8) & 255u);
            L.sample = static_cast<int>(packed >> 16);
            f3 radiance = mk(0.0f, 0.0f, 0.0f);
            if (shade(L, p, shade_src, __float_as_uint(th.y), th.x, radiance)) {  // the path ended
                L.work = c * kWfChunk + slot;
                L.sum = mk(0.0f, 0.0f, 0.0f);
                if (L.sample > 0) {
                    const float4 s = p.wf_sum[L.work];
                    L.sum = mk(s.x, s.y, s.z);
                }
                L.sum = L.sum + radiance;
                L.sample += 1;
                if (L.sample < p.aa) {  // the pixel's next sample continues the RNG stream (util.glsl:38-50)
                    p.wf_sum[L.work] = make_float4(L.sum.x, L.sum.y, L.sum.z, 0.0f);
                    uint32_t frame_offset, pixel;
                    split_work(p, L.work, frame_offset, pixel);
                    decode_work(p, pixel, L.gx, L.gy);
                    begin_sample(L, p);
                } else {
                    finish_pixel(L, p);
                    alive = false;
                }
            }
        }
        if (p.stats != nullptr) {
            segments += static_cast<uint32_t>(__builtin_popcountll(ballot(threadIdx.x < n)));
            begun += static_cast<uint32_t>(__builtin_popcountll(ballot(alive && L.bounce == 0)));
        }
        uint32_t total;
        const uint32_t at = c * kWfChunk + compact_rank(alive, wave_counts, total);  // (barrier: every record of the chunk has been read)
        if (alive) {
            store_ray(p.wf_rays, at, L.o, L.d);
            store_aux(p.wf_aux, at, L, slot);
        }
        if (threadIdx.x == 0) {
            p.wf_count[c] = total;
            live += total;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && live) atomicAdd(&p.wf_live[p.wf_iteration + 1], live);
    if (p.stats != nullptr && (threadIdx.x & 63u) == 0) {
        if (segments) atomicAdd(&p.stats[0], static_cast<unsigned long long>(segments));
        if (begun) atomicAdd(&p.stats[1], static_cast<unsigned long long>(begun));
    }
}

// ------------------------------------------------------------------------------------------------
// Closest hit of every live ray: intersect_bvh (intersection.glsl:361-413) exactly as trace_bvh walks it (rvpt_kernels.hip: both
// children tested at their parent, two-word stack slots with the exact entry distance and the packed head, leaves parked and run in
// batches, first stack levels in LDS, the top of the tree in LDS), with nothing but the walk in the wave.
namespace {

// The record stream of one wave: ranges of chunks — a static first range, then claims from sharded counters — and inside a
// range one open chunk whose live records are handed out front to back.
struct ChunkPool {
    uint32_t range_begin = 0, c_next = 0, c_end = 0;  // claimed chunks [c_next, c_end) not opened yet
    uint32_t counts = 0;                              // per lane: live records of chunk range_begin + lane
    uint32_t base = 0, pos = 0, cnt = 0;              // open chunk: first record index, records handed out, live records
    uint32_t shard = 0, shards_dry = 0;
    bool first = true, exhausted = false;
};

__device__ __forceinline__ bool pool_dry(const ChunkPool &pool) { return pool.exhausted && pool.c_next == pool.c_end && pool.pos == pool.cnt; }

// opens the next non-empty chunk; false when the stream has ended.  Uniform control flow (every lane of the wave calls it).
__device__ __forceinline__ bool open_next_chunk(ChunkPool &pool, const FrameParams &p, unsigned long long *counter, const uint32_t claim_chunks,
                                                const uint32_t lane, const uint32_t wave_id)
{
    for (;;) {
        while (pool.c_next < pool.c_end) {
            const uint32_t c = pool.c_next++;
            const uint32_t n = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(pool.counts), static_cast<int>(c - pool.range_begin)));
            if (n) {
                pool.base = c * kWfChunk;
                pool.pos = 0;
                pool.cnt = n;
                return true;
            }
        }
        if (pool.exhausted) return false;
        uint32_t c0, n;
        if (pool.first) {  // static first range: no atomic, no thundering herd at kernel start
            pool.first = false;
            c0 = wave_id * p.first_units;
            n = p.first_units;
            if (c0 >= p.dyn_base) n = 0;
            n = min(n, p.dyn_base - min(c0, p.dyn_base));
        } else {
            for (;;) {
                uint32_t pos = 0;
                if (lane == 0) pos = static_cast<uint32_t>(atomicAdd(&counter[kShardStride * pool.shard], static_cast<unsigned long long>(claim_chunks)));
                pos = uniform(pos);
                const uint32_t shard_begin = p.dyn_base + pool.shard * p.shard_len;
                const uint32_t shard_end = min(p.wf_chunks, shard_begin + p.shard_len);
                c0 = shard_begin + pos;
                if (pos < p.shard_len && c0 < shard_end) {
                    n = min(claim_chunks, shard_end - c0);
                    break;
                }
                pool.shard = (pool.shard + 1u) % kClaimShards;
                if (++pool.shards_dry >= kClaimShards) {
                    pool.exhausted = true;
                    return false;
                }
            }
        }
        pool.range_begin = c0;
        pool.c_next = c0;
        pool.c_end = c0 + n;
        pool.counts = (lane < n) ? p.wf_count[c0 + lane] : 0u;  // one coalesced load for the whole range (n <= 64)
    }
}

}  // namespace

template <bool ORDERED>
__global__ __launch_bounds__(kBlock, RV_WF_MIN_WAVES) void wf_traverse(const FrameParams p)
{
    const uint32_t live = p.wf_live[p.wf_iteration];
    if (live == 0) return;  // every path of the launch has ended
    // LDS: [stack: stack_lds_levels x 2 words x kBlock] [top of the tree: bvh_top_nodes x 32 B]
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    float4 *lds_top = reinterpret_cast<float4 *>(lds_stack + 2u * p.stack_lds_levels * kBlock);
    const uint32_t top_nodes = p.bvh_top_nodes;
    for (uint32_t i = threadIdx.x; i < 2u * top_nodes; i += kBlock) lds_top[i] = p.nodes[i];
    __syncthreads();
    const float4 *nodes = p.nodes;
    const v4f *prep = reinterpret_cast<const v4f *>(p.prep);
    float4 *rays = p.wf_rays;
    const uint32_t top_level = p.stack_levels - 1u;
    const uint32_t head_shift = p.head_shift;
    const uint32_t lds_levels = p.stack_lds_levels;
    uint32_t *const ovf = p.stack_overflow + (static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x);
    const size_t ovf_stride = static_cast<size_t>(gridDim.x) * kBlock;

    const uint32_t lane = lane_id();
    const uint32_t wave_id = uniform(blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6));
    unsigned long long *counter = p.counter;  // this iteration's claim counters (zeroed once per dispatch)
    // a claim aims at RV_WF_CLAIM_RAYS live rays: few chunks while the launch is dense, many in its sparse tail
    const uint32_t claim_chunks = static_cast<uint32_t>(min(64ull, max(1ull, (static_cast<unsigned long long>(RV_WF_CLAIM_RAYS) * p.wf_chunks + live - 1ull) / live)));
    ChunkPool pool;
    pool.shard = wave_id % kClaimShards;

    bool active = false;  // this lane walks the tree for the ray of record `qpos`
    bool finished = false;  // ... its walk is over and (closest, hit) wait to be written: the store happens at the next refill, not in the walk loop
                            // (a store in the loop makes every following step wait for its acknowledgement: loads and stores share vmcnt)
    f3 o = mk(0.0f, 0.0f, 0.0f), d = o, inv = o;
    float closest = kInf;
    uint32_t hit = 0xFFFFFFFFu, sp = 0, qpos = 0;
    uint32_t first = 0, leaf_count = 0;  // current node's (first_child_or_primitive, primitive_count): an inner node's children pair, or a parked leaf
#ifdef RV_BVH_PROFILE  // experiments only (tools/wf_phase_profile.py): where a traverse wave's time goes, summed over the launches of a sequence
    unsigned long long pf_refill = 0, pf_inner = 0, pf_leaf = 0, pf_iters = 0, pf_leaf_phases = 0, pf_inner_lanes = 0, pf_leaf_lanes = 0,
                       pf_refills = 0, pf_refill_lanes = 0, pf_dry_iters = 0, pf_t0 = __builtin_amdgcn_s_memtime(), pf_mark = 0;
#endif
    auto put_result = [&]() {  // 8 bytes per ray into the chunk's 2 KiB of the hit array (its own array: partial-line writes of neighbours merge in L2)
        v2f r;
        r.x = closest, r.y = __uint_as_float(hit);
        __builtin_nontemporal_store(r, reinterpret_cast<v2f *>(p.wf_hits) + qpos);
    };

    for (;;) {
#ifdef RV_BVH_PROFILE
        pf_mark = __builtin_amdgcn_s_memtime();
        pf_refills += 1;
        pf_refill_lanes += __builtin_popcountll(ballot(!active));
#endif
        if (finished) {
            put_result();
            finished = false;
        }
        // ---- refill: every idle lane takes the next record of the stream and starts its walk at the root
        for (;;) {
            const uint64_t mask = ballot(!active);
            if (mask == 0) break;
            if (pool.pos == pool.cnt && !open_next_chunk(pool, p, counter, claim_chunks, lane, wave_id)) break;
            const uint32_t avail = pool.cnt - pool.pos;
            const uint32_t rank = prefix_rank(mask);
            if (!active && rank < avail) {
                qpos = pool.base + pool.pos + rank;
#ifdef RV_WF_EXP_FOOTPRINT  // experiment (wrong images): every record access of the traverse kernel lands in the first 64 Ki records
                qpos &= 0xFFFFu;
#endif
                const v4f r0 = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(rays + 2 * qpos));
                const v2f r1 = __builtin_nontemporal_load(reinterpret_cast<const v2f *>(rays + 2 * qpos + 1));
                o = mk(r0.x, r0.y, r0.z);
                d = mk(r0.w, r1.x, r1.y);
                closest = kInf;
                hit = 0xFFFFFFFFu;
                inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                sp = 0;
                const float4 n0 = lds_top[0], n1 = lds_top[1];
                float entry;
                if (slab_entry(o, inv, n0, n1, closest, entry)) {
                    first = __float_as_uint(n0.x);
                    leaf_count = __float_as_uint(n0.y);
                    active = true;
                } else {
                    put_result();  // the ray misses the root box
                }
            }
            pool.pos += min(static_cast<uint32_t>(__builtin_popcountll(mask)), avail);
        }
#ifdef RV_BVH_PROFILE
        pf_refill += __builtin_amdgcn_s_memtime() - pf_mark;
#endif
        if (ballot(active) == 0) break;
        const bool dry = pool_dry(pool);

        // ---- walk: every iteration each walking lane handles one node; a lane that reaches a leaf parks there until enough
        // lanes have one (or nobody walks), then they run their triangle tests together
        for (uint32_t steps = 0;; ++steps) {
#ifdef RV_BVH_PROFILE
            pf_mark = __builtin_amdgcn_s_memtime();
            pf_iters += 1;
            pf_inner_lanes += __builtin_popcountll(ballot(active && leaf_count == 0));
            if (dry) pf_dry_iters += 1;
#endif
            bool need_pop = false;
            if (active && leaf_count == 0) {
                const uint32_t c = first;  // sibling pair = one 64-byte line in the device layout
                const float4 *pair = (c + 1u < top_nodes) ? lds_top + 2 * c : nodes + 2 * c;  // one flat address per lane
                const float4 a0 = pair[0], a1 = pair[1], b0 = pair[2], b1 = pair[3];
                float e0, e1;
                const bool h0 = slab_entry(o, inv, a0, a1, closest, e0);
                const bool h1 = slab_entry(o, inv, b0, b1, closest, e1);
                const bool right_first = h1 && (!h0 || (ORDERED && e1 < e0));
                if (h0 && h1) {
                    const uint32_t far_entry = __float_as_uint(right_first ? e0 : e1);
                    const float4 far_head = right_first ? a0 : b0;
                    const uint32_t far_node = head_shift ? (__float_as_uint(far_head.x) | (__float_as_uint(far_head.y) << head_shift)) : (right_first ? c : c + 1u);
                    const uint32_t at = min(sp, top_level);
                    if (at < lds_levels) {
                        lds_stack[(2u * at + 0u) * kBlock + threadIdx.x] = far_entry;
                        lds_stack[(2u * at + 1u) * kBlock + threadIdx.x] = far_node;
                    } else {
                        ovf[(2u * (at - lds_levels) + 0u) * ovf_stride] = far_entry;
                        ovf[(2u * (at - lds_levels) + 1u) * ovf_stride] = far_node;
                    }
                    sp += 1;
                }
                if (h0 || h1) {
                    const float4 near_head = right_first ? b0 : a0;
                    first = __float_as_uint(near_head.x);
                    leaf_count = __float_as_uint(near_head.y);
                } else {
                    need_pop = true;
                }
            }
#ifdef RV_BVH_PROFILE
            pf_inner += __builtin_amdgcn_s_memtime() - pf_mark;
            pf_mark = __builtin_amdgcn_s_memtime();
#endif
            const uint32_t at_leaf = static_cast<uint32_t>(__builtin_popcountll(ballot(leaf_count > 0)));
            const uint32_t at_inner = static_cast<uint32_t>(__builtin_popcountll(ballot(active && leaf_count == 0)));
            const bool run_leaves = at_leaf > 0 && (at_inner == 0 || at_leaf >= p.bvh_leaf_batch);
#ifdef RV_BVH_PROFILE
            if (run_leaves) {
                pf_leaf_phases += 1;
                pf_leaf_lanes += at_leaf;
            }
#endif
            if (run_leaves && leaf_count > 0) {
                for (uint32_t i = first; i < first + leaf_count; ++i) {
                    const v4f *tp = prep + 4 * i;
                    const PrepTri t = unpack(tp[0], tp[1], tp[2], tp[3]);
                    test_triangle(t, o, d, i, closest, hit);
                }
                leaf_count = 0;
                need_pop = true;
            }
#ifdef RV_BVH_PROFILE
            pf_leaf += __builtin_amdgcn_s_memtime() - pf_mark;
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
                    if (closest >= __uint_as_float(entry_bits)) {  // the reference's box test at pop time (see trace_bvh)
                        if (head_shift) {
                            first = cand & ((1u << head_shift) - 1u);
                            leaf_count = cand >> head_shift;
                        } else {
                            const float2 fc = *reinterpret_cast<const float2 *>((cand < top_nodes) ? lds_top + 2 * cand : nodes + 2 * cand);
                            first = __float_as_uint(fc.x);
                            leaf_count = __float_as_uint(fc.y);
                        }
                        found = true;
                    }
                }
                if (!found) {  // nothing left to visit: the walk is over
                    finished = true;
                    active = false;
                    leaf_count = 0;
                }
            }
#ifdef RV_BVH_PROFILE
            pf_inner += __builtin_amdgcn_s_memtime() - pf_mark;  // pops count as inner-node work
#endif
            if (ballot(active) == 0) break;
            if (!dry) {
                const uint32_t idle = 64u - static_cast<uint32_t>(__builtin_popcountll(ballot(active)));
                if (idle >= p.bvh_refill || (idle > 0 && steps >= 4u * p.bvh_refill)) break;
            }
        }
    }
#ifdef RV_BVH_PROFILE
    if (p.timeline && lane == 0) {  // (the launches of a sequence add into the same rows)
        unsigned long long *t = p.timeline + 8ull * wave_id;
        atomicAdd(&t[0], pf_refill), atomicAdd(&t[1], pf_inner), atomicAdd(&t[2], pf_leaf), atomicAdd(&t[3], pf_iters | (pf_leaf_phases << 32)),
            atomicAdd(&t[4], pf_inner_lanes | (pf_leaf_lanes << 32)), atomicAdd(&t[5], pf_dry_iters), atomicAdd(&t[6], pf_refill_lanes | (pf_refills << 32)),
            atomicAdd(&t[7], __builtin_amdgcn_s_memtime() - pf_t0);
    }
#endif
}

template __global__ void wf_traverse<false>(const FrameParams);
template __global__ void wf_traverse<true>(const FrameParams);

// ------------------------------------------------------------------------------------------------
// Brute force in wavefront form (scenes resident in LDS): closest hit of every live ray over all triangles in buffer order
// (intersection.glsl:267-323 per triangle, the strict accept rule of :311 => the first triangle on exact ties), 64 rays of one
// chunk per packet.  Every ray costs the same n_tris tests, so a packet's lanes finish together: no refill, no tail, full lanes.
// EARLY_OUT (the iteration that holds the camera rays): the records of a chunk are the pixels of one 16 x 16 tile, so a packet's
// rays share their origin and point the same way, and for most triangles NO ray of the packet can accept — accept implies
// 0 < t < closest (a NaN t fails `t < closest`; otherwise min3(t, u, v) > 0 gives t > 0), and t is the plane distance alone, 15 of
// the test's 38 VALU.  The loop computes t for four triangles, and finishes a test (barycentrics, 23 VALU) only if some lane of
// the packet passes 0 < t < closest with the interval as it stood before the group (closest only shrinks: a superset of what the
// sequential rule lets through).  Measured on the default scene: 61 % of all (packet, triangle) pairs skip the second half.
#ifndef RV_WF_BRUTE_MIN_WAVES
#define RV_WF_BRUTE_MIN_WAVES 6
#endif
template <bool EARLY_OUT>
__global__ __launch_bounds__(kBlock, RV_WF_BRUTE_MIN_WAVES) void wf_trace_brute(const FrameParams p)
{
    const uint32_t live = p.wf_live[p.wf_iteration];
    if (live == 0) return;
    extern __shared__ __attribute__((aligned(16))) float4 lds_tris[];
    for (uint32_t i = threadIdx.x; i < 4u * p.n_tris; i += kBlock) lds_tris[i] = p.prep[i];
    __syncthreads();
    const v4f *src = reinterpret_cast<const v4f *>(lds_tris);
    const uint32_t lane = lane_id();
    const uint32_t wave_id = uniform(blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6));
    const uint32_t claim_chunks = static_cast<uint32_t>(min(64ull, max(1ull, (static_cast<unsigned long long>(RV_WF_CLAIM_RAYS) * p.wf_chunks + live - 1ull) / live)));
    ChunkPool pool;
    pool.shard = wave_id % kClaimShards;
    while (open_next_chunk(pool, p, p.counter, claim_chunks, lane, wave_id)) {
        for (uint32_t pos = 0; pos < pool.cnt; pos += 64u) {
            if (pos + lane < pool.cnt) {
                const uint32_t q = pool.base + pos + lane;
                const v4f r0 = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p.wf_rays + 2 * q));
                const v2f r1 = __builtin_nontemporal_load(reinterpret_cast<const v2f *>(p.wf_rays + 2 * q + 1));
                const f3 o = mk(r0.x, r0.y, r0.z), d = mk(r0.w, r1.x, r1.y);
                float closest = kInf;
                uint32_t hit = 0xFFFFFFFFu;
                if (EARLY_OUT)
                    intersect_run_early(src, p.n_tris, o, d, closest, hit);
                else
                    intersect_run<4>(src, 0u, p.n_tris, o, d, closest, hit);
                v2f r;
                r.x = closest, r.y = __uint_as_float(hit);
                __builtin_nontemporal_store(r, reinterpret_cast<v2f *>(p.wf_hits) + q);
            }
        }
    }
}
template __global__ void wf_trace_brute<false>(const FrameParams);
template __global__ void wf_trace_brute<true>(const FrameParams);

}  // namespace rv

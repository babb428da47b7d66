// rvpt_bvh8.hip — the reference's BVH traversal (intersection.glsl:361-413) over the 8-WIDE form of its tree: rvpt_bvh4.hip's walk with eight children per step.
//
// Why: the walk is bound by the length of a step's dependent chain times the waves that hide it (profiles/EXPERIMENTS.md: what binds the binary walk), and a host-side count over the bench scenes
// (profiles/r05_wide_steps.txt) says what another doubling of the width buys: on the Cornell scene a ray visits 3.9 four-wide nodes and tests 15.2 boxes, but
// only 2.0 eight-wide nodes and 14.4 boxes — half the steps for the same slab arithmetic; on the 1 M-triangle terrain 17.9 -> 12.6 steps per bounce ray for
// 69 -> 90 boxes.  Exactness is rvpt_bvh4.hip's argument unchanged (bvh_wide.cpp: a wide node lists descendants of one binary node in depth-first order,
// collapsed only across boxes that contain their children; every child's OWN box is tested at the parent with the closest_t of that moment; a stacked child
// carries its exact entry distance, so the reference's test at its visit is closest_t >= entry).
//
// Device layout (bvh_wide.cpp: build_wide8_nodes): 256 B per node — minx[8] maxx[8] miny[8] maxy[8] minz[8] maxz[8] (two quads each: children 0-3, 4-7), head[8],
// two quads of padding.  One flat address per step and fourteen dwordx4 loads off it; the first nodes (the upper levels) and the first stack levels live in LDS.
// The seven pushes of a step are branch-free when every lane's pushes fit the LDS levels: slot writes go out unconditionally at the lane's stack pointer, which
// advances only for a child that passed (and is not the one entered) — a write past it lands on a free slot; one spare LDS level takes the writes of a lane
// whose pointer stands at the last level.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rvpt_device.h"

#ifndef RV_BVH8_MIN_WAVES
#define RV_BVH8_MIN_WAVES 5  // 56 VGPRs of node per step beside the path's state: 96-102 VGPRs
#endif

namespace rv {

__global__ __launch_bounds__(kBlock, RV_BVH8_MIN_WAVES) void trace_bvh8(const FrameParams p)
{
    // LDS: [stack: (stack_lds_levels + 1) x 2 words x kBlock — the last level is the spare one][root record: 2 float4][the first wide_top_nodes nodes: 16 float4 each]
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    const uint32_t lds_levels = p.stack_lds_levels;
    float4 *lds_root = reinterpret_cast<float4 *>(lds_stack + 2u * (lds_levels + 1u) * kBlock);
    float4 *lds_top = lds_root + 2;
    const uint32_t top_nodes = p.wide_top_nodes;
    if (threadIdx.x < 2u) lds_root[threadIdx.x] = p.nodes[threadIdx.x];
    for (uint32_t i = threadIdx.x; i < 16u * top_nodes; i += kBlock) lds_top[i] = p.wide[i];
    __syncthreads();
    const v4f *prep = reinterpret_cast<const v4f *>(p.prep);
    const ShadeSrc shade_src{p.prep, p.mat_index, p.mats, p.unit_n};
    const uint32_t top_level = p.stack_levels - 1u;
    const uint32_t head_shift = p.head_shift;
    uint32_t *const ovf = p.stack_overflow + (static_cast<size_t>(blockIdx.x) * kBlock + threadIdx.x);
    const size_t ovf_stride = static_cast<size_t>(gridDim.x) * kBlock;

    const uint32_t lane = lane_id();
    const uint32_t wave_id = uniform(blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6));
    WavePool pool;
    pool.shard = wave_id % kClaimShards;
    Lane L{};
    uint32_t nsmp = 0;
    bool have_pixel = false, need_sample = false;

    enum { S_IDLE = 0, S_TRAV = 1, S_HIT = 2 };
    int state = S_IDLE;
    bool walking = false;
    float closest = kInf;
    uint32_t hit = 0xFFFFFFFFu, sp = 0;
    uint32_t cur = 0;                         // wide node being processed
    uint32_t leaf_first = 0, leaf_count = 0;  // leaf reached (its box passed), waiting for its triangle tests
    auto enter = [&](const uint32_t head) {   // head = first | count << head_shift (leaf, count > 0) or a wide node index (count 0)
        const uint32_t first = head & ((1u << head_shift) - 1u), count = head >> head_shift;
        cur = first;
        leaf_first = first;
        leaf_count = count;
    };
    auto push_slow = [&](const float entry, const uint32_t head) {  // any level (rvpt_bvh4.hip's push)
        report_stack_overflow(p, sp > top_level);
        const uint32_t at = min(sp, top_level);
        if (at < lds_levels) {
            lds_stack[(2u * at + 0u) * kBlock + threadIdx.x] = __float_as_uint(entry);
            lds_stack[(2u * at + 1u) * kBlock + threadIdx.x] = head;
        } else {
            ovf[(2u * (at - lds_levels) + 0u) * ovf_stride] = __float_as_uint(entry);
            ovf[(2u * (at - lds_levels) + 1u) * ovf_stride] = head;
        }
        sp = at + 1u;
    };
    f3 inv = mk(0.0f, 0.0f, 0.0f);

    for (;;) {
        // ---- refill: every lane that is not traversing gets its next query (trace_bvh's loop)
        for (;;) {
            if (state == S_HIT) {
                f3 radiance = mk(0.0f, 0.0f, 0.0f);
                L.nseg += 1;
                const bool done = shade(L, p, shade_src, hit, closest, radiance);
                state = S_IDLE;
                if (done)
                    retire(L, p, true, radiance, have_pixel, need_sample);
                else
                    state = S_TRAV;
            }
            regenerate<true, false>(pool, p, lane, wave_id, have_pixel, need_sample, L);
            if (have_pixel && need_sample && state == S_IDLE) {
                begin_sample(L, p);
                need_sample = false;
                nsmp += 1;
                if (p.max_bounces > 0)
                    state = S_TRAV;
                else
                    retire(L, p, true, mk(0.0f, 0.0f, 0.0f), have_pixel, need_sample);
            }
            if (state == S_TRAV && !walking) {  // start at the root: its own box first (the root is a node like any other, intersection.glsl:369-380)
                closest = kInf;
                hit = 0xFFFFFFFFu;
                inv = mk(1.0f / L.d.x, 1.0f / L.d.y, 1.0f / L.d.z);
                sp = 0;
                float entry;
                if (slab_entry(L.o, inv, lds_root[0], lds_root[1], closest, entry)) {
                    cur = 0;  // the wide root
                    leaf_count = 0;
                    walking = true;
                } else {
                    state = S_HIT;
                }
            }
            const bool more = (have_pixel && state != S_TRAV) || (!have_pixel && !pool.exhausted);
            if (ballot(more) == 0) break;
        }
        if (ballot(state == S_TRAV) == 0) break;

        // ---- traverse: every iteration each walking lane handles one wide node; leaves are parked and run in batches (trace_bvh)
        for (uint32_t steps = 0;; ++steps) {
            bool need_pop = false;
            if (state == S_TRAV && leaf_count == 0) {
                const float4 *node = (cur < top_nodes) ? lds_top + 16 * cur : p.wide + 16 * cur;
                const float4 minx0 = node[0], minx1 = node[1], maxx0 = node[2], maxx1 = node[3], miny0 = node[4], miny1 = node[5], maxy0 = node[6], maxy1 = node[7];
                const float4 minz0 = node[8], minz1 = node[9], maxz0 = node[10], maxz1 = node[11], hq0 = node[12], hq1 = node[13];
                const uint32_t hd[8] = {__float_as_uint(hq0.x), __float_as_uint(hq0.y), __float_as_uint(hq0.z), __float_as_uint(hq0.w),
                                        __float_as_uint(hq1.x), __float_as_uint(hq1.y), __float_as_uint(hq1.z), __float_as_uint(hq1.w)};
                float e[8];
                bool h[8];
                h[0] = slab_child(L.o, inv, minx0.x, maxx0.x, miny0.x, maxy0.x, minz0.x, maxz0.x, closest, e[0]);  // (a wide node has at least two children)
                h[1] = slab_child(L.o, inv, minx0.y, maxx0.y, miny0.y, maxy0.y, minz0.y, maxz0.y, closest, e[1]);
                h[2] = slab_child(L.o, inv, minx0.z, maxx0.z, miny0.z, maxy0.z, minz0.z, maxz0.z, closest, e[2]) && hd[2] != kWideEmpty;
                h[3] = slab_child(L.o, inv, minx0.w, maxx0.w, miny0.w, maxy0.w, minz0.w, maxz0.w, closest, e[3]) && hd[3] != kWideEmpty;
                h[4] = slab_child(L.o, inv, minx1.x, maxx1.x, miny1.x, maxy1.x, minz1.x, maxz1.x, closest, e[4]) && hd[4] != kWideEmpty;
                h[5] = slab_child(L.o, inv, minx1.y, maxx1.y, miny1.y, maxy1.y, minz1.y, maxz1.y, closest, e[5]) && hd[5] != kWideEmpty;
                h[6] = slab_child(L.o, inv, minx1.z, maxx1.z, miny1.z, maxy1.z, minz1.z, maxz1.z, closest, e[6]) && hd[6] != kWideEmpty;
                h[7] = slab_child(L.o, inv, minx1.w, maxx1.w, miny1.w, maxy1.w, minz1.w, maxz1.w, closest, e[7]) && hd[7] != kWideEmpty;
                // the first child that passes is visited now, the others wait on the stack in order: the last is pushed first
                bool lower[8];  // lower[k]: some child before k passes
                lower[0] = false;
#pragma unroll
                for (int k = 1; k < 8; ++k) lower[k] = lower[k - 1] || h[k - 1];
                const uint32_t n_push = (h[1] && lower[1]) + (h[2] && lower[2]) + (h[3] && lower[3]) + (h[4] && lower[4]) + (h[5] && lower[5]) + (h[6] && lower[6]) + (h[7] && lower[7]);
                if (ballot(sp + n_push > lds_levels) == 0) {
                    // every lane's pushes fit the LDS levels: unconditional slot writes at the lane's own pointer (a write for a child that does not count lands
                    // on the free slot above the stack — the spare level when the pointer stands at lds_levels — and is overwritten or ignored)
#pragma unroll
                    for (int k = 7; k >= 1; --k) {
                        lds_stack[(2u * sp + 0u) * kBlock + threadIdx.x] = __float_as_uint(e[k]);
                        lds_stack[(2u * sp + 1u) * kBlock + threadIdx.x] = hd[k];
                        sp += (h[k] && lower[k]) ? 1u : 0u;
                    }
                } else {
#pragma unroll
                    for (int k = 7; k >= 1; --k)
                        if (h[k] && lower[k]) push_slow(e[k], hd[k]);
                }
                if (lower[7] || h[7]) {
                    uint32_t head = hd[7];
#pragma unroll
                    for (int k = 6; k >= 0; --k) head = h[k] ? hd[k] : head;
                    enter(head);
                } else {
                    need_pop = true;
                }
            }
            const uint32_t at_leaf = static_cast<uint32_t>(__builtin_popcountll(ballot(leaf_count > 0)));
            const uint32_t at_inner = static_cast<uint32_t>(__builtin_popcountll(ballot(state == S_TRAV && leaf_count == 0)));
            const bool run_leaves = at_leaf > 0 && (at_inner == 0 || at_leaf >= p.bvh_leaf_batch);
            if (run_leaves && leaf_count > 0) {
                for (uint32_t i = leaf_first; i < leaf_first + leaf_count; ++i) {
                    const v4f *tp = prep + 4 * i;
                    const PrepTri t = unpack(tp[0], tp[1], tp[2], tp[3]);
                    test_triangle(t, L.o, L.d, i, closest, hit);
                }
                leaf_count = 0;
                need_pop = true;
            }
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
                    if (closest >= __uint_as_float(entry_bits)) {  // the reference's box test at the visit (see the header comment)
                        enter(cand);
                        found = true;
                    }
                }
                if (!found) {
                    state = S_HIT;
                    walking = false;
                }
            }
            if (ballot(state == S_TRAV) == 0) break;
            const uint32_t waiting = static_cast<uint32_t>(__builtin_popcountll(ballot(state == S_HIT || (!have_pixel && !pool.exhausted))));
            if (waiting >= p.bvh_refill || (waiting > 0 && steps >= 4u * p.bvh_refill)) break;
        }
    }
    wave_exit(p, lane, L.nseg, nsmp);
}

}  // namespace rv

// rvpt_bvh4.hip — the reference's BVH traversal (intersection.glsl:361-413) over the 4-WIDE form of its tree.
//
// trace_bvh (rvpt_kernels.hip) is bound by the length of a traversal step's dependent chain — pop / decide, one address, a 64-byte fetch at L2
// latency, two slab tests, decide again — times the waves per SIMD available to hide it (profiles/EXPERIMENTS.md: what binds the binary walk), and a ray needs ten to thirty such steps.
// A step over a node with FOUR children has the same chain and decides two levels of the binary tree at once: half the dependent round trips per
// ray, four independent slab tests in flight instead of two.
//
// Why it is the same traversal, bit for bit (bvh_wide.cpp: build_wide_nodes has the argument in full): in a tree whose boxes contain their
// children's boxes the slab test is monotone under containment, so the reference visits a node iff the node's OWN box passes at the moment its
// depth-first, left-first order reaches it; inner nodes only cull.  A wide node lists up to four descendants of one binary node in that order
// (children of children, collapsed only across boxes that do contain their children), the kernel tests all of them at the parent with the
// closest_t of that moment, continues with the first that passes and stacks the others in order with their exact entry distances — the test the
// reference makes when it reaches a stacked node, min(t_exit, closest_t) >= entry, is then exactly closest_t >= entry (trace_bvh's argument:
// closest_t only shrinks and t_exit >= entry held when the node was stacked).  Leaves, triangle tests, shading: trace_bvh's code.
//
// Four instances (reference child order, ray regeneration): scenes that do not fit LDS / scenes that do (every wide node, the triangles and the
// materials in LDS, camera packets over the wide nodes in the lean instance), each for the lean configuration (Kajiya everywhere, pinhole camera) and
// GENERIC (every render / camera mode of compute_pass.comp).  Stack slots are trace_bvh's two words (entry distance, packed head); the first stack
// levels and the first wide nodes (the upper levels of the tree) live in LDS, the rest of the stack in one global column per thread and level.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rvpt_device.h"

#ifndef RV_BVH4_BRANCH_FREE_PUSH
#define RV_BVH4_BRANCH_FREE_PUSH 1
#endif
#ifndef RV_BVH4_TOP_QUADS
#define RV_BVH4_TOP_QUADS 8
#endif
#ifndef RV_BVH4Q_MIN_WAVES
#define RV_BVH4Q_MIN_WAVES 5  // 92 VGPRs; forced into 80 it spills six and loses another 10-20 % (profiles/r05_bvh4_quant.txt)
#endif
#ifndef RV_BVH4_MIN_WAVES
#define RV_BVH4_MIN_WAVES 6  // 80 VGPRs + one spilled register: a sixth wave per SIMD (84 without: five) measured +2 % C3, +4.5 % C4 geometry (tools/archive/sweep_wide_knobs.sh)
#endif

namespace rv {

// RESIDENT: the whole scene fits LDS beside the stack — every wide node (wide_top_nodes == n_wide), the prepared triangles, material indices and
// materials are copied there and nothing but the stack's overflow levels and the sample stores touches global memory.
// GENERIC: the other render / camera modes of compute_pass.comp (rvpt_device.h: shade_generic, begin_sample_generic) over the same walk; camera packets only in
// the lean instances (the other cameras have no common origin to make neighbouring rays coherent).
constexpr uint32_t kTopQuads = RV_BVH4_TOP_QUADS;  // float4 between two wide nodes in LDS (rvpt_abi.hip sizes the LDS with the same figure: kWideTopQuads)

template <bool RESIDENT, bool GENERIC, bool QUANT>
__device__ __forceinline__ void bvh4_body(const FrameParams &p)
{
    // LDS: [stack: stack_lds_levels x 2 words x kBlock][root record: 2 float4]
    //      [the first wide_top_nodes wide nodes, kTopQuads float4 apart][RESIDENT: prepared triangles, material index per triangle, materials]
    // kTopQuads: 8 as shipped (nodes packed, 128 bytes apart).  Lanes in different top nodes read the same quad of their nodes at once (seven ds_read_b128 per
    // step) and 128 bytes apart all of them start on bank 0 or 32 (SQ_LDS_BANK_CONFLICT: 39 % of the LDS-active cycles on the Cornell scene,
    // profiles/r04_c3_wide_pmc.json); 9 = 144 bytes apart is conflict free and measured +-0 (profiles/EXPERIMENTS.md 5.9), so it stays a build knob (RV_BVH4_TOP_QUADS).
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_stack[];
    float4 *lds_root = reinterpret_cast<float4 *>(lds_stack + 2u * p.stack_lds_levels * kBlock);
    float4 *lds_top = lds_root + 2;
    const uint32_t top_nodes = p.wide_top_nodes;
    if (threadIdx.x < 2u) lds_root[threadIdx.x] = p.nodes[threadIdx.x];
    if (QUANT)
        for (uint32_t i = threadIdx.x; i < 4u * top_nodes; i += kBlock) lds_top[i] = p.wide[i];  // 64-byte nodes, packed
    else
        for (uint32_t i = threadIdx.x; i < 8u * top_nodes; i += kBlock) lds_top[(i >> 3) * kTopQuads + (i & 7u)] = p.wide[i];
    float4 *lds_prep = lds_top + (QUANT ? 4u : kTopQuads) * top_nodes;
    uint32_t *lds_mat_index = reinterpret_cast<uint32_t *>(lds_prep + 4u * p.n_tris);
    float4 *lds_mats = reinterpret_cast<float4 *>(lds_mat_index + ((p.n_tris + 3u) & ~3u));
    if (RESIDENT) {
        for (uint32_t i = threadIdx.x; i < 4u * p.n_tris; i += kBlock) lds_prep[i] = p.prep[i];
        for (uint32_t i = threadIdx.x; i < p.n_tris; i += kBlock) lds_mat_index[i] = p.mat_index[i];
        for (uint32_t i = threadIdx.x; i < 3u * p.n_mats; i += kBlock) lds_mats[i] = p.mats[i];
    }
    __syncthreads();
    const v4f *prep = reinterpret_cast<const v4f *>(RESIDENT ? lds_prep : p.prep);
    const ShadeSrc shade_src = RESIDENT ? ShadeSrc{lds_prep, lds_mat_index, lds_mats, p.unit_n} : ShadeSrc{p.prep, p.mat_index, p.mats, p.unit_n};
    const uint32_t top_level = p.stack_levels - 1u;
    const uint32_t head_shift = p.head_shift;  // (> 0: the wide form exists only for trees whose heads pack)
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
    auto push = [&](const float entry, const uint32_t head) {
        report_stack_overflow(p, sp > top_level);
        const uint32_t at = min(sp, top_level);  // the host sized the stack from the wide tree (build_wide_nodes): sp never passes top_level
        if (at < lds_levels) {
            lds_stack[(2u * at + 0u) * kBlock + threadIdx.x] = __float_as_uint(entry);
            lds_stack[(2u * at + 1u) * kBlock + threadIdx.x] = head;
        } else {
            ovf[(2u * (at - lds_levels) + 0u) * ovf_stride] = __float_as_uint(entry);
            ovf[(2u * (at - lds_levels) + 1u) * ovf_stride] = head;
        }
        sp = at + 1u;  // (= sp + 1 unless the push was clamped: the pops then stay inside the stack)
    };
    f3 inv = mk(0.0f, 0.0f, 0.0f);
    // QUANT (trace_bvh4q): the nodes are the 64-byte form of bvh_wide.cpp (build_quant_nodes) — child boxes as 8-bit offsets from the node's corner, rounded
    // outward — and the children are tested with the CONSERVATIVE slab test of rvpt_device.h (quant_slab_*): it accepts whatever the reference's test
    // accepts on the child's exact box, so inner children cull as before, only a little less; a LEAF's own box is tested exactly — the reference's
    // test at its visit — from p.leaf_box when the lane reaches it.  Four 16-byte loads per step instead of seven: what a step costs is its vector-memory
    // instructions through the CU's address / L1 path (profiles/r05_bvh4_fast.txt).
    QuantRay qr{};
    auto quant_setup = [&]() {
        if (QUANT) qr = quant_slab_setup(L.o, inv, p.slab_extent);
    };
#ifdef RV_BVH_PROFILE  // experiments only (tools/bvh_phase_profile.py): where a wave's time goes — trace_bvh's rows
    unsigned long long pf_refill = 0, pf_inner = 0, pf_leaf = 0, pf_pop = 0, pf_iters = 0, pf_leaf_phases = 0, pf_inner_lanes = 0, pf_leaf_lanes = 0,
                       pf_refill_lanes = 0, pf_refills = 0, pf_t0 = __builtin_amdgcn_s_memtime(), pf_mark = 0, pf_hist = 0, pf_dry_iters = 0, pf_ld = 0, pf_lds = 0;
    const unsigned long long pf_wall0 = wall_clock64();  // 100 MHz wall clock: when the wave started, found the pixel pool dry, ended (second half of the timeline rows)
    unsigned long long pf_wall_dry = 0;
#endif

    for (;;) {
#ifdef RV_BVH_PROFILE
        pf_mark = __builtin_amdgcn_s_memtime();
        pf_refills += 1;
        pf_refill_lanes += __builtin_popcountll(ballot(state != S_TRAV));
#endif
        // ---- refill: every lane that is not traversing gets its next query (trace_bvh's loop)
        for (;;) {
            if (state == S_HIT) {
                f3 radiance = mk(0.0f, 0.0f, 0.0f);
                L.nseg += 1;
                const bool done = shade_t<GENERIC>(L, p, shade_src, hit, closest, radiance);
                state = S_IDLE;
                if (done)
                    retire(L, p, true, radiance, have_pixel, need_sample);
                else
                    state = S_TRAV;
            }
            regenerate<true, GENERIC>(pool, p, lane, wave_id, have_pixel, need_sample, L);
            if (have_pixel && need_sample && state == S_IDLE) {
                begin_sample_t<GENERIC>(L, p);
                need_sample = false;
                nsmp += 1;
                if (wants_trace<GENERIC>(L, p))
                    state = S_TRAV;
                else
                    retire(L, p, true, mk(0.0f, 0.0f, 0.0f), have_pixel, need_sample);
            }
            if (RESIDENT && !GENERIC) {
                // ---- camera packet (trace_bvh<..., CAMPACK>'s walk, DESIGN.md 5.3 / profiles/EXPERIMENTS.md 4.1, over wide nodes): lanes that start a camera ray in this refill walk
                // the tree TOGETHER — wave-uniform node, every lane masked by its own box tests, the stack in the lanes' own columns with NaN entry
                // distances for the lanes a stacked child does not concern — and leave it (continue per lane, their column being their stack) when at
                // most bvh_detach of them are in a node.  The reference's fixed child order makes every lane's sequence of passed boxes and tested
                // triangles its solo sequence.
                const bool fresh = state == S_TRAV && !walking && L.bounce == 0;
                if (static_cast<uint32_t>(__builtin_popcountll(ballot(fresh))) >= p.bvh_cam_min) {
                    bool pk = fresh;
                    if (pk) {
                        closest = kInf;
                        hit = 0xFFFFFFFFu;
                        inv = mk(1.0f / L.d.x, 1.0f / L.d.y, 1.0f / L.d.z);
                        sp = 0;
                    }
                    float e_root;
                    bool in = pk && slab_entry(L.o, inv, lds_root[0], lds_root[1], closest, e_root);
                    if (pk && !in) {
                        state = S_HIT;
                        pk = false;
                    }
                    uint32_t ufirst = 0u, ucount = 0u;  // the current node (wave-uniform): the wide root
                    uint32_t usp = 0;
                    for (;;) {
                        const uint64_t m_in = ballot(in);
                        bool pop = true;
                        if (m_in != 0) {
                            if (static_cast<uint32_t>(__builtin_popcountll(m_in)) <= p.bvh_detach) {
                                if (in) {
                                    cur = ufirst;
                                    leaf_first = ufirst;
                                    leaf_count = ucount;
                                    sp = usp;
                                    walking = true;
                                    pk = false;
                                    in = false;
                                }
                            } else if (ucount == 0) {
                                const float4 *node = lds_top + kTopQuads * ufirst;  // one address for the whole wave
                                const float4 minx = node[0], maxx = node[1], miny = node[2], maxy = node[3], minz = node[4], maxz = node[5], hq = node[6];
                                const uint32_t hd[4] = {uniform(__float_as_uint(hq.x)), uniform(__float_as_uint(hq.y)), uniform(__float_as_uint(hq.z)), uniform(__float_as_uint(hq.w))};
                                float e[4];
                                bool h[4];
                                h[0] = slab_child(L.o, inv, minx.x, maxx.x, miny.x, maxy.x, minz.x, maxz.x, closest, e[0]) && in;
                                h[1] = slab_child(L.o, inv, minx.y, maxx.y, miny.y, maxy.y, minz.y, maxz.y, closest, e[1]) && in;
                                h[2] = slab_child(L.o, inv, minx.z, maxx.z, miny.z, maxy.z, minz.z, maxz.z, closest, e[2]) && in && hd[2] != kWideEmpty;
                                h[3] = slab_child(L.o, inv, minx.w, maxx.w, miny.w, maxy.w, minz.w, maxz.w, closest, e[3]) && in && hd[3] != kWideEmpty;
                                const bool any0 = ballot(h[0]) != 0, any1 = ballot(h[1]) != 0, any2 = ballot(h[2]) != 0, any3 = ballot(h[3]) != 0;
                                const bool anyk[4] = {any0, any1, any2, any3};
                                // the first child some lane is in is visited now; the later ones some lane is in wait on the stack, the last pushed first
                                bool lower = any0 || any1 || any2;
#pragma unroll
                                for (int k = 3; k >= 1; --k) {
                                    if (k == 2) lower = any0 || any1;
                                    if (k == 1) lower = any0;
                                    if (anyk[k] && lower) {
                                        report_stack_overflow(p, usp > top_level);
                                        const uint32_t at = min(usp, top_level);
                                        const uint32_t ent = h[k] ? __float_as_uint(e[k]) : 0x7FC00000u;
                                        if (pk) {
                                            if (at < lds_levels) {
                                                lds_stack[(2u * at + 0u) * kBlock + threadIdx.x] = ent;
                                                lds_stack[(2u * at + 1u) * kBlock + threadIdx.x] = hd[k];
                                            } else {
                                                ovf[(2u * (at - lds_levels) + 0u) * ovf_stride] = ent;
                                                ovf[(2u * (at - lds_levels) + 1u) * ovf_stride] = hd[k];
                                            }
                                        }
                                        usp = at + 1u;
                                    }
                                }
                                if (any0 || any1 || any2 || any3) {
                                    const int first = any0 ? 0 : (any1 ? 1 : (any2 ? 2 : 3));
                                    in = any0 ? h[0] : (any1 ? h[1] : (any2 ? h[2] : h[3]));
                                    const uint32_t head = hd[first];
                                    ufirst = head & ((1u << head_shift) - 1u);
                                    ucount = head >> head_shift;
                                    pop = false;
                                }
                            } else {
                                for (uint32_t i = ufirst; i < ufirst + ucount; ++i) {
                                    const v4f *tp = prep + 4 * i;
                                    const PrepTri t = unpack(tp[0], tp[1], tp[2], tp[3]);
                                    float c2 = closest;
                                    uint32_t h2 = hit;
                                    test_triangle(t, L.o, L.d, i, c2, h2);
                                    closest = in ? c2 : closest;
                                    hit = in ? h2 : hit;
                                }
                            }
                        }
                        if (pop) {
                            bool found = false;
                            while (usp > 0 && !found) {
                                usp -= 1;
                                uint32_t entry_bits, cand;
                                if (usp < lds_levels) {
                                    entry_bits = lds_stack[(2u * usp + 0u) * kBlock + threadIdx.x];
                                    cand = lds_stack[(2u * usp + 1u) * kBlock + threadIdx.x];
                                } else {
                                    entry_bits = ovf[(2u * (usp - lds_levels) + 0u) * ovf_stride];
                                    cand = ovf[(2u * (usp - lds_levels) + 1u) * ovf_stride];
                                }
                                in = pk && closest >= __uint_as_float(entry_bits);
                                const uint64_t m = ballot(in);
                                if (m != 0) {
                                    const uint32_t ucand = __builtin_amdgcn_readlane(cand, static_cast<uint32_t>(__builtin_ctzll(m)));
                                    ufirst = ucand & ((1u << head_shift) - 1u);
                                    ucount = ucand >> head_shift;
                                    found = true;
                                }
                            }
                            if (!found) break;
                        }
                    }
                    if (pk) state = S_HIT;  // never left the packet: its traversal is complete
                }
            }
            if (state == S_TRAV && !walking) {  // start at the root: its own box first (the root is a node like any other, intersection.glsl:369-380)
                closest = kInf;
                hit = 0xFFFFFFFFu;
                inv = mk(1.0f / L.d.x, 1.0f / L.d.y, 1.0f / L.d.z);
                quant_setup();
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
#ifdef RV_BVH_PROFILE
        pf_refill += __builtin_amdgcn_s_memtime() - pf_mark;
#endif

        // ---- traverse: every iteration each walking lane handles one wide node; leaves are parked and run in batches (trace_bvh)
        for (uint32_t steps = 0;; ++steps) {
#ifdef RV_BVH_PROFILE
            pf_mark = __builtin_amdgcn_s_memtime();
            pf_iters += 1;
            {
                const unsigned long long nw = __builtin_popcountll(ballot(state == S_TRAV && leaf_count == 0));
                pf_inner_lanes += nw;
                pf_hist += 1ull << (16u * static_cast<uint32_t>(nw > 48 ? 3 : nw > 32 ? 2 : nw > 16 ? 1 : 0));
                if (pool.exhausted) pf_dry_iters += 1;
                if (pool.exhausted && pf_wall_dry == 0) pf_wall_dry = wall_clock64();
            }
#endif
            bool need_pop = false;
            if (state == S_TRAV && leaf_count == 0) {
                // ONE address per lane — the node's 128 bytes in the LDS copy of the tree top or in global memory — and seven FLAT loads off it
                // (written as select + add on integers: with the two strides the compiler otherwise turns the pointer select into a divergent branch)
                const bool in_lds = RESIDENT || cur < top_nodes;
                const uint64_t node_base = in_lds ? reinterpret_cast<uint64_t>(lds_top) : reinterpret_cast<uint64_t>(p.wide);
                const uint32_t node_off = QUANT ? (cur << 6) : (cur << 7) + (in_lds ? cur * (16u * kTopQuads - 128u) : 0u);  // 32 bits: upload_scene keeps n_wide < 2^25 (kWideMaxNodes)
                float e0, e1, e2, e3;
                bool h0, h1, h2, h3;
                uint32_t hd0, hd1, hd2, hd3;
                if (QUANT) {
                    const float4 *node = reinterpret_cast<const float4 *>(node_base + node_off);
                    const float4 qa = node[0], qb = node[1], qc = node[2], hq = node[3];
                    hd0 = __float_as_uint(hq.x), hd1 = __float_as_uint(hq.y), hd2 = __float_as_uint(hq.z), hd3 = __float_as_uint(hq.w);
                    const QuantNode qn = quant_slab_node(qr, inv_q(qr, inv), qa, qb, qc);
                    h0 = quant_slab_child<0>(qn, closest, e0);
                    h1 = quant_slab_child<1>(qn, closest, e1);
                    h2 = quant_slab_child<2>(qn, closest, e2) && hd2 != kWideEmpty;
                    h3 = quant_slab_child<3>(qn, closest, e3) && hd3 != kWideEmpty;
                } else {
                    const float4 *node = reinterpret_cast<const float4 *>(node_base + node_off);
#ifdef RV_BVH_PROFILE
                    asm volatile("" ::: "memory");
                    const unsigned long long pf_l0 = __builtin_amdgcn_s_memtime();
                    asm volatile("" ::: "memory");
#endif
                    const float4 minx = node[0], maxx = node[1], miny = node[2], maxy = node[3], minz = node[4], maxz = node[5], hq = node[6];
#ifdef RV_BVH_PROFILE  // issue-to-arrival time of a step's seven loads, as the wave sees it
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    pf_ld += __builtin_amdgcn_s_memtime() - pf_l0;
                    pf_lds += 1;
#endif
                    hd0 = __float_as_uint(hq.x), hd1 = __float_as_uint(hq.y), hd2 = __float_as_uint(hq.z), hd3 = __float_as_uint(hq.w);
                    h0 = slab_child(L.o, inv, minx.x, maxx.x, miny.x, maxy.x, minz.x, maxz.x, closest, e0);  // (a wide node has at least two children)
                    h1 = slab_child(L.o, inv, minx.y, maxx.y, miny.y, maxy.y, minz.y, maxz.y, closest, e1);
                    h2 = slab_child(L.o, inv, minx.z, maxx.z, miny.z, maxy.z, minz.z, maxz.z, closest, e2) && hd2 != kWideEmpty;
                    h3 = slab_child(L.o, inv, minx.w, maxx.w, miny.w, maxy.w, minz.w, maxz.w, closest, e3) && hd3 != kWideEmpty;
                }
                // the first child that passes is visited now, the others wait on the stack in order: the last is pushed first
                const bool p3 = h3 && (h0 || h1 || h2), p2 = h2 && (h0 || h1), p1 = h1 && h0;
                if (RV_BVH4_BRANCH_FREE_PUSH && ballot(sp + 3u > lds_levels) == 0) {
                    // every lane has three free LDS levels above its stack (the common case: the stack is a few entries deep): the three slot writes go out
                    // unconditionally at the lane's own stack pointer, which advances only for a child that counts — a write that does not count lands on a
                    // free slot above the stack and is overwritten or ignored.  No branch, no exec-mask bookkeeping.
                    lds_stack[(2u * sp + 0u) * kBlock + threadIdx.x] = __float_as_uint(e3);
                    lds_stack[(2u * sp + 1u) * kBlock + threadIdx.x] = hd3;
                    sp += p3 ? 1u : 0u;
                    lds_stack[(2u * sp + 0u) * kBlock + threadIdx.x] = __float_as_uint(e2);
                    lds_stack[(2u * sp + 1u) * kBlock + threadIdx.x] = hd2;
                    sp += p2 ? 1u : 0u;
                    lds_stack[(2u * sp + 0u) * kBlock + threadIdx.x] = __float_as_uint(e1);
                    lds_stack[(2u * sp + 1u) * kBlock + threadIdx.x] = hd1;
                    sp += p1 ? 1u : 0u;
                } else {
                    if (p3) push(e3, hd3);
                    if (p2) push(e2, hd2);
                    if (p1) push(e1, hd1);
                }
                if (h0 || h1 || h2 || h3)
                    enter(h0 ? hd0 : (h1 ? hd1 : (h2 ? hd2 : hd3)));
                else
                    need_pop = true;
            }
#ifdef RV_BVH_PROFILE
            pf_inner += __builtin_amdgcn_s_memtime() - pf_mark;
            pf_mark = __builtin_amdgcn_s_memtime();
#endif
            bool run_leaves = true;  // LDS-resident scenes: traversals are short, parking does not pay (trace_bvh: measured)
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
                if (QUANT) {
                    // the leaf's OWN box now, exactly — the reference's test at the visit (intersection.glsl:377-380).  The first triangle's record is fetched
                    // beside the box and tested whatever the box says (its result counts only if the box passes: nearly always), so that a leaf visit
                    // stays ONE round trip to memory.
                    const v4f *tp = prep + 4 * leaf_first;
                    const float4 b0 = p.leaf_box[2u * leaf_first], b1 = p.leaf_box[2u * leaf_first + 1u];
                    const v4f q0 = tp[0], q1 = tp[1], q2 = tp[2], q3 = tp[3];
                    asm volatile("" ::"v"(q0), "v"(q1), "v"(q2), "v"(q3));  // (fetched HERE, with the box: the compiler would sink them behind the box test)
                    const PrepTri t = unpack(q0, q1, q2, q3);
                    const bool pass = slab_leaf(L.o, inv, b0, b1, closest);
                    float c2 = closest;
                    uint32_t h2 = hit;
                    test_triangle(t, L.o, L.d, leaf_first, c2, h2);
                    closest = pass ? c2 : closest;
                    hit = pass ? h2 : hit;
                    if (pass)
                        for (uint32_t i = leaf_first + 1u; i < leaf_first + leaf_count; ++i) {
                            const v4f *tq = prep + 4 * i;
                            test_triangle(unpack(tq[0], tq[1], tq[2], tq[3]), L.o, L.d, i, closest, hit);
                        }
                } else {
                    for (uint32_t i = leaf_first; i < leaf_first + leaf_count; ++i) {
                        const v4f *tp = prep + 4 * i;
                        const PrepTri t = unpack(tp[0], tp[1], tp[2], tp[3]);
                        test_triangle(t, L.o, L.d, i, closest, hit);
                    }
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
#ifdef RV_BVH_PROFILE
            pf_pop += __builtin_amdgcn_s_memtime() - pf_mark;
#endif
            if (ballot(state == S_TRAV) == 0) break;
            const uint32_t waiting = static_cast<uint32_t>(__builtin_popcountll(ballot(state == S_HIT || (!have_pixel && !pool.exhausted))));
            if (waiting >= p.bvh_refill || (waiting > 0 && steps >= 4u * p.bvh_refill)) break;
        }
    }
#ifdef RV_BVH_PROFILE
    if (p.timeline && lane == 0) {
        unsigned long long *t = p.timeline + 8ull * wave_id;
        t[0] = pf_refill;
        t[1] = pf_inner + pf_pop;  // (tools/bvh_phase_profile.py's rows: pops count as inner-node work; the pop share alone rides in the upper half of row 2)
        t[2] = pf_leaf | (pf_pop << 40);
        t[3] = pf_iters | (pf_leaf_phases << 32);
        t[4] = pf_inner_lanes | (pf_leaf_lanes << 32);
        t[5] = pf_hist;
        t[6] = pf_refill_lanes | (pf_refills << 32);
        t[7] = (__builtin_amdgcn_s_memtime() - pf_t0) | (pf_dry_iters << 40);
        unsigned long long *w = p.timeline + 8ull * p.n_waves + 8ull * wave_id;
        w[0] = pf_wall0;
        w[1] = pf_wall_dry;
        w[2] = wall_clock64();
        w[3] = pf_ld;
        w[4] = pf_lds;
    }
#endif
    wave_exit(p, lane, L.nseg, nsmp);
}

__global__ __launch_bounds__(kBlock, RV_BVH4_MIN_WAVES) void trace_bvh4(const FrameParams p) { bvh4_body<false, false, false>(p); }
#if RVPT_HIP_LAB
__global__ __launch_bounds__(kBlock, RV_BVH4Q_MIN_WAVES) void trace_bvh4q(const FrameParams p) { bvh4_body<false, false, true>(p); }
#endif
__global__ __launch_bounds__(kBlock, 1) void trace_bvh4_resident(const FrameParams p) { bvh4_body<true, false, false>(p); }
__global__ __launch_bounds__(kBlock, 1) void trace_bvh4_generic(const FrameParams p) { bvh4_body<false, true, false>(p); }
__global__ __launch_bounds__(kBlock, 1) void trace_bvh4_resident_generic(const FrameParams p) { bvh4_body<true, true, false>(p); }

}  // namespace rv

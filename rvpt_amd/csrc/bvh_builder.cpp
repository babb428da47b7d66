// bvh_builder.cpp — host-side binned-SAH BVH build producing the reference node layout.
//
// Role of BinnedBvhBuilder (reference src/rvpt/bvh_builder.{h,cpp}; run once at init,
// rvpt.cpp:83-86): 16 bins per axis, leaves of 2..8 primitives, median split when binning finds
// nothing better (bvh_builder.h:46-50).  Written from scratch: iterative (explicit work stack, no
// recursion), bins over the centroid bounds, and with a depth guard so that the tree always fits the
// traversal's 64-entry stack (intersection.glsl:363).  The kernel only depends on the node layout
// (bvh.h:12-19): root 0, sibling pairs adjacent, leaf iff primitive_count > 0.
//
// Large scenes are built by several host threads (the reference's builder is single-threaded and recursive): the
// top of the tree is split serially down to subtrees of bounded size, the subtrees are built concurrently — every
// node only reads and permutes its own index range, so a subtree's result does not depend on who builds it or
// when — and a final pass numbers the nodes in the order the serial algorithm allocates them.  The output is
// byte-identical for any thread count (tests/test_host_utils.py).
#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
#include <thread>
#include <vector>

#include "../../include/rvpt_hip.h"

namespace {

constexpr int kBins = 16;
constexpr uint32_t kMinLeaf = 2;  // below this a node is never split
constexpr uint32_t kMaxLeaf = 8;  // above this a node is always split
constexpr int kBalanceDepth = 30; // from this depth on only median splits (bounds the height)

struct Box {
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    float hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    void grow(const float p[3])
    {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], p[a]);
            hi[a] = std::max(hi[a], p[a]);
        }
    }
    void grow(const Box &b)
    {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], b.lo[a]);
            hi[a] = std::max(hi[a], b.hi[a]);
        }
    }
    float half_area() const
    {
        const float dx = std::max(hi[0] - lo[0], 0.0f), dy = std::max(hi[1] - lo[1], 0.0f), dz = std::max(hi[2] - lo[2], 0.0f);
        return dx * (dy + dz) + dy * dz;
    }
};

struct Job {
    uint32_t node;
    int depth;
};

void store_bounds(rvpt_bvh_node &n, const Box &b)
{
    n.bounds[0] = b.lo[0];
    n.bounds[1] = b.hi[0];
    n.bounds[2] = b.lo[1];
    n.bounds[3] = b.hi[1];
    n.bounds[4] = b.lo[2];
    n.bounds[5] = b.hi[2];
}

struct BuildInput {
    const Box *boxes;
    const float *cent;
    uint32_t *idx;
    float traversal_cost;
    unsigned threads;  // helpers for the per-node passes of very large nodes (top of the tree)
};

constexpr uint32_t kParallelNode = 1u << 16;  // nodes with at least this many primitives share their passes over threads

// fn(part, lo, hi) over [begin, end) cut into `parts` contiguous pieces, one thread each (part 0 on the caller)
template <typename F>
void for_parts(uint32_t begin, uint32_t end, unsigned parts, F fn)
{
    const uint32_t n = end - begin;
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < parts; ++t)
        pool.emplace_back([=]() { fn(t, begin + static_cast<uint32_t>(static_cast<uint64_t>(n) * t / parts), begin + static_cast<uint32_t>(static_cast<uint64_t>(n) * (t + 1) / parts)); });
    fn(0u, begin, begin + static_cast<uint32_t>(static_cast<uint64_t>(n) / parts));
    for (auto &th : pool) th.join();
}

struct Deferred {  // a node of the top tree whose subtree is built separately
    uint32_t node, begin, count;
    int depth;
};

// Builds the tree over idx[begin, begin+count) into `nodes` (node 0 = its root, children pairs appended in the order
// the work stack allocates them: the popped node's pair, left child processed first).  Nodes with at most
// `defer_below` primitives at depth > 0 are not split but reported in `deferred` (0 = build everything).
void build_tree(const BuildInput &in, uint32_t begin, uint32_t count, int depth0, uint32_t defer_below, std::vector<rvpt_bvh_node> &nodes,
                std::vector<Deferred> *deferred)
{
    const Box *boxes = in.boxes;
    const float *cent = in.cent;
    uint32_t *idx = in.idx;
    nodes.clear();
    nodes.emplace_back();
    nodes[0].first_child_or_primitive = begin;
    nodes[0].primitive_count = count;
    std::vector<Job> work;
    work.push_back({0u, depth0});

    while (!work.empty()) {
        const Job job = work.back();
        work.pop_back();
        const uint32_t begin_n = nodes[job.node].first_child_or_primitive, count_n = nodes[job.node].primitive_count, end = begin_n + count_n;
        if (deferred && job.node != 0 && count_n <= defer_below && count_n >= kMinLeaf) {
            deferred->push_back({job.node, begin_n, count_n, job.depth});
            continue;
        }

        // min / max / counts are order-independent, so sharing a pass over threads changes nothing in the result
        const unsigned parts = (in.threads > 1 && count_n >= kParallelNode) ? in.threads : 1u;
        Box bounds, cbounds;
        if (parts > 1) {
            std::vector<Box> pb(parts), pc(parts);
            for_parts(begin_n, end, parts, [&](unsigned t, uint32_t lo, uint32_t hi) {
                Box b, c;
                for (uint32_t i = lo; i < hi; ++i) {
                    b.grow(boxes[idx[i]]);
                    c.grow(&cent[3 * idx[i]]);
                }
                pb[t] = b;
                pc[t] = c;
            });
            for (unsigned t = 0; t < parts; ++t) {
                bounds.grow(pb[t]);
                cbounds.grow(pc[t]);
            }
        } else {
            for (uint32_t i = begin_n; i < end; ++i) {
                bounds.grow(boxes[idx[i]]);
                cbounds.grow(&cent[3 * idx[i]]);
            }
        }
        store_bounds(nodes[job.node], bounds);
        if (count_n < kMinLeaf) continue;

        // --- binned SAH over the centroid bounds ---------------------------------------------
        float best_cost = FLT_MAX;
        int best_axis = -1, best_bin = 0;
        if (job.depth < kBalanceDepth) {
            for (int axis = 0; axis < 3; ++axis) {
                const float extent = cbounds.hi[axis] - cbounds.lo[axis];
                if (!(extent > 0.0f)) continue;
                const float scale = static_cast<float>(kBins) / extent;
                Box bin_box[kBins];
                uint32_t bin_cnt[kBins] = {};
                auto fill = [&](uint32_t lo, uint32_t hi, Box *bb, uint32_t *bc) {
                    for (uint32_t i = lo; i < hi; ++i) {
                        const int b = std::min(kBins - 1, std::max(0, static_cast<int>((cent[3 * idx[i] + axis] - cbounds.lo[axis]) * scale)));
                        bb[b].grow(boxes[idx[i]]);
                        bc[b] += 1;
                    }
                };
                if (parts > 1) {
                    std::vector<Box> pb(static_cast<size_t>(parts) * kBins);
                    std::vector<uint32_t> pn(static_cast<size_t>(parts) * kBins, 0u);
                    for_parts(begin_n, end, parts, [&](unsigned t, uint32_t lo, uint32_t hi) { fill(lo, hi, &pb[static_cast<size_t>(t) * kBins], &pn[static_cast<size_t>(t) * kBins]); });
                    for (unsigned t = 0; t < parts; ++t)
                        for (int b = 0; b < kBins; ++b) {
                            bin_box[b].grow(pb[static_cast<size_t>(t) * kBins + b]);
                            bin_cnt[b] += pn[static_cast<size_t>(t) * kBins + b];
                        }
                } else {
                    fill(begin_n, end, bin_box, bin_cnt);
                }
                float right_cost[kBins];
                Box acc;
                uint32_t cnt = 0;
                for (int b = kBins - 1; b > 0; --b) {
                    acc.grow(bin_box[b]);
                    cnt += bin_cnt[b];
                    right_cost[b] = cnt ? acc.half_area() * static_cast<float>(cnt) : FLT_MAX;
                }
                acc = Box();
                cnt = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    acc.grow(bin_box[b]);
                    cnt += bin_cnt[b];
                    if (cnt == 0 || right_cost[b + 1] == FLT_MAX) continue;
                    const float cost = acc.half_area() * static_cast<float>(cnt) + right_cost[b + 1];
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_axis = axis;
                        best_bin = b + 1;  // primitives with bin < best_bin go left
                    }
                }
            }
        }
        // leaf cost in the same units as the split cost (area x primitives) plus what descending one level costs:
        // traversal_cost node visits' worth of triangle tests spread over the parent's area
        const float leaf_cost = bounds.half_area() * (static_cast<float>(count_n) - in.traversal_cost);
        uint32_t mid = 0;
        if (best_axis >= 0 && best_cost < leaf_cost) {
            const float lo = cbounds.lo[best_axis];
            const float scale = static_cast<float>(kBins) / (cbounds.hi[best_axis] - lo);
            uint32_t *m = std::partition(idx + begin_n, idx + end, [&](uint32_t i) {
                const int b = std::min(kBins - 1, std::max(0, static_cast<int>((cent[3 * i + best_axis] - lo) * scale)));
                return b < best_bin;
            });
            mid = static_cast<uint32_t>(m - idx);
        } else if (count_n <= kMaxLeaf) {
            continue;  // cheap enough as a leaf
        }
        if (mid <= begin_n || mid >= end) {
            // median split along the widest centroid axis
            int axis = 0;
            float widest = -1.0f;
            for (int a = 0; a < 3; ++a) {
                const float e = cbounds.hi[a] - cbounds.lo[a];
                if (e > widest) {
                    widest = e;
                    axis = a;
                }
            }
            mid = begin_n + count_n / 2;
            // strict weak order also for NaN centroids (a NaN vertex): NaN sorts after every number, ties by index
            std::nth_element(idx + begin_n, idx + mid, idx + end, [&](uint32_t i, uint32_t j) {
                const float ci = cent[3 * i + axis], cj = cent[3 * j + axis];
                const bool ni = ci != ci, nj = cj != cj;
                if (ni || nj) return ni == nj ? i < j : nj;
                return ci < cj || (ci == cj && i < j);
            });
        }
        const uint32_t left = static_cast<uint32_t>(nodes.size());
        nodes.emplace_back();
        nodes.emplace_back();
        nodes[left].first_child_or_primitive = begin_n;
        nodes[left].primitive_count = mid - begin_n;
        nodes[left + 1].first_child_or_primitive = mid;
        nodes[left + 1].primitive_count = end - mid;
        nodes[job.node].first_child_or_primitive = left;
        nodes[job.node].primitive_count = 0;
        work.push_back({left + 1, job.depth + 1});
        work.push_back({left, job.depth + 1});
    }
}

}  // namespace

extern "C" int rvpt_bvh_build(const rvpt_triangle *tris, size_t n_tris, rvpt_bvh_node *nodes_out, size_t *n_nodes_out,
                              uint32_t *prim_indices_out)
{
    if (!tris || !nodes_out || !n_nodes_out || !prim_indices_out || n_tris == 0 || n_tris > 0x3FFFFFFFull) return RVPT_HIP_ERR_INVALID;
    const uint32_t n = static_cast<uint32_t>(n_tris);
    try {
        std::vector<Box> boxes(n);
        std::vector<float> cent(static_cast<size_t>(n) * 3);
        uint32_t *idx = prim_indices_out;
        auto prepare = [&](unsigned, uint32_t lo, uint32_t hi) {
            for (uint32_t i = lo; i < hi; ++i) {
                boxes[i].grow(tris[i].vert0);
                boxes[i].grow(tris[i].vert1);
                boxes[i].grow(tris[i].vert2);
                for (int a = 0; a < 3; ++a) cent[3 * i + a] = (tris[i].vert0[a] + tris[i].vert1[a] + tris[i].vert2[a]) * (1.0f / 3.0f);
                idx[i] = i;
            }
        };
        BuildInput in{boxes.data(), cent.data(), idx, 0.0f, 1u};
        // relative cost of visiting an inner node, in triangle tests (0 = the reference's pure area x count rule,
        // bvh_builder.cpp:148, which splits down to 1-2 primitives per leaf); a tuning knob for experiments
        if (const char *e = getenv("RVPT_BVH_TRAVERSAL_COST")) in.traversal_cost = static_cast<float>(atof(e));
        unsigned threads = std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        if (const char *e = getenv("RVPT_BVH_THREADS")) threads = static_cast<unsigned>(std::max(1, std::min(64, atoi(e))));
        if (n < 32768u) threads = 1;
        in.threads = threads;
        for_parts(0u, n, threads, prepare);

        std::vector<rvpt_bvh_node> top;
        std::vector<Deferred> deferred;
        std::vector<std::vector<rvpt_bvh_node>> sub;
        if (threads == 1) {
            build_tree(in, 0, n, 0, 0, top, nullptr);
        } else {
            build_tree(in, 0, n, 0, std::max(4096u, n / (8u * threads)), top, &deferred);
            sub.resize(deferred.size());
            BuildInput serial = in;
            serial.threads = 1;
            std::vector<std::thread> pool;
            std::vector<int> failed(threads, 0);
            for (unsigned t = 0; t < threads; ++t)
                pool.emplace_back([&, t]() {
                    try {
                        for (size_t k = t; k < deferred.size(); k += threads)  // (sizes are similar: split down to a common bound)
                            build_tree(serial, deferred[k].begin, deferred[k].count, deferred[k].depth, 0, sub[k], nullptr);
                    } catch (const std::bad_alloc &) {
                        failed[t] = 1;
                    }
                });
            for (auto &th : pool) th.join();
            for (int f : failed)
                if (f) return RVPT_HIP_ERR_HIP;
        }
        // number the nodes the way the single work stack allocates them: a popped inner node takes the next pair, its
        // left child is popped next.  Reference = (tree, local index); tree -1 is the top tree.
        std::vector<int> deferred_of(top.size(), -1);
        for (size_t k = 0; k < deferred.size(); ++k) deferred_of[deferred[k].node] = static_cast<int>(k);
        struct Ref {
            int tree;
            uint32_t local, out;
        };
        std::vector<Ref> stack;
        stack.push_back({-1, 0u, 0u});
        size_t n_nodes = 1;
        while (!stack.empty()) {
            Ref r = stack.back();
            stack.pop_back();
            if (r.tree < 0 && deferred_of[r.local] >= 0) r = {deferred_of[r.local], 0u, r.out};
            const rvpt_bvh_node &src = r.tree < 0 ? top[r.local] : sub[static_cast<size_t>(r.tree)][r.local];
            rvpt_bvh_node &dst = nodes_out[r.out];
            dst = src;
            if (src.primitive_count == 0) {
                const uint32_t pair = static_cast<uint32_t>(n_nodes);
                n_nodes += 2;
                dst.first_child_or_primitive = pair;
                stack.push_back({r.tree, src.first_child_or_primitive + 1u, pair + 1u});
                stack.push_back({r.tree, src.first_child_or_primitive, pair});
            }
        }
        *n_nodes_out = n_nodes;
    } catch (const std::bad_alloc &) {
        return RVPT_HIP_ERR_HIP;
    }
    return RVPT_HIP_OK;
}

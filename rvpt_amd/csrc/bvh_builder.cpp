// bvh_builder.cpp — host-side binned-SAH BVH build producing the reference node layout.
//
// Role of BinnedBvhBuilder (reference src/rvpt/bvh_builder.{h,cpp}; run once at init,
// rvpt.cpp:83-86): 16 bins per axis, leaves of 2..8 primitives, median split when binning finds
// nothing better (bvh_builder.h:46-50).  Written from scratch: iterative (explicit work stack, no
// recursion), bins over the centroid bounds, and with a depth guard so that the tree always fits the
// traversal's 64-entry stack (intersection.glsl:363).  The kernel only depends on the node layout
// (bvh.h:12-19): root 0, sibling pairs adjacent, leaf iff primitive_count > 0.
#include <algorithm>
#include <cfloat>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <numeric>
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

}  // namespace

extern "C" int rvpt_bvh_build(const rvpt_triangle *tris, size_t n_tris, rvpt_bvh_node *nodes_out, size_t *n_nodes_out,
                              uint32_t *prim_indices_out)
{
    if (!tris || !nodes_out || !n_nodes_out || !prim_indices_out || n_tris == 0 || n_tris > 0x3FFFFFFFull) return RVPT_HIP_ERR_INVALID;
    const uint32_t n = static_cast<uint32_t>(n_tris);
    std::vector<Box> boxes;
    std::vector<float> cent;
    try {
        boxes.resize(n);
        cent.resize(static_cast<size_t>(n) * 3);
    } catch (const std::bad_alloc &) {
        return RVPT_HIP_ERR_HIP;
    }
    for (uint32_t i = 0; i < n; ++i) {
        boxes[i].grow(tris[i].vert0);
        boxes[i].grow(tris[i].vert1);
        boxes[i].grow(tris[i].vert2);
        for (int a = 0; a < 3; ++a) cent[3 * i + a] = (tris[i].vert0[a] + tris[i].vert1[a] + tris[i].vert2[a]) * (1.0f / 3.0f);
    }
    uint32_t *idx = prim_indices_out;
    std::iota(idx, idx + n, 0u);

    size_t n_nodes = 1;
    nodes_out[0].first_child_or_primitive = 0;
    nodes_out[0].primitive_count = n;
    // relative cost of visiting an inner node, in triangle tests (0 = the reference's pure area x count rule,
    // bvh_builder.cpp:148, which splits down to 1-2 primitives per leaf); a tuning knob for experiments
    float traversal_cost = 0.0f;
    if (const char *e = getenv("RVPT_BVH_TRAVERSAL_COST")) traversal_cost = static_cast<float>(atof(e));
    std::vector<Job> work;
    work.push_back({0u, 0});

    while (!work.empty()) {
        const Job job = work.back();
        work.pop_back();
        rvpt_bvh_node &node = nodes_out[job.node];
        const uint32_t begin = node.first_child_or_primitive, count = node.primitive_count, end = begin + count;

        Box bounds, cbounds;
        for (uint32_t i = begin; i < end; ++i) {
            bounds.grow(boxes[idx[i]]);
            cbounds.grow(&cent[3 * idx[i]]);
        }
        store_bounds(node, bounds);
        if (count < kMinLeaf) continue;

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
                for (uint32_t i = begin; i < end; ++i) {
                    const int b = std::min(kBins - 1, std::max(0, static_cast<int>((cent[3 * idx[i] + axis] - cbounds.lo[axis]) * scale)));
                    bin_box[b].grow(boxes[idx[i]]);
                    bin_cnt[b] += 1;
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
        // kTraversalCost node visits' worth of triangle tests spread over the parent's area
        const float leaf_cost = bounds.half_area() * (static_cast<float>(count) - traversal_cost);
        uint32_t mid = 0;
        if (best_axis >= 0 && best_cost < leaf_cost) {
            const float lo = cbounds.lo[best_axis];
            const float scale = static_cast<float>(kBins) / (cbounds.hi[best_axis] - lo);
            uint32_t *m = std::partition(idx + begin, idx + end, [&](uint32_t i) {
                const int b = std::min(kBins - 1, std::max(0, static_cast<int>((cent[3 * i + best_axis] - lo) * scale)));
                return b < best_bin;
            });
            mid = static_cast<uint32_t>(m - idx);
        } else if (count <= kMaxLeaf) {
            continue;  // cheap enough as a leaf
        }
        if (mid <= begin || mid >= end) {
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
            mid = begin + count / 2;
            std::nth_element(idx + begin, idx + mid, idx + end, [&](uint32_t i, uint32_t j) {
                const float ci = cent[3 * i + axis], cj = cent[3 * j + axis];
                return ci < cj || (ci == cj && i < j);
            });
        }
        const uint32_t left = static_cast<uint32_t>(n_nodes);
        n_nodes += 2;
        nodes_out[left].first_child_or_primitive = begin;
        nodes_out[left].primitive_count = mid - begin;
        nodes_out[left + 1].first_child_or_primitive = mid;
        nodes_out[left + 1].primitive_count = end - mid;
        node.first_child_or_primitive = left;
        node.primitive_count = 0;
        work.push_back({left + 1, job.depth + 1});
        work.push_back({left, job.depth + 1});
    }
    *n_nodes_out = n_nodes;
    return RVPT_HIP_OK;
}

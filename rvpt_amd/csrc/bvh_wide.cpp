// bvh_wide.cpp — host side of rvpt_bvh4.hip: the 4-wide regrouping of a binary BVH in the reference node layout (children of an inner node at
// first, first + 1; leaf iff primitive_count > 0).  Called by rvpt_hip_upload_scene on the breadth-first device copy of the caller's tree, and
// exported as rvpt_bvh_wide_form so that hosts and tests can look at what the kernel walks.  No GPU needed.
#include "bvh_wide.h"

#include <algorithm>
#include <array>
#include <cstring>

namespace rv {

// The reference walks its binary tree depth first, left child first, and tests a node's box when it visits the node, with the ray's closest_t of that
// moment (intersection.glsl:361-413).  When every box of the tree CONTAINS the boxes of its two children (float comparisons; true of any tree
// built bottom-up from min/max of child bounds, as the reference's builder and ours do) the slab test is monotone under containment — (b - o) * inv and
// the min/max chain of intersect_aabb are monotone in b, whatever the rounding — so a child that passes implies its parent passed at the same
// closest_t, and a node is visited by the reference IFF ITS OWN BOX passes at the moment the depth-first order reaches it.  Inner nodes are then only
// an acceleration, and any regrouping that keeps the depth-first order of the nodes it keeps visits the same leaves, tests the same triangles in the
// same order and finds the same closest_t and hit, bit for bit.  build_wide_nodes regroups: a wide node = a binary inner node whose child list
// [left, right] has had inner children replaced, in place, by THEIR two children (largest box first) until it holds four — only across nodes that do
// contain their children; a node that does not keeps its own slot and is tested itself, so caller trees with loose boxes stay exact, just less wide.
// Device layout: 8 quads (128 B) per wide node — minx[4] maxx[4] miny[4] maxy[4] minz[4] maxz[4] head[4] pad — breadth first (upper levels first:
// the kernel keeps the first nodes in LDS); head = first | count << head_shift for a leaf (count > 0), the wide index of an inner child (count 0),
// kWideEmpty for an unused slot.  Returns the wide nodes (empty: no wide form — single-leaf tree, heads that do not pack) and the stack need.
std::vector<float> build_wide_nodes(const rvpt_bvh_node *nodes, size_t n_nodes, uint32_t head_shift, uint32_t &stack_need)
{
    stack_need = 0;
    std::vector<float> out;
    if (nodes == nullptr || n_nodes == 0 || nodes[0].primitive_count > 0 || head_shift == 0) return out;
    auto contains = [&](const rvpt_bvh_node &a, const rvpt_bvh_node &b) {  // a's box contains b's (bounds = minx maxx miny maxy minz maxz)
        for (int ax = 0; ax < 3; ++ax)
            if (!(b.bounds[2 * ax] >= a.bounds[2 * ax] && b.bounds[2 * ax + 1] <= a.bounds[2 * ax + 1])) return false;
        return true;
    };
    auto area = [&](const rvpt_bvh_node &n) {
        const double dx = double(n.bounds[1]) - n.bounds[0], dy = double(n.bounds[3]) - n.bounds[2], dz = double(n.bounds[5]) - n.bounds[4];
        return dx * dy + dy * dz + dz * dx;
    };
    std::vector<uint32_t> queue{0u};  // binary inner nodes that become wide nodes, in wide-index order (breadth first)
    std::vector<std::array<uint32_t, 4>> kids;  // per wide node: binary indices of its children, 0xFFFFFFFF = unused
    for (size_t head = 0; head < queue.size(); ++head) {
        const rvpt_bvh_node &b = nodes[queue[head]];
        std::vector<uint32_t> c{b.first_child_or_primitive, b.first_child_or_primitive + 1u};
        for (;;) {
            if (c.size() >= kWideFormChildren) break;
            int pick = -1;
            double best = -1.0;
            for (size_t i = 0; i < c.size(); ++i) {
                const rvpt_bvh_node &n = nodes[c[i]];
                if (n.primitive_count > 0) continue;
                const rvpt_bvh_node &l = nodes[n.first_child_or_primitive], &r = nodes[n.first_child_or_primitive + 1u];
                if (!contains(n, l) || !contains(n, r)) continue;  // this box must be tested itself
                if (area(n) > best) best = area(n), pick = static_cast<int>(i);
            }
            if (pick < 0) break;
            const uint32_t f = nodes[c[pick]].first_child_or_primitive;
            c[pick] = f;
            c.insert(c.begin() + pick + 1, f + 1u);
        }
        std::array<uint32_t, 4> k{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        for (size_t i = 0; i < c.size(); ++i) {
            k[i] = c[i];
            if (nodes[c[i]].primitive_count == 0) queue.push_back(c[i]);
        }
        kids.push_back(k);
    }
    // wide index of a binary inner node = its position in `queue`
    std::vector<uint32_t> wide_of(n_nodes, 0xFFFFFFFFu);
    for (size_t i = 0; i < queue.size(); ++i) wide_of[queue[i]] = static_cast<uint32_t>(i);
    out.assign(queue.size() * 32, 0.0f);
    for (size_t w = 0; w < queue.size(); ++w) {
        float *q = out.data() + w * 32;
        uint32_t *heads = reinterpret_cast<uint32_t *>(q + 24);
        for (int i = 0; i < 4; ++i) {
            heads[i] = kWideFormEmpty;
            if (kids[w][i] == 0xFFFFFFFFu) continue;
            const rvpt_bvh_node &n = nodes[kids[w][i]];
            for (int b6 = 0; b6 < 6; ++b6) q[4 * b6 + i] = n.bounds[b6];
            const uint32_t hd = n.primitive_count > 0 ? (n.first_child_or_primitive | (n.primitive_count << head_shift)) : wide_of[kids[w][i]];
            if (hd == kWideFormEmpty) return std::vector<float>();  // (cannot happen below 2^31 nodes; the marker must stay unambiguous)
            heads[i] = hd;
        }
    }
    // stack need: a walk that descends into child i of a node leaves up to (children - 1 - i) siblings stacked
    std::vector<uint32_t> need(queue.size(), 0);
    for (size_t w = queue.size(); w-- > 0;) {
        uint32_t n_children = 0;
        for (int i = 0; i < 4; ++i) n_children += kids[w][i] != 0xFFFFFFFFu;
        uint32_t worst = 0;
        for (uint32_t i = 0; i < n_children; ++i) {
            const rvpt_bvh_node &n = nodes[kids[w][i]];
            const uint32_t below = n.primitive_count > 0 ? 0u : need[wide_of[kids[w][i]]];
            worst = std::max(worst, (n_children - 1u - i) + below);
        }
        need[w] = worst;
    }
    stack_need = std::max(1u, need[0]);
    return out;
}

}  // namespace rv

extern "C" int rvpt_bvh_wide_form(const rvpt_bvh_node *nodes, size_t n_nodes, uint32_t head_shift, float *wide_out, size_t wide_capacity, size_t *n_wide_out,
                                  uint32_t *stack_need_out)
{
    if (!nodes || n_nodes == 0 || !n_wide_out) return RVPT_HIP_ERR_INVALID;
    for (size_t i = 0; i < n_nodes; ++i)  // the same range checks upload_scene makes: children inside the array
        if (nodes[i].primitive_count == 0 && static_cast<uint64_t>(nodes[i].first_child_or_primitive) + 1 >= n_nodes) return RVPT_HIP_ERR_INVALID;
    uint32_t need = 0;
    const std::vector<float> wide = rv::build_wide_nodes(nodes, n_nodes, head_shift, need);
    *n_wide_out = wide.size() / 32;
    if (stack_need_out) *stack_need_out = need;
    if (wide.empty()) return RVPT_HIP_OK;  // no wide form (single-leaf tree, heads that do not pack): the binary walk serves it
    if (!wide_out || wide_capacity < wide.size() / 32) return RVPT_HIP_ERR_SIZE;
    std::memcpy(wide_out, wide.data(), wide.size() * sizeof(float));
    return RVPT_HIP_OK;
}

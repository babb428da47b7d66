// bvh_wide.cpp — host side of rvpt_bvh4.hip / rvpt_bvh8.hip: the WIDE regrouping (4 or 8 children per node) of a binary BVH in the reference node layout
// (children of an inner node at first, first + 1; leaf iff primitive_count > 0).  Called by rvpt_hip_upload_scene on the breadth-first device copy of the
// caller's tree, and exported as rvpt_bvh_wide_form so that hosts and tests can look at what the kernels walk.  No GPU needed.
#include "bvh_wide.h"

#include <algorithm>
#include <array>
#include <cstring>

namespace rv {

namespace {

// The reference walks its binary tree depth first, left child first, and tests a node's box when it visits the node, with the ray's closest_t of that
// moment (intersection.glsl:361-413).  When every box of the tree CONTAINS the boxes of its two children (float comparisons; true of any tree
// built bottom-up from min/max of child bounds, as the reference's builder and ours do) the slab test is monotone under containment — (b - o) * inv and
// the min/max chain of intersect_aabb are monotone in b, whatever the rounding — so a child that passes implies its parent passed at the same
// closest_t, and a node is visited by the reference IFF ITS OWN BOX passes at the moment the depth-first order reaches it.  Inner nodes are then only
// an acceleration, and any regrouping that keeps the depth-first order of the nodes it keeps visits the same leaves, tests the same triangles in the
// same order and finds the same closest_t and hit, bit for bit.  regroup() does that: a wide node = a binary inner node whose child list
// [left, right] has had inner children replaced, in place, by THEIR two children (largest box first) until it holds `width` — only across nodes that do
// contain their children; a node that does not keeps its own slot and is tested itself, so caller trees with loose boxes stay exact, just less wide.
// (The NaN corner — ADVICE r4: a ray with a zero direction component whose origin lies exactly ON a face of a box gives (b - o) * inv = 0 * inf = NaN
// for that slab, which minNum / maxNum drop; a parent can then fail where a FLAT descendant (min == max == o on that axis) passes, and the wide walk
// visits a leaf the binary walk skips.  Such a leaf's triangles lie in the plane min == max, the ray runs inside that plane, the triangle test's
// denominator dot(d, n) is 0 and its quotient NaN: nothing is accepted, closest_t and the hit do not change.  The IMAGE claim holds; node-visit counters are
// not exact in that corner.)
struct Regrouped {
    std::vector<uint32_t> queue;                 // binary inner nodes that become wide nodes, in wide-index order (breadth first)
    std::vector<std::array<uint32_t, 8>> kids;   // per wide node: binary indices of its children, 0xFFFFFFFF = unused
    std::vector<uint32_t> wide_of;               // binary inner node -> wide index
    uint32_t stack_need = 0;
    bool ok = false;
};

Regrouped regroup(const rvpt_bvh_node *nodes, size_t n_nodes, uint32_t head_shift, uint32_t width)
{
    Regrouped r;
    if (nodes == nullptr || n_nodes == 0 || nodes[0].primitive_count > 0 || head_shift == 0 || head_shift >= 32 || (width != 4 && width != 8)) return r;
    const uint64_t index_limit = 1ull << head_shift, count_limit = 1ull << (32 - head_shift);
    // every head must pack: first | count << shift for a leaf; and the input must be a TREE (every node reachable at most once: ADVICE r4 — the exported
    // entry point takes any array)
    for (size_t i = 0; i < n_nodes; ++i) {
        const rvpt_bvh_node &n = nodes[i];
        if (n.primitive_count > 0) {
            if (n.first_child_or_primitive >= index_limit || n.primitive_count >= count_limit) return r;
        } else if (static_cast<uint64_t>(n.first_child_or_primitive) + 1 >= n_nodes) {
            return r;
        }
    }
    auto contains = [&](const rvpt_bvh_node &a, const rvpt_bvh_node &b) {  // a's box contains b's (bounds = minx maxx miny maxy minz maxz)
        for (int ax = 0; ax < 3; ++ax)
            if (!(b.bounds[2 * ax] >= a.bounds[2 * ax] && b.bounds[2 * ax + 1] <= a.bounds[2 * ax + 1])) return false;
        return true;
    };
    auto area = [&](const rvpt_bvh_node &n) {
        const double dx = double(n.bounds[1]) - n.bounds[0], dy = double(n.bounds[3]) - n.bounds[2], dz = double(n.bounds[5]) - n.bounds[4];
        return dx * dy + dy * dz + dz * dx;
    };
    std::vector<uint8_t> seen(n_nodes, 0);
    seen[0] = 1;
    r.queue.push_back(0u);
    for (size_t head = 0; head < r.queue.size(); ++head) {
        const rvpt_bvh_node &b = nodes[r.queue[head]];
        std::vector<uint32_t> c{b.first_child_or_primitive, b.first_child_or_primitive + 1u};
        for (uint32_t x : c) {
            if (seen[x]) return Regrouped();  // not a tree
            seen[x] = 1;
        }
        for (;;) {
            if (c.size() >= width) break;
            int pick = -1;
            double best = -1.0;
            for (size_t i = 0; i < c.size(); ++i) {
                const rvpt_bvh_node &n = nodes[c[i]];
                if (n.primitive_count > 0) continue;
                const rvpt_bvh_node &l = nodes[n.first_child_or_primitive], &rr = nodes[n.first_child_or_primitive + 1u];
                if (!contains(n, l) || !contains(n, rr)) continue;  // this box must be tested itself
                if (area(n) > best) best = area(n), pick = static_cast<int>(i);
            }
            if (pick < 0) break;
            const uint32_t f = nodes[c[pick]].first_child_or_primitive;
            if (seen[f] || seen[f + 1u]) return Regrouped();
            seen[f] = seen[f + 1u] = 1;
            c[pick] = f;
            c.insert(c.begin() + pick + 1, f + 1u);
        }
        std::array<uint32_t, 8> k;
        k.fill(0xFFFFFFFFu);
        for (size_t i = 0; i < c.size(); ++i) {
            k[i] = c[i];
            if (nodes[c[i]].primitive_count == 0) r.queue.push_back(c[i]);
        }
        r.kids.push_back(k);
    }
    if (r.queue.size() >= index_limit) return Regrouped();  // a wide index must fit below the count bits of a head
    r.wide_of.assign(n_nodes, 0xFFFFFFFFu);
    for (size_t i = 0; i < r.queue.size(); ++i) r.wide_of[r.queue[i]] = static_cast<uint32_t>(i);
    // stack need: a walk that descends into child i of a node leaves up to (children - 1 - i) siblings stacked
    std::vector<uint32_t> need(r.queue.size(), 0);
    for (size_t w = r.queue.size(); w-- > 0;) {
        uint32_t n_children = 0;
        for (uint32_t i = 0; i < width; ++i) n_children += r.kids[w][i] != 0xFFFFFFFFu;
        uint32_t worst = 0;
        for (uint32_t i = 0; i < n_children; ++i) {
            const rvpt_bvh_node &n = nodes[r.kids[w][i]];
            const uint32_t below = n.primitive_count > 0 ? 0u : need[r.wide_of[r.kids[w][i]]];
            worst = std::max(worst, (n_children - 1u - i) + below);
        }
        need[w] = worst;
    }
    r.stack_need = std::max(1u, need[0]);
    r.ok = true;
    return r;
}

uint32_t head_of(const Regrouped &r, const rvpt_bvh_node *nodes, uint32_t binary, uint32_t head_shift)
{
    const rvpt_bvh_node &n = nodes[binary];
    return n.primitive_count > 0 ? (n.first_child_or_primitive | (n.primitive_count << head_shift)) : r.wide_of[binary];
}

}  // namespace

// Device layout, width 4: 8 quads (128 B) per wide node — minx[4] maxx[4] miny[4] maxy[4] minz[4] maxz[4] head[4] pad — breadth first (upper levels
// first: the kernel keeps the first nodes in LDS); head = first | count << head_shift for a leaf (count > 0), the wide index of an inner child (count 0),
// kWideEmpty for an unused slot.  Returns the wide nodes (empty: no wide form — single-leaf tree, heads that do not pack, not a tree) and the stack need.
std::vector<float> build_wide_nodes(const rvpt_bvh_node *nodes, size_t n_nodes, uint32_t head_shift, uint32_t &stack_need)
{
    stack_need = 0;
    std::vector<float> out;
    const Regrouped r = regroup(nodes, n_nodes, head_shift, 4);
    if (!r.ok) return out;
    out.assign(r.queue.size() * 32, 0.0f);
    for (size_t w = 0; w < r.queue.size(); ++w) {
        float *q = out.data() + w * 32;
        uint32_t *heads = reinterpret_cast<uint32_t *>(q + 24);
        for (int i = 0; i < 4; ++i) {
            heads[i] = kWideFormEmpty;
            if (r.kids[w][i] == 0xFFFFFFFFu) continue;
            const rvpt_bvh_node &n = nodes[r.kids[w][i]];
            for (int b6 = 0; b6 < 6; ++b6) q[4 * b6 + i] = n.bounds[b6];
            const uint32_t hd = head_of(r, nodes, r.kids[w][i], head_shift);
            if (hd == kWideFormEmpty) return std::vector<float>();  // (cannot happen below 2^31 nodes; the marker must stay unambiguous)
            heads[i] = hd;
        }
    }
    stack_need = r.stack_need;
    return out;
}

// Width 8 (rvpt_bvh8.hip): 16 quads (256 B) per node — for each of the six bounds two quads (children 0-3, 4-7): minx maxx miny maxy minz maxz —, then
// head[8] (two quads) and two quads of padding; everything else as above.
std::vector<float> build_wide8_nodes(const rvpt_bvh_node *nodes, size_t n_nodes, uint32_t head_shift, uint32_t &stack_need)
{
    stack_need = 0;
    std::vector<float> out;
    const Regrouped r = regroup(nodes, n_nodes, head_shift, 8);
    if (!r.ok) return out;
    out.assign(r.queue.size() * 64, 0.0f);
    for (size_t w = 0; w < r.queue.size(); ++w) {
        float *q = out.data() + w * 64;
        uint32_t *heads = reinterpret_cast<uint32_t *>(q + 48);
        for (int i = 0; i < 8; ++i) {
            heads[i] = kWideFormEmpty;
            if (r.kids[w][i] == 0xFFFFFFFFu) continue;
            const rvpt_bvh_node &n = nodes[r.kids[w][i]];
            for (int b6 = 0; b6 < 6; ++b6) q[8 * b6 + i] = n.bounds[b6];
            const uint32_t hd = head_of(r, nodes, r.kids[w][i], head_shift);
            if (hd == kWideFormEmpty) return std::vector<float>();
            heads[i] = hd;
        }
    }
    stack_need = r.stack_need;
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
    if (wide.empty()) return RVPT_HIP_OK;  // no wide form (single-leaf tree, heads that do not pack, not a tree): the binary walk serves it
    if (!wide_out || wide_capacity < wide.size() / 32) return RVPT_HIP_ERR_SIZE;
    std::memcpy(wide_out, wide.data(), wide.size() * sizeof(float));
    return RVPT_HIP_OK;
}

// bvh_wide.cpp — host side of rvpt_bvh4.hip / rvpt_bvh8.hip: the WIDE regrouping (4 or 8 children per node) of a binary BVH in the reference node layout
// (children of an inner node at first, first + 1; leaf iff primitive_count > 0).  Called by rvpt_hip_upload_scene on the breadth-first device copy of the
// caller's tree, and exported as rvpt_bvh_wide_form so that hosts and tests can look at what the kernels walk.  No GPU needed.
#include "bvh_wide.h"

#include <algorithm>
#include <array>
#include <cmath>
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

// What trace_bvh4_fast needs beside the wide nodes (rvpt_device.h: fast_slab_*): its conservative child test may replace the reference's only where boxes
// ONLY CULL, i.e. in a tree EVERY inner node of which contains both its children (regroup() above keeps a node that does not as a child slot of its own,
// to be tested exactly — the fast walk has no exact test for an inner slot), and it tests a leaf's own box exactly at the visit, fetched by the leaf's
// first triangle: 8 floats at [8 first] = minx maxx miny maxy | minz maxz 0 0.  Empty (the exact kernel serves the tree) when a node does not contain a
// child, a bound is not finite, two leaves start at the same triangle, or a leaf runs past n_tris.  extent = the largest |bound| (the test's margin).
std::vector<float> build_leaf_boxes(const rvpt_bvh_node *nodes, size_t n_nodes, size_t n_tris, float &extent)
{
    extent = 0.0f;
    std::vector<float> out;
    if (nodes == nullptr || n_nodes == 0 || n_tris == 0) return out;
    std::vector<float> boxes(n_tris * 8, 0.0f);
    std::vector<uint8_t> taken(n_tris, 0), seen(n_nodes, 0);
    float ext = 0.0f;
    std::vector<uint32_t> todo{0u};  // the nodes a walk can reach (the device copy of a tree has an unused slot 1; a caller's array may hold anything else)
    seen[0] = 1;
    while (!todo.empty()) {
        const rvpt_bvh_node &n = nodes[todo.back()];
        todo.pop_back();
        for (int b6 = 0; b6 < 6; ++b6) {
            if (!(std::fabs(n.bounds[b6]) <= 3.0e38f)) return out;  // NaN, inf
            ext = std::max(ext, std::fabs(n.bounds[b6]));
        }
        if (n.primitive_count > 0) {
            const uint64_t first = n.first_child_or_primitive;
            if (first + n.primitive_count > n_tris || taken[first]) return out;
            taken[first] = 1;
            float *q = boxes.data() + 8 * first;
            for (int b6 = 0; b6 < 6; ++b6) q[b6] = n.bounds[b6];
        } else {
            const uint64_t f = n.first_child_or_primitive;
            if (f + 1 >= n_nodes) return out;
            for (uint64_t c = f; c <= f + 1; ++c) {
                if (seen[c]) return out;  // not a tree
                seen[c] = 1;
                for (int ax = 0; ax < 3; ++ax)
                    if (!(nodes[c].bounds[2 * ax] >= n.bounds[2 * ax] && nodes[c].bounds[2 * ax + 1] <= n.bounds[2 * ax + 1])) return out;
                todo.push_back(static_cast<uint32_t>(c));
            }
        }
    }
    if (!(ext >= 1.0e-30f)) return out;  // (a tree of subnormal size: the margin's relative-error argument does not cover it)
    extent = ext;
    return boxes;
}

// The 64-BYTE form of a wide node (trace_bvh4q): the child boxes as 8-bit offsets from the node's own corner, rounded OUTWARD —
//   words 0-2  origin x y z (float: the smallest child minimum per axis)      words 3-5  scale x y z (float, a power of two)
//   words 6-11 qminx qmaxx qminy qmaxy qminz qmaxz, byte k = child k: the box [origin + qmin scale, origin + qmax scale] CONTAINS the child's exact box
//   words 12-15 the four heads of the 128-byte form
// Four 16-byte loads per step instead of seven.  Under containment inner boxes only cull (regroup() above), so a conservative box is as good as the exact one
// for deciding where to walk; a leaf's own exact box is tested at its visit (build_leaf_boxes).  Every q is verified here, in double, against the exact
// bound it replaces (origin + q scale is exact in double: a 24-bit and an 8-bit significand, the scale kept within 2^-28 of the corner's magnitude).  extent = the largest |coordinate| any dequantised bound or origin can take (the margin of the
// kernel's conservative test, rvpt_device.h: quant_slab_setup).  Empty: no quantised form.
std::vector<uint32_t> build_quant_nodes(const std::vector<float> &wide, float &extent)
{
    extent = 0.0f;
    std::vector<uint32_t> out;
    const size_t n = wide.size() / 32;
    if (n == 0) return out;
    out.assign(n * 16, 0u);
    double ext = 0.0;
    for (size_t w = 0; w < n; ++w) {
        const float *q = wide.data() + w * 32;
        const uint32_t *heads = reinterpret_cast<const uint32_t *>(q + 24);
        uint32_t *o = out.data() + w * 16;
        uint32_t words[6] = {0, 0, 0, 0, 0, 0};
        for (int ax = 0; ax < 3; ++ax) {
            double lo = 1e300, hi = -1e300;
            for (int k = 0; k < 4; ++k) {
                if (heads[k] == kWideFormEmpty) continue;
                lo = std::min(lo, double(q[4 * (2 * ax) + k]));
                hi = std::max(hi, double(q[4 * (2 * ax + 1) + k]));
            }
            if (!(lo <= hi) || !(std::fabs(lo) <= 1e37) || !(std::fabs(hi) <= 1e37)) return std::vector<uint32_t>();
            const float origin = static_cast<float>(lo);  // (a child minimum: exactly a float)
            int e = -100;                                  // scale = 2^e: the smallest power of two that puts hi within 255 steps of origin
            if (hi > lo) {
                int ex;
                (void)std::frexp((hi - lo) / 255.0, &ex);  // (hi - lo) / 255 = f 2^ex, 0.5 <= f < 1: 2^ex >= it
                e = std::max(ex, -100);
            }
            // ... and no finer than 2^-28 of |origin|: origin + q scale must be exact in double (and a scale far below the corner's own spacing says nothing).
            // A flat node (hi == lo: two coplanar triangles of an axis-aligned wall) gets the scale of its corner; its q are 0 either way.
            if (lo != 0.0) e = std::max(e, std::ilogb(std::fabs(lo)) - 28);
            uint32_t wmin = 0, wmax = 0;
            for (;; ++e) {
                if (e > 60) return std::vector<uint32_t>();  // (rvpt_device.h: scale |inv| must not overflow for |inv| <= 2^60)
                const double sc = std::ldexp(1.0, e);
                bool fits = true;
                wmin = wmax = 0;
                for (int k = 0; k < 4 && fits; ++k) {
                    uint32_t a = 255u, b = 0u;  // an unused slot: an empty interval
                    if (heads[k] != kWideFormEmpty) {
                        const double bmin = q[4 * (2 * ax) + k], bmax = q[4 * (2 * ax + 1) + k];
                        double fa = std::floor((bmin - lo) / sc), fb = std::ceil((bmax - lo) / sc);
                        while (fa > 0.0 && lo + fa * sc > bmin) fa -= 1.0;
                        while (lo + fb * sc < bmax) fb += 1.0;
                        if (fa < 0.0) fa = 0.0;
                        if (fb > 255.0) { fits = false; break; }
                        a = static_cast<uint32_t>(fa), b = static_cast<uint32_t>(fb);
                        if (!(lo + a * sc <= bmin && lo + b * sc >= bmax)) return std::vector<uint32_t>();  // (cannot happen)
                    }
                    wmin |= a << (8 * k);
                    wmax |= b << (8 * k);
                }
                if (fits) break;
            }
            const float scale = static_cast<float>(std::ldexp(1.0, e));
            std::memcpy(o + ax, &origin, 4);
            std::memcpy(o + 3 + ax, &scale, 4);
            words[2 * ax] = wmin;
            words[2 * ax + 1] = wmax;
            ext = std::max({ext, std::fabs(lo), std::fabs(lo + 255.0 * std::ldexp(1.0, e))});
        }
        for (int i = 0; i < 6; ++i) o[6 + i] = words[i];
        for (int k = 0; k < 4; ++k) o[12 + k] = heads[k];
    }
    if (!(ext >= 1.0e-30) || !(ext <= 1.0e37)) return std::vector<uint32_t>();
    extent = static_cast<float>(ext * 1.0000002);  // (rounded up)
    return out;
}

}  // namespace rv

#if RVPT_HIP_LAB  // include/rvpt_hip_lab.h: the device forms of the tree on the host, for the GPU-free tests
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

extern "C" int rvpt_bvh_quant_form(const rvpt_bvh_node *nodes, size_t n_nodes, uint32_t head_shift, size_t n_tris, uint32_t *quant_out, size_t quant_capacity,
                                   size_t *n_quant_out, float *leaf_boxes_out, float *extent_out)
{
    if (!nodes || n_nodes == 0 || !n_quant_out) return RVPT_HIP_ERR_INVALID;
    for (size_t i = 0; i < n_nodes; ++i)
        if (nodes[i].primitive_count == 0 && static_cast<uint64_t>(nodes[i].first_child_or_primitive) + 1 >= n_nodes) return RVPT_HIP_ERR_INVALID;
    *n_quant_out = 0;
    if (extent_out) *extent_out = 0.0f;
    uint32_t need = 0;
    const std::vector<float> wide = rv::build_wide_nodes(nodes, n_nodes, head_shift, need);
    if (wide.empty()) return RVPT_HIP_OK;
    float extent = 0.0f, box_extent = 0.0f;
    const std::vector<uint32_t> quant = rv::build_quant_nodes(wide, extent);
    if (quant.empty()) return RVPT_HIP_OK;
    const std::vector<float> boxes = rv::build_leaf_boxes(nodes, n_nodes, n_tris, box_extent);
    if (boxes.empty()) return RVPT_HIP_OK;  // a node that does not contain a child, two leaves on one triangle, ...: the exact nodes serve the tree
    if (!quant_out || quant_capacity < quant.size() / 16) return RVPT_HIP_ERR_SIZE;
    std::memcpy(quant_out, quant.data(), quant.size() * sizeof(uint32_t));
    if (leaf_boxes_out) std::memcpy(leaf_boxes_out, boxes.data(), boxes.size() * sizeof(float));
    *n_quant_out = quant.size() / 16;
    if (extent_out) *extent_out = std::max(extent, box_extent);
    return RVPT_HIP_OK;
}
#endif  // RVPT_HIP_LAB

// ASan / UBSan run of the wide regrouping of a tree (rvpt_amd/csrc/bvh_wide.cpp) on a tree built by rvpt_bvh_build: every leaf of the binary tree must come out of
// a depth-first, slot-order walk of the wide tree once and in the binary tree's left-first order, and the walk must never stack more than stack_need slots.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <utility>
#include <vector>

#include "rvpt_hip_lab.h"  // (the host-side forms of the tree live in the laboratory ABI since ABI 8)

int main()
{
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> u(-4.f, 4.f), s(0.01f, 0.3f);
    const size_t n = 20000;
    std::vector<rvpt_triangle> tris(n);
    for (auto &t : tris) {
        std::memset(&t, 0, sizeof t);
        const float c[3] = {u(rng), u(rng), u(rng)};
        for (int k = 0; k < 3; ++k) t.vert0[k] = c[k] + s(rng), t.vert1[k] = c[k] - s(rng), t.vert2[k] = c[k] + 0.5f * s(rng);
    }
    std::vector<rvpt_bvh_node> nodes(2 * n - 1);
    std::vector<uint32_t> idx(n);
    size_t n_nodes = 0;
    if (rvpt_bvh_build(tris.data(), n, nodes.data(), &n_nodes, idx.data()) != RVPT_HIP_OK) return 1;
    uint32_t shift = 1;
    while ((1ull << shift) <= n_nodes) shift += 1;
    std::vector<float> wide(n_nodes * 32);
    size_t n_wide = 0;
    uint32_t need = 0;
    if (rvpt_bvh_wide_form(nodes.data(), n_nodes, shift, wide.data(), n_nodes, &n_wide, &need) != RVPT_HIP_OK || n_wide == 0) return 2;
    std::vector<std::pair<uint32_t, uint32_t>> want, got;  // (first, count) of the leaves in visiting order
    {
        std::vector<uint32_t> st{0};
        while (!st.empty()) {
            const uint32_t i = st.back();
            st.pop_back();
            if (nodes[i].primitive_count) want.emplace_back(nodes[i].first_child_or_primitive, nodes[i].primitive_count);
            else st.push_back(nodes[i].first_child_or_primitive + 1), st.push_back(nodes[i].first_child_or_primitive);
        }
    }
    size_t deepest = 0;
    {
        std::vector<uint32_t> st{0u};  // heads; the root is wide node 0 = head 0 (count 0)
        while (!st.empty()) {
            deepest = st.size() - 1 > deepest ? st.size() - 1 : deepest;
            const uint32_t head = st.back();
            st.pop_back();
            const uint32_t count = head >> shift, first = head & ((1u << shift) - 1u);
            if (count) { got.emplace_back(first, count); continue; }
            if (first >= n_wide) return 3;
            uint32_t heads[4];
            std::memcpy(heads, &wide[first * 32 + 24], sizeof heads);
            for (int k = 3; k >= 0; --k)
                if (heads[k] != 0xFFFFFFFFu) st.push_back(heads[k]);
        }
    }
    if (got != want) return 4;
    if (deepest > need) return 5;
    std::printf("wide form: %zu binary nodes -> %zu wide nodes, %zu leaves in order, stack %zu <= %u\n", n_nodes, n_wide, got.size(), deepest, need);
    // the 64-byte quantised form of the same nodes (rvpt_bvh_quant_form): one per wide node, every child interval a superset of the exact one
    std::vector<uint32_t> quant(n_nodes * 16);
    std::vector<float> boxes(n * 8);
    size_t n_quant = 0;
    float extent = 0.f;
    if (rvpt_bvh_quant_form(nodes.data(), n_nodes, shift, n, quant.data(), n_nodes, &n_quant, boxes.data(), &extent) != RVPT_HIP_OK || n_quant != n_wide) return 6;
    const float extent_of_tree = extent;
    for (size_t w = 0; w < n_wide; ++w) {
        float origin[3], scale[3];
        std::memcpy(origin, &quant[w * 16], 12), std::memcpy(scale, &quant[w * 16 + 3], 12);
        for (int ax = 0; ax < 3; ++ax)
            for (int k = 0; k < 4; ++k) {
                uint32_t head;
                std::memcpy(&head, &wide[w * 32 + 24 + k], 4);
                if (head == 0xFFFFFFFFu) continue;
                const double lo = origin[ax] + double((quant[w * 16 + 6 + 2 * ax] >> (8 * k)) & 0xFF) * scale[ax];
                const double hi = origin[ax] + double((quant[w * 16 + 7 + 2 * ax] >> (8 * k)) & 0xFF) * scale[ax];
                if (lo > wide[w * 32 + 4 * (2 * ax) + k] || hi < wide[w * 32 + 4 * (2 * ax + 1) + k]) return 7;
            }
    }
    // ... and the refusals: a quantised form that does not fit the caller's capacity, a tree with a child sticking out of its parent
    if (rvpt_bvh_quant_form(nodes.data(), n_nodes, shift, n, quant.data(), 1, &n_quant, nullptr, nullptr) != RVPT_HIP_ERR_SIZE) return 8;
    nodes[nodes[0].first_child_or_primitive].bounds[1] += 100.f;
    if (rvpt_bvh_quant_form(nodes.data(), n_nodes, shift, n, quant.data(), n_nodes, &n_quant, boxes.data(), &extent) != RVPT_HIP_OK || n_quant != 0) return 9;
    std::printf("quantised form: %zu nodes, every child interval contains the exact one; extent %g\n", n_wide, extent_of_tree);
    return 0;
}

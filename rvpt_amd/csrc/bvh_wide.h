// bvh_wide.h — the 4-wide regrouping of a binary BVH (bvh_wide.cpp), shared by the launcher (rvpt_abi.hip) and the exported rvpt_bvh_wide_form.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/rvpt_hip.h"

namespace rv {

constexpr uint32_t kWideFormChildren = 4;           // == kWideChildren (rvpt_kernels.h)
constexpr uint32_t kWideFormEmpty = 0xFFFFFFFFu;    // == kWideEmpty: head word of an unused child slot

// 32 floats (8 quads: minx[4] maxx[4] miny[4] maxy[4] minz[4] maxz[4] head[4] pad) per wide node, breadth first; empty when the tree has no
// wide form.  stack_need = the most slots a depth-first walk can hold at once.
std::vector<float> build_wide_nodes(const rvpt_bvh_node *nodes, size_t n_nodes, uint32_t head_shift, uint32_t &stack_need);
// the same regrouping eight wide (rvpt_bvh8.hip): 64 floats (16 quads) per node — minx[8] maxx[8] miny[8] maxy[8] minz[8] maxz[8] head[8] pad[8]
std::vector<float> build_wide8_nodes(const rvpt_bvh_node *nodes, size_t n_nodes, uint32_t head_shift, uint32_t &stack_need);
// trace_bvh4_fast's leaf boxes (8 floats per triangle index, the exact box of the leaf that starts there) and the tree's largest |bound|; empty when the
// tree is not one whose every inner node contains its children (bvh_wide.cpp has the conditions)
std::vector<float> build_leaf_boxes(const rvpt_bvh_node *nodes, size_t n_nodes, size_t n_tris, float &extent);
// the 64-byte quantised form of the wide nodes (trace_bvh4q; bvh_wide.cpp has the layout): 16 words per node; empty when the tree has none
std::vector<uint32_t> build_quant_nodes(const std::vector<float> &wide, float &extent);

}  // namespace rv

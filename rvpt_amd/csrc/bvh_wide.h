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

}  // namespace rv

/*
 * harness.c — dispatches the translated reference shader over an image (TEST INFRASTRUCTURE, authoring container only).
 *
 * Replays what the reference host does around its compute pipeline: bind the eight descriptors of set 0
 * (rvpt.cpp:798-866), then vkCmdDispatch(width/16, height/16, 1) (rvpt.cpp:1035-1036, integer division) of work groups
 * of the module's LocalSize.  Invocations are independent (no shared memory, no barriers in the module), so they are
 * simply run one after the other / over OpenMP threads.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include "spv_shim.h"

void ref_spv_invoke(const shim_bindings* b, uint32_t gid_x, uint32_t gid_y);
void ref_spv_local_size(uint32_t* xyz);

/* One frame.  temporal: RGBA32F W*H*4, read and written (the accumulation image); result: written (the output image).
 * unorm8 = 1 stores through the rgba8 conversion the reference's images have; 0 keeps float radiance. */
__attribute__((visibility("default"))) int ref_spv_render(const void* settings40, const void* camera80, const void* nodes,
                                                          uint32_t n_nodes, const void* tris, uint32_t n_tris, const void* mats,
                                                          uint32_t n_mats, uint32_t width, uint32_t height, float* temporal,
                                                          float* result, int unorm8)
{
    uint32_t ls[3];
    ref_spv_local_size(ls);
    shim_image out = {(int32_t)width, (int32_t)height, result, unorm8};
    shim_image acc = {(int32_t)width, (int32_t)height, temporal, unorm8};
    static const float no_random[4] = {0, 0, 0, 0};
    shim_bindings b;
    memset(&b, 0, sizeof b);
    b.binding[0] = settings40;
    b.binding[1] = &out;
    b.binding[2] = &acc;
    b.binding[3] = no_random; /* uploaded by the reference every frame, never read by the module */
    b.binding[4] = camera80;
    b.binding[5] = nodes;
    b.length[5] = n_nodes;
    b.binding[6] = tris;
    b.length[6] = n_tris;
    b.binding[7] = mats;
    b.length[7] = n_mats;
    const uint32_t gx = width / ls[0], gy = height / ls[1];
    const long rows = (long)gy * ls[1];
#pragma omp parallel for schedule(dynamic, 1)
    for (long y = 0; y < rows; ++y)
        for (uint32_t x = 0; x < gx * ls[0]; ++x) ref_spv_invoke(&b, x, (uint32_t)y);
    return 0;
}

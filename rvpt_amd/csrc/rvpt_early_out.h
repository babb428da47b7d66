// rvpt_early_out.h — the packet-uniform early-out of the brute-force intersect loop, shared by the packet kernel (rvpt_packets.hip) and
// the wavefront trace kernel (rvpt_wavefront.hip).
//
// A ray can only accept a triangle if 0 < t < closest: a NaN t fails `t < closest`, and otherwise min3(t, u, v) > 0 gives t > 0
// (intersection.glsl:311 as rvpt_device.h evaluates it: (min3(t, u, v) > 0) & (u + v < 1) & (t < closest)).  t is the ray's distance
// to the triangle's plane alone — 15 of the test's 38 VALU.  For a packet of CAMERA rays of one 16 x 4 pixel block (same origin,
// nearly the same direction) most triangles fail that pre-test for every ray at once: the loop computes t for four triangles and
// finishes a test (barycentrics, 23 VALU, the rest of the record) only if some lane of the packet passes 0 < t < closest with the
// interval as it stood before the group (closest only shrinks, so this lets through a superset of what the sequential rule
// accepts: results are unchanged).  Default scene, 1920x1080: 61 % of all (packet, triangle) pairs skip the second half.
#pragma once

#include "rvpt_device.h"

namespace rv {

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float plane_distance(const v4f q0, const v4f q1, const f3 o, const f3 d)
{
    const f3 v0 = mk(q0.x, q0.y, q0.z), n = mk(q0.w, q1.x, q1.y);
    return div_dots(dot(v0 - o, n), dot(d, n));
}
// test_triangle_open (rvpt_device.h) from its second statement on, the plane distance given
__device__ __forceinline__ OpenTest finish_open(const PrepTri &t, const f3 o, const f3 d, const float tt)
{
    OpenTest r;
    r.tt = tt;
    const f3 p0 = fma3(d, tt, o) - t.v0;
    const float b0 = dot(p0, t.e0);
    const float b1 = dot(p0, t.e1);
    const float u = t.inv_det * fma_(t.a01, b1, t.a00 * b0);
    const float v = t.inv_det * fma_(t.a11, b1, t.a01 * b0);
    r.m = __builtin_fminf(__builtin_fminf(tt, u), v);
    r.s = u + v;
    return r;
}
// the four plane distances of a group keep v0 and n in registers (24 VGPRs) so that a test that goes on reads only the rest of its
// record: 6 LDS cycles for the first half of a record (ds_read_b128 + ds_read_b64), 10 for the rest, against 16 for the whole
struct PlaneHalf {
    f3 v0, n;
};
#ifndef RV_EARLY_GROUP
#define RV_EARLY_GROUP 4  // triangles whose plane distances are computed together before the first pre-test (2 / 4 / 8 measured: see profiles/r03_packets_sweep.txt)
#endif
__device__ __forceinline__ void intersect_run_early(const v4f *src, const uint32_t count, const f3 o, const f3 d, float &closest, uint32_t &hit)
{
    constexpr uint32_t G = RV_EARLY_GROUP;
    uint32_t i = 0;
    for (; i + G <= count; i += G) {
        float tt[G];
        PlaneHalf h[G];
#pragma unroll
        for (uint32_t k = 0; k < G; ++k) {
            const v4f q0 = src[4 * (i + k) + 0];
            const v2f q1 = *reinterpret_cast<const v2f *>(src + 4 * (i + k) + 1);
            h[k].v0 = mk(q0.x, q0.y, q0.z);
            h[k].n = mk(q0.w, q1.x, q1.y);
            tt[k] = div_dots(dot(h[k].v0 - o, h[k].n), dot(d, h[k].n));
        }
#pragma unroll
        for (uint32_t k = 0; k < G; ++k) asm volatile("" ::"v"(tt[k]));  // the plane distances of the group are all computed before the first branch
#pragma unroll
        for (uint32_t k = 0; k < G; ++k) {
            const bool maybe = (tt[k] > 0.0f) & (tt[k] < closest);
            if (ballot(maybe) != 0) {
                asm volatile("" ::: "memory");  // keep this a wave-uniform branch
                const uint32_t j = i + k;
                const v2f q1b = reinterpret_cast<const v2f *>(src + 4 * j + 1)[1];
                const v4f q2 = src[4 * j + 2], q3 = src[4 * j + 3];
                PrepTri t;
                t.v0 = h[k].v0;
                t.n = h[k].n;
                t.e0 = mk(q1b.x, q1b.y, q2.x);
                t.e1 = mk(q2.y, q2.z, q2.w);
                t.a00 = q3.x, t.a01 = q3.y, t.a11 = q3.z, t.inv_det = q3.w;
                accept_hit(finish_open(t, o, d, tt[k]), j, closest, hit);
            }
        }
    }
    for (; i < count; ++i)
        accept_hit(test_triangle_open(unpack(src[4 * i + 0], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]), o, d), i, closest, hit);
}

// Camera packets (one origin for the whole launch: begin_sample's L.o is the camera position, compute_pass.comp:151-156 +
// camera.glsl:29-51): the numerator of the plane distance, dot(v0 - o, n), is the same number for every ray of every camera packet,
// so it is computed once per triangle and work-group (camera_record, the operations of the per-ray code in their order: the same
// bits) and the pre-test of a camera round is dot(d, n) and the quotient — 9 VALU and ONE 16-byte record (n, numerator) per test
// instead of 15 VALU and 24 bytes.
__device__ __forceinline__ v4f camera_record(const v4f q0, const v4f q1, const f3 o)
{
    const f3 v0 = mk(q0.x, q0.y, q0.z), n = mk(q0.w, q1.x, q1.y);
    v4f r;
    r.x = n.x, r.y = n.y, r.z = n.z;
    r.w = dot(v0 - o, n);
    return r;
}
__device__ __forceinline__ void intersect_run_camera(const v4f *src, const v4f *cam, const uint32_t count, const f3 o, const f3 d, float &closest, uint32_t &hit)
{
    constexpr uint32_t G = RV_EARLY_GROUP;
    uint32_t i = 0;
    for (; i + G <= count; i += G) {
        float tt[G];
#pragma unroll
        for (uint32_t k = 0; k < G; ++k) {
            const v4f r = cam[i + k];
            tt[k] = div_dots(r.w, dot(d, mk(r.x, r.y, r.z)));
        }
#pragma unroll
        for (uint32_t k = 0; k < G; ++k) asm volatile("" ::"v"(tt[k]));
#pragma unroll
        for (uint32_t k = 0; k < G; ++k) {
            const bool maybe = (tt[k] > 0.0f) & (tt[k] < closest);
            if (ballot(maybe) != 0) {
                asm volatile("" ::: "memory");  // keep this a wave-uniform branch
                const uint32_t j = i + k;
                accept_hit(finish_open(unpack(src[4 * j + 0], src[4 * j + 1], src[4 * j + 2], src[4 * j + 3]), o, d, tt[k]), j, closest, hit);
            }
        }
    }
    for (; i < count; ++i)
        accept_hit(test_triangle_open(unpack(src[4 * i + 0], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]), o, d), i, closest, hit);
}

}  // namespace

}  // namespace rv

// rvpt_early_out.h — the packet-uniform early-out of the brute-force intersect loop of the packet kernel (rvpt_packets.hip).
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
#include "rvpt_rect.h"

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
// so it is computed once per triangle and work-group (camera_record) and the pre-test of a camera round needs only dot(d, n).
//
// DIVISION-FREE pre-test (round 4).  The record holds the plane equation with its sign normalised: n' = s n, a = |num| with
// s = sign(num), so that t = num / dot(d, n) = a / den' with den' = dot(d, n') — the SAME bits: negating n negates every product and
// every fused sum of the dot product exactly (round-to-nearest is symmetric), v_rcp_f32 is odd (checked on the device for every
// binary32, test_fast_division_model), and Markstein's sequence maps (-a, -b) to the same q and t.  A ray can accept the triangle
// only if 0 < t < closest (rvpt_device.h: accept_hit).  Claim: whenever the quotient t = div_dots(a, den') satisfies that,
//      !(a > closest * den')                                              (one multiply, one compare: v_cmp_ngt_f32)
// holds, so testing it instead of the quotient lets through a SUPERSET of the lanes the quotient lets through, and a test that goes
// on computes the quotient as before: the image cannot change.  Proof sketch, all binary32 inputs:
//   * den' or closest NaN, or closest = inf with den' = +-0: the product is NaN and `!(a > NaN)` is true — passes (conservative);
//   * den' = +-0 or subnormal: v_rcp_f32 returns +-inf and t is NaN — the quotient accepts nothing, nothing to show;  den' < 0:
//     t <= 0 (a >= 0), nothing to show;  |den'| > 2^126: the reciprocal flushes to 0 and t = 0, nothing to show;
//   * records are SAFE when 2^-60 <= a <= 2^60 and |n'| <= 2^60 per component: with |d| <= 1 + 2^-22 per component (camera rays are
//     normalised) no intermediate of the sequence leaves the normal range unless q = a r overflows (t = inf or NaN: rejected), so
//     t = RN(Q), Q = a / den' the real quotient.  RN(Q) < closest implies Q < closest (else RN(Q) >= RN(closest) =
//     closest by monotonicity), i.e. a < closest den' in the reals (den' > 0), hence a = RN(a) <= RN(closest den') — rounding is
//     monotone, overflow to +inf and gradual underflow (float_denorm_mode_32 = preserve) included — which is `!(a > closest den')`;
//     closest = +inf: the product is +inf and a <= inf;
//   * records that are NOT safe (a = 0: the camera lies in the triangle's plane; absurd scales; NaN / inf anywhere) store a = NaN:
//     the pre-test passes for every lane, and the finished test recomputes the numerator from the full record.
__device__ __forceinline__ v4f camera_record(const v4f q0, const v4f q1, const f3 o)
{
    const f3 v0 = mk(q0.x, q0.y, q0.z), n = mk(q0.w, q1.x, q1.y);
    bool neg;
    const float a = camera_numerator(v0, n, o, neg);  // (rvpt_rect.h: shared with the screen rectangles, host + device)
    v4f r;
    r.x = neg ? -n.x : n.x, r.y = neg ? -n.y : n.y, r.z = neg ? -n.z : n.z;
    r.w = a;
    return r;
}
// the pre-test: false only where the quotient cannot satisfy 0 < t < closest (see above)
__device__ __forceinline__ bool camera_pretest(const float a, const float den, const float closest) { return !(a > closest * den); }
// the plane distance of a test that goes on: the record's numerator, or — records marked not safe — the per-ray code's own
__device__ __forceinline__ float camera_plane_distance(const float a, const float den, const PrepTri &t, const f3 o)
{
    float num = a;
    if (ballot(!(a == a)) != 0) {  // (wave-uniform: the record is the same for every lane) a record marked not safe — rare
        asm volatile("" ::: "memory");
        num = __builtin_fabsf(dot(t.v0 - o, t.n));
    }
    return div_dots(num, den);
}
// ONE triangle of a camera round (the rectangles have left only a few: rvpt_packets.hip): pre-test on its camera record, the test finished if some lane
// of the packet passes — intersect_run_camera's per-triangle operations, with the interval as it stands now
template <typename CamPtr>
__device__ __forceinline__ void camera_test_one(const v4f *src, const CamPtr cam, const uint32_t j, const f3 o, const f3 d, float &closest, uint32_t &hit)
{
    const v4f r = cam[j];
    const float den = dot(d, mk(r.x, r.y, r.z));
    if (ballot(camera_pretest(r.w, den, closest)) != 0) {
        asm volatile("" ::: "memory");  // keep this a wave-uniform branch
        const PrepTri t = unpack(src[4 * j + 0], src[4 * j + 1], src[4 * j + 2], src[4 * j + 3]);
        accept_hit(finish_open(t, o, d, camera_plane_distance(r.w, den, t, o)), j, closest, hit);
    }
}
// `count` triangles whose prepared records start at `src` and camera records at `cam`; their indices are index0, index0 + 1, ...
template <typename CamPtr>
__device__ __forceinline__ void intersect_run_camera(const v4f *src, const CamPtr cam, const uint32_t index0, const uint32_t count, const f3 o, const f3 d, float &closest, uint32_t &hit)
{
    constexpr uint32_t G = RV_EARLY_GROUP;
    uint32_t i = 0;
    for (; i + G <= count; i += G) {
        float den[G], a[G];
        bool maybe[G];
#pragma unroll
        for (uint32_t k = 0; k < G; ++k) {
            const v4f r = cam[i + k];
            a[k] = r.w;
            den[k] = dot(d, mk(r.x, r.y, r.z));
            maybe[k] = camera_pretest(a[k], den[k], closest);  // with the interval as it stands before the group: a superset (closest only shrinks)
        }
        uint64_t any[G];
#pragma unroll
        for (uint32_t k = 0; k < G; ++k) any[k] = ballot(maybe[k]);
#pragma unroll
        for (uint32_t k = 0; k < G; ++k) asm volatile("" ::"s"(any[k]), "v"(den[k]));  // the pre-tests of the group are all decided before the first branch
#pragma unroll
        for (uint32_t k = 0; k < G; ++k) {
            if (any[k] != 0) {
                asm volatile("" ::: "memory");  // keep this a wave-uniform branch
                const uint32_t j = i + k;
                const PrepTri t = unpack(src[4 * j + 0], src[4 * j + 1], src[4 * j + 2], src[4 * j + 3]);
                accept_hit(finish_open(t, o, d, camera_plane_distance(a[k], den[k], t, o)), index0 + j, closest, hit);
            }
        }
    }
    for (; i < count; ++i)
        accept_hit(test_triangle_open(unpack(src[4 * i + 0], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]), o, d), index0 + i, closest, hit);
}

}  // namespace

}  // namespace rv

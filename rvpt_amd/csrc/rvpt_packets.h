// rvpt_packets.h — the packet form of the LDS-resident brute-force frame kernel (rvpt_packets.hip)
#pragma once

#include "rvpt_kernels.h"

namespace rv {

constexpr uint32_t kPacketQueueWords = 18;     // words of a parked path; a wave's queue holds 64 of them (4.5 KiB)
constexpr uint32_t kPacketQueueWordsAA1 = 15;  // ... one sample per pixel: without the pixel's sum of finished samples (trace_brute_packets_aa1)

// lean configuration only (Kajiya in all quadrants, pinhole camera, max_bounces >= 1), scene + materials resident in LDS
__global__ void trace_brute_packets(const FrameParams p);
__global__ void trace_brute_packets_aa1(const FrameParams p);  // aa == 1: 15-word queue entries, six work-groups per CU
__global__ void trace_brute_packets_culls(const FrameParams p);      // the same two for launches that ride with all three exact culls and whole-block work plans
__global__ void trace_brute_packets_aa1_culls(const FrameParams p);  // (the walks without a cull are not in them; nor the interleaved claim order of short launches:)
__global__ void trace_brute_packets_culls_order(const FrameParams p);
__global__ void trace_brute_packets_aa1_culls_order(const FrameParams p);
#if RVPT_HIP_LAB
// diagnostics (rvpt_hip_selftest_pretest): per element, bit 0 = the division-free pre-test of a camera round lets the pair through, bit 1 = the
// quotient's own condition 0 < t < closest holds; the numerator goes through the camera record's rule (not safe -> NaN -> always through)
__global__ void selftest_camera_pretest(const float *__restrict__ a, const float *__restrict__ den, const float *__restrict__ closest,
                                        unsigned char *__restrict__ out, uint32_t n);
#endif

// the screen rectangles of the prepared triangles for the camera of `p` (rvpt_rect.h): rects[i] = (x0 | x1 << 16, y0 | y1 << 16); one thread per triangle
__global__ void camera_rects(const FrameParams p, uint2 *__restrict__ rects, float4 *__restrict__ records);  // (+ the camera records, 16 B per triangle, or nullptr)
// the bounce cull's table for `n` prepared triangles: out[(2 A + s) * words + w] bit b = 0 only when triangle B = 32 w + b lies wholly behind the plane of A as
// seen from side s (s = 0: the side A's normal cross(e0, e1) points to), by more than `margin`, and both triangles are well shaped; bits >= n are 0
__global__ void bounce_visibility(const float4 *__restrict__ prep, uint32_t n, double margin, uint32_t words, uint32_t stride, uint32_t *__restrict__ out);
#if RVPT_HIP_LAB
// diagnostics (rvpt_hip_selftest_bounce_cull): every pixel x n_samples paths traced against EVERY triangle; on segments that leave a triangle, out[0] += pairs the
// float test accepts with the interval wide open, out[1] += those whose triangle is NOT in the row of where the segment leaves from (must stay 0), out[2] += those whose
// ray fails the slab test of the triangle's leaf box (must stay 0), out[3] / out[4] += (ray, leaf box) pairs tested / passed
__global__ void selftest_bounce_cull(const FrameParams p, uint32_t n_samples, unsigned long long *__restrict__ out);
// diagnostics (rvpt_hip_selftest_camera_rects): every pixel of the image x n_samples jittered camera rays x every triangle through the float test with
// an open interval; out[0] += accepted pairs, out[1] += accepted pairs whose block lies OUTSIDE the triangle's rectangle (must stay 0),
// out[2] += (16 x 4 block, triangle) pairs whose rectangle holds the block, out[3] += all such pairs
__global__ void selftest_camera_rects(const FrameParams p, const uint2 *__restrict__ rects, uint32_t n_samples, unsigned long long *__restrict__ out);
#endif

}  // namespace rv

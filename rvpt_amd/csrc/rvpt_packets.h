// rvpt_packets.h — the packet form of the LDS-resident brute-force frame kernel (rvpt_packets.hip)
#pragma once

#include "rvpt_kernels.h"

namespace rv {

constexpr uint32_t kPacketQueueWords = 18;  // words of a parked path; a wave's queue holds 64 of them (4.5 KiB)

// lean configuration only (Kajiya in all quadrants, pinhole camera, max_bounces >= 1), scene + materials resident in LDS
__global__ void trace_brute_packets(const FrameParams p);
// diagnostics (rvpt_hip_selftest_pretest): per element, bit 0 = the division-free pre-test of a camera round lets the pair through, bit 1 = the
// quotient's own condition 0 < t < closest holds; the numerator goes through the camera record's rule (not safe -> NaN -> always through)
__global__ void selftest_camera_pretest(const float *__restrict__ a, const float *__restrict__ den, const float *__restrict__ closest,
                                        unsigned char *__restrict__ out, uint32_t n);

}  // namespace rv

// rvpt_packets.h — the packet form of the LDS-resident brute-force frame kernel (rvpt_packets.hip)
#pragma once

#include "rvpt_kernels.h"

namespace rv {

constexpr uint32_t kPacketQueueWords = 18;  // words of a parked path; a wave's queue holds 64 of them (4.5 KiB)

// lean configuration only (Kajiya in all quadrants, pinhole camera, max_bounces >= 1), scene + materials resident in LDS
__global__ void trace_brute_packets(const FrameParams p);

}  // namespace rv

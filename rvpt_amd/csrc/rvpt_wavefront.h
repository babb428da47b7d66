// rvpt_wavefront.h — kernels of rvpt_wavefront.hip declared outside rvpt_kernels.h (that header is part of the megakernels' source
// identity, rvpt_amd.build.kernel_sha; these are not)
#pragma once

#include "rvpt_kernels.h"

namespace rv {

// closest hit of every live ray over all triangles (scene resident in LDS); EARLY_OUT: the packet-coherent form for camera rays
template <bool EARLY_OUT> __global__ void wf_trace_brute(const FrameParams p);

}  // namespace rv

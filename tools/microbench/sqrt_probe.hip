// Is there a lean, correctly rounded sqrt for normalize()?  The compiler's IEEE sqrtf is v_sqrt_f32 plus a dozen instructions of integer-stepped correction,
// half of them of the half-rate classes (profiles/r06_valu_issue_probe.txt).  Candidate: Markstein's coupled iteration on v_rsq_f32 —
//     r = rsq(x); g = x r; h = r / 2; e = fma(-h, g, 1/2); g = fma(g, e, g); h = fma(h, e, h); d = fma(-g, g, x); g = fma(d, h, g)          (8 instructions)
// compared with __builtin_sqrtf for EVERY positive normal binary32 (254 exponents x 2^23 mantissas), mismatches counted per exponent, the way rcp_probe.hip settled the reciprocal.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off sqrt_probe.hip -o bin/sqrt_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float lean_sqrt(float x)
{
    const float r = __builtin_amdgcn_rsqf(x);
    float g = x * r, h = 0.5f * r;
    const float e = fma_(-h, g, 0.5f);
    g = fma_(g, e, g);
    h = fma_(h, e, h);
    const float d = fma_(-g, g, x);
    return fma_(d, h, g);
}
__global__ void sweep(unsigned long long *bad)  // grid: (2^23 / 256) x 254
{
    const unsigned exp = blockIdx.y + 1;
    const unsigned man = blockIdx.x * 256 + threadIdx.x;
    const float x = __uint_as_float((exp << 23) | man);
    if (__float_as_uint(lean_sqrt(x)) != __float_as_uint(__builtin_sqrtf(x))) atomicAdd(&bad[exp], 1ull);
}
int main()
{
    unsigned long long *d, h[256];
    (void)hipMalloc(&d, sizeof h);
    (void)hipMemset(d, 0, sizeof h);
    hipLaunchKernelGGL(sweep, dim3((1u << 23) / 256, 254), dim3(256), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    unsigned long long total = 0;
    int first = 0, last = 0;
    for (int e = 1; e <= 254; ++e) {
        total += h[e];
        if (h[e]) {
            if (!first) first = e;
            last = e;
        }
    }
    std::printf("lean sqrt != IEEE sqrtf for %llu of %llu positive normal binary32", total, 254ull << 23);
    if (total) std::printf(" (exponents %d .. %d, i.e. 2^%d .. 2^%d)", first, last, first - 127, last - 127);
    std::printf("\n");
    for (int e = 1; e <= 254; ++e)
        if (h[e]) std::printf("  exponent %3d (2^%4d): %llu mantissas\n", e, e - 127, h[e]);
    return 0;
}

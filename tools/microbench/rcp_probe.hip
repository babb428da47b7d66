// Probe of v_rcp_f32 on gfx950, as the lean division of the triangle test builds on it:
//   1. for EVERY binary32 b (all exponents 1..254, all 2^23 mantissas, both signs not needed: rcp is odd) compare
//      r1 = fma(fma(-b, r0, 1), r0, r0), r0 = v_rcp_f32(b), with the correctly rounded 1/b (IEEE divide);
//   2. print v_rcp_f32 and the lean quotient for a table of special operands (zero, subnormal, huge, inf, NaN).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off rcp_probe.hip -o rcp_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float lean_div(float a, float b)
{
    float r = __builtin_amdgcn_rcpf(b);
    const float e = fma_(-b, r, 1.0f);
    r = fma_(e, r, r);
    const float q = a * r;
    const float rem = fma_(-b, q, a);
    return fma_(rem, r, q);
}
__global__ void sweep(unsigned long long *bad_r0, unsigned long long *bad_r1)  // grid: 254 exponents x (2^23 / 256) blocks
{
    const unsigned exp = blockIdx.y + 1;
    const unsigned man = blockIdx.x * 256 + threadIdx.x;
    const float b = __uint_as_float((exp << 23) | man);
    const float want = 1.0f / b;
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float r1 = fma_(fma_(-b, r0, 1.0f), r0, r0);
    if (__float_as_uint(r0) != __float_as_uint(want)) atomicAdd(&bad_r0[exp], 1ull);
    if (__float_as_uint(r1) != __float_as_uint(want)) atomicAdd(&bad_r1[exp], 1ull);
}
__global__ void table(const float *in, unsigned n, unsigned *out)
{
    const unsigned i = threadIdx.x;
    if (i >= n) return;
    out[4 * i] = __float_as_uint(__builtin_amdgcn_rcpf(in[i]));
    out[4 * i + 1] = __float_as_uint(lean_div(1.0f, in[i]));
    out[4 * i + 2] = __float_as_uint(lean_div(3.0f, in[i]));
    out[4 * i + 3] = __float_as_uint(lean_div(in[i], 3.0f));
}
static float f(unsigned u) { float x; std::memcpy(&x, &u, 4); return x; }
int main()
{
    unsigned long long *d0, *d1, h0[256], h1[256];
    hipMalloc(&d0, 256 * 8); hipMalloc(&d1, 256 * 8);
    hipMemset(d0, 0, 256 * 8); hipMemset(d1, 0, 256 * 8);
    hipLaunchKernelGGL(sweep, dim3((1u << 23) / 256, 254), dim3(256), 0, 0, d0, d1);
    hipMemcpy(h0, d0, sizeof h0, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, sizeof h1, hipMemcpyDeviceToHost);
    unsigned long long t0 = 0, t1 = 0;
    for (int e = 1; e <= 254; ++e) {
        t0 += h0[e]; t1 += h1[e];
        if (h1[e]) std::printf("exponent %3d (2^%d): refined reciprocal != RN(1/b) for %llu mantissas (raw v_rcp_f32: %llu)\n", e, e - 127, h1[e], h0[e]);
    }
    std::printf("all normal b: raw v_rcp_f32 differs from RN(1/b) for %llu values, refined for %llu\n", t0, t1);
    const unsigned bits[] = {0x00000000u, 0x80000000u, 0x00000001u, 0x007fffffu, 0x00400000u, 0x00800000u, 0x7e800000u /*2^126*/, 0x7e800001u, 0x7ec00000u,
                             0x7f000000u /*2^127*/, 0x7f7fffffu, 0x7f800000u, 0xff800000u, 0x7fc00000u, 0x3f800000u, 0x40400000u, 0x00ffffffu, 0x7e7fffffu};
    const unsigned n = sizeof bits / 4;
    float hin[32]; for (unsigned i = 0; i < n; ++i) hin[i] = f(bits[i]);
    float *din; unsigned *dout, hout[128];
    hipMalloc(&din, 128); hipMalloc(&dout, 512);
    hipMemcpy(din, hin, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(table, dim3(1), dim3(64), 0, 0, din, n, dout);
    hipMemcpy(hout, dout, n * 16, hipMemcpyDeviceToHost);
    std::printf("%-10s %-10s %-12s %-12s %-12s   (host IEEE: 1/x, 3/x, x/3)\n", "x", "rcp(x)", "lean(1,x)", "lean(3,x)", "lean(x,3)");
    for (unsigned i = 0; i < n; ++i) {
        volatile float x = hin[i];
        float a = 1.0f / x, b = 3.0f / x, c = x / 3.0f;
        unsigned ua, ub, uc; std::memcpy(&ua, &a, 4); std::memcpy(&ub, &b, 4); std::memcpy(&uc, &c, 4);
        std::printf("%08x   %08x   %08x     %08x     %08x       %08x %08x %08x\n", bits[i], hout[4 * i], hout[4 * i + 1], hout[4 * i + 2], hout[4 * i + 3], ua, ub, uc);
    }
    return 0;
}

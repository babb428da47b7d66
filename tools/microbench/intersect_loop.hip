// Micro-benchmark of the brute-force intersect loop alone (no shading, no regeneration, no tail): every lane
// tests its ray(s) against N LDS-resident prepared triangles, repeated R times.  Variants:
//   A: 1 ray per lane, 4 triangles per iteration (the production loop)
//   B: 2 rays per lane, 2 triangles per iteration (half the LDS reads per test)
//   C: 1 ray per lane, 2 triangles per iteration
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize intersect_loop.hip -o intersect_loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>

typedef float v4f __attribute__((ext_vector_type(4)));
struct f3 { float x, y, z; };
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ void test(const v4f q0, const v4f q1, const v4f q2, const v4f q3, const f3 o, const f3 d, unsigned idx, float &closest, unsigned &hit)
{
    const f3 v0{q0.x, q0.y, q0.z}, n{q0.w, q1.x, q1.y}, e0{q1.z, q1.w, q2.x}, e1{q2.y, q2.z, q2.w};
    const f3 q{v0.x - o.x, v0.y - o.y, v0.z - o.z};
    const float t = dot(q, n) / dot(d, n);
    const f3 p0{fma_(d.x, t, o.x) - v0.x, fma_(d.y, t, o.y) - v0.y, fma_(d.z, t, o.z) - v0.z};
    const float b0 = dot(p0, e0), b1 = dot(p0, e1);
    const float u = q3.w * fma_(q3.y, b1, q3.x * b0), v = q3.w * fma_(q3.z, b1, q3.y * b0);
    const bool acc = (0.f < t) & (t < closest) & (0.f < u) & (0.f < v) & (u + v < 1.f);
    closest = acc ? t : closest;
    hit = acc ? idx : hit;
}

template <int VARIANT>
__global__ __launch_bounds__(256) void loop(const v4f *prep, unsigned n, unsigned reps, float *out)
{
    extern __shared__ __attribute__((aligned(16))) v4f lds[];
    for (unsigned i = threadIdx.x; i < 4 * n; i += 256) lds[i] = prep[i];
    __syncthreads();
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    f3 o{0.01f * (gid & 63), 0.02f * ((gid >> 6) & 63), -3.f}, d{0.001f * (gid & 31), 0.002f * ((gid >> 5) & 31), 1.f};
    f3 o2{o.x + 0.3f, o.y - 0.2f, -3.f}, d2{-d.x, d.y * 0.5f, 1.f};
    float acc = 0.f;
    for (unsigned r = 0; r < reps; ++r) {
        float c1 = __builtin_inff(), c2 = __builtin_inff();
        unsigned h1 = ~0u, h2 = ~0u;
        if (VARIANT == 0) {
#pragma unroll 4
            for (unsigned i = 0; i < n; ++i) test(lds[4 * i], lds[4 * i + 1], lds[4 * i + 2], lds[4 * i + 3], o, d, i, c1, h1);
        } else if (VARIANT == 1) {
#pragma unroll 2
            for (unsigned i = 0; i < n; ++i) {
                const v4f q0 = lds[4 * i], q1 = lds[4 * i + 1], q2 = lds[4 * i + 2], q3 = lds[4 * i + 3];
                test(q0, q1, q2, q3, o, d, i, c1, h1);
                test(q0, q1, q2, q3, o2, d2, i, c2, h2);
            }
        } else {
#pragma unroll 2
            for (unsigned i = 0; i < n; ++i) test(lds[4 * i], lds[4 * i + 1], lds[4 * i + 2], lds[4 * i + 3], o, d, i, c1, h1);
        }
        acc += (h1 != ~0u ? c1 : 0.f) + (h2 != ~0u ? c2 : 0.f);
        o.x += 1e-4f;  // keep the compiler from hoisting the loop
        o2.y += 1e-4f;
    }
    out[gid] = acc;
}

int main()
{
    const unsigned n = 143, reps = 200, blocks = 256 * 4;
    std::vector<float> h(16 * n);
    for (unsigned i = 0; i < n; ++i) {  // plausible prepared records
        float *q = &h[16 * i];
        const float x = std::cos(i * 0.7f), y = std::sin(i * 1.3f), z = 0.2f * (i % 7);
        const float e0[3] = {0.3f, 0.05f, 0.02f}, e1[3] = {0.04f, 0.35f, -0.03f};
        const float nn[3] = {e0[1] * e1[2] - e0[2] * e1[1], e0[2] * e1[0] - e0[0] * e1[2], e0[0] * e1[1] - e0[1] * e1[0]};
        const float a00 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2], a11 = e0[0] * e0[0] + e0[1] * e0[1] + e0[2] * e0[2];
        const float a01 = -(e0[0] * e1[0] + e0[1] * e1[1] + e0[2] * e1[2]);
        const float rec[16] = {x, y, z, nn[0], nn[1], nn[2], e0[0], e0[1], e0[2], e1[0], e1[1], e1[2], a00, a01, a11, 1.f / (a00 * a11 - a01 * a01)};
        for (int k = 0; k < 16; ++k) q[k] = rec[k];
    }
    v4f *d_prep; float *d_out;
    hipMalloc(&d_prep, h.size() * 4); hipMalloc(&d_out, blocks * 256 * 4);
    hipMemcpy(d_prep, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char *names[3] = {"A 1 ray/lane x4 tris", "B 2 rays/lane x2 tris", "C 1 ray/lane x2 tris"};
    for (int v = 0; v < 3; ++v) {
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(e0);
            if (v == 0) hipLaunchKernelGGL(loop<0>, dim3(blocks), dim3(256), n * 64, 0, d_prep, n, reps, d_out);
            if (v == 1) hipLaunchKernelGGL(loop<1>, dim3(blocks), dim3(256), n * 64, 0, d_prep, n, reps, d_out);
            if (v == 2) hipLaunchKernelGGL(loop<2>, dim3(blocks), dim3(256), n * 64, 0, d_prep, n, reps, d_out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double tests = double(blocks) * 256 * n * reps * (v == 1 ? 2 : 1);
            if (it == 2) std::printf("%-24s %8.3f ms  %.3e tests/s\n", names[v], ms, tests / (ms * 1e-3));
        }
    }
    return 0;
}

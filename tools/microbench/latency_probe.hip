// Unloaded latency of one dependent load on gfx950 (one wave per CU, pointer chase; core clocks by s_memtime's
// shader clock and, as a cross-check, wall time at the nominal 2.4 GHz): global load hitting the vL1D / the L2, an LDS read,
// and a FLAT load that resolves to LDS — the building blocks of the BVH traversal's per-step chain (profiles/EXPERIMENTS.md: what binds the binary walk).
// build: hipcc --offload-arch=gfx950 -O3 latency_probe.hip -o latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void chase_global(const uint32_t *__restrict__ next, uint32_t start, uint32_t iters, uint32_t *out, unsigned long long *ticks)
{
    uint32_t i = start + threadIdx.x * 16u;  // each lane its own chain element (same latency class, divergent lines)
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t k = 0; k < iters; ++k) i = next[i];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
    if (i == 0xFFFFFFFFu) out[0] = i;
}
template <bool FLAT>
__global__ void chase_lds(uint32_t iters, uint32_t *out, unsigned long long *ticks)
{
    __shared__ uint32_t ring[4096];
    for (uint32_t j = threadIdx.x; j < 4096; j += 64) ring[j] = (j * 17u + 64u) & 4095u;
    __syncthreads();
    uint32_t i = threadIdx.x;
    const uint32_t *generic = ring;
    if (FLAT) asm volatile("" : "+v"(generic));  // launder the pointer: the compiler can no longer prove it is LDS -> flat_load
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t k = 0; k < iters; ++k) i = FLAT ? generic[i] : ring[i];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
    if (i == 0xFFFFFFFFu) out[0] = i;
}
// dependent VALU chain: latency of back-to-back dependent v_fma_f32 with one wave on the SIMD
__global__ void chase_valu(uint32_t iters, float *out, unsigned long long *ticks)
{
    float x = threadIdx.x * 1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (uint32_t k = 0; k < iters; ++k) {
#pragma unroll
        for (int u = 0; u < 16; ++u) x = __builtin_fmaf(x, 1.0001f, 0.5f);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) *ticks = t1 - t0;
    if (x == 123.0f) out[0] = x;
}

int main()
{
    const uint32_t iters = 20000;
    uint32_t *d_next, *d_out;
    unsigned long long *d_ticks, ticks;
    hipMalloc(&d_out, 64), hipMalloc(&d_ticks, 8);
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    auto report = [&](const char *name, double per) {
        float ms;
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        hipMemcpy(&ticks, d_ticks, 8, hipMemcpyDeviceToHost);
        std::printf("%-44s %8.1f s_memtime ticks   %8.1f clocks at 2.4 GHz (wall)\n", name, double(ticks) / per, ms * 1e-3 * 2.4e9 / per);
    };
    for (uint32_t kib : {8u, 2048u, 262144u}) {  // ring: 8 KiB (vL1D), 2 MiB (L2), 256 MiB (beyond L2: Infinity Cache / HBM)
        const uint32_t n = kib * 256u;       // dwords
        std::vector<uint32_t> h(n);
        const uint32_t stride = 16u * 67u;   // a different 64-byte line every hop, co-prime walk over the ring
        for (uint32_t i = 0; i < n; ++i) h[i] = (i + stride) % n;
        hipMalloc(&d_next, size_t(n) * 4);
        hipMemcpy(d_next, h.data(), size_t(n) * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(chase_global, dim3(1), dim3(64), 0, 0, d_next, 0u, 2000u, d_out, d_ticks);
        hipEventRecord(a);
        hipLaunchKernelGGL(chase_global, dim3(1), dim3(64), 0, 0, d_next, 0u, iters, d_out, d_ticks);
        hipEventRecord(b);
        char name[96];
        std::snprintf(name, sizeof name, "global_load_dword, ring of %u KiB", kib);
        report(name, iters);
        hipFree(d_next);
    }
    hipEventRecord(a);
    hipLaunchKernelGGL((chase_lds<false>), dim3(1), dim3(64), 0, 0, iters, d_out, d_ticks);
    hipEventRecord(b);
    report("ds_read_b32", iters);
    hipEventRecord(a);
    hipLaunchKernelGGL((chase_lds<true>), dim3(1), dim3(64), 0, 0, iters, d_out, d_ticks);
    hipEventRecord(b);
    report("flat_load_dword resolving to LDS", iters);
    hipEventRecord(a);
    hipLaunchKernelGGL(chase_valu, dim3(1), dim3(64), 0, 0, iters, reinterpret_cast<float *>(d_out), d_ticks);
    hipEventRecord(b);
    report("dependent v_fma_f32 (per instruction)", iters * 16.0);
    return 0;
}

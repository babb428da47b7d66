// What does a divergent wave-level vector load cost on gfx950's texture addresser / vL1D, as a function of the load's
// width (dword, dwordx2, dwordx4) and of the number of ACTIVE lanes?  The HBM-resident BVH kernel is bound by this unit
// (profiles/EXPERIMENTS.md: what binds the binary walk); the answer decides whether narrower records or lane-cooperative loads can pay.
//   every lane reads W dwords at a pseudo-random 64-byte-aligned record of a table (16 KiB: vL1D hits; 4 MiB: L2 hits),
//   U independent loads per loop iteration, `active` lanes of each wave enabled (the others leave at the top).
// build: hipcc --offload-arch=gfx950 -O3 ta_probe.hip -o ta_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int W> struct Vec;
template <> struct Vec<1> { using T = float; };
template <> struct Vec<2> { using T = float2; };
template <> struct Vec<4> { using T = float4; };
__device__ __forceinline__ float sum(float v) { return v; }
__device__ __forceinline__ float sum(float2 v) { return v.x + v.y; }
__device__ __forceinline__ float sum(float4 v) { return v.x + v.y + v.z + v.w; }

template <int W, int U, bool QUAD = false>
__global__ __launch_bounds__(256) void probe(const float *__restrict__ table, uint32_t record_mask, uint32_t active, uint32_t iters, float *out)
{
    const uint32_t lane = threadIdx.x & 63u;
    if (lane >= active) return;
    uint32_t x = ((blockIdx.x * 256u + threadIdx.x) >> (QUAD ? 2 : 0)) * 2654435761u + 12345u;  // QUAD: four neighbouring lanes share a record
    float acc = 0.0f;
    for (uint32_t i = 0; i < iters; ++i) {
        typename Vec<W>::T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            x = x * 1664525u + 1013904223u;
            const uint32_t rec = (x >> 8) & record_mask;  // one 64-byte record per lane: fully divergent
            v[u] = *reinterpret_cast<const typename Vec<W>::T *>(table + 16u * rec + (QUAD ? 4u * (lane & 3u) : 4u * (u & 3) * (W == 4 ? 1 : 0)));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += sum(v[u]);
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int W, int U, bool QUAD = false>
static double run(const float *table, uint32_t records, uint32_t active, int wg_per_cu, float *out)
{
    const uint32_t iters = 4000;
    const int cus = 256;
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    hipLaunchKernelGGL((probe<W, U, QUAD>), dim3(cus * wg_per_cu), dim3(256), 0, 0, table, records - 1, active, 200u, out);
    hipEventRecord(a);
    hipLaunchKernelGGL((probe<W, U, QUAD>), dim3(cus * wg_per_cu), dim3(256), 0, 0, table, records - 1, active, iters, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double wave_loads_per_cu = double(wg_per_cu) * 4.0 * iters * U;
    return ms * 1e-3 * 2.4e9 / wave_loads_per_cu;  // CU clocks (at the nominal 2.4 GHz) per wave-level load instruction
}

int main()
{
    const uint32_t max_records = (4u << 20) / 64u;
    std::vector<float> h(16u * max_records, 1.0f);
    float *table, *out;
    hipMalloc(&table, h.size() * 4), hipMalloc(&out, 64);
    hipMemcpy(table, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const uint32_t sizes[2] = {(16u << 10) / 64u, max_records};
    const char *names[2] = {"16 KiB table (vL1D hits)", "4 MiB table (L2 hits)"};
    for (int s = 0; s < 2; ++s) {
        std::printf("%s: CU clocks at 2.4 GHz per wave-level load, 6 work-groups of 256 per CU, 4 independent loads per iteration\n", names[s]);
        std::printf("  active lanes      dword    dwordx2    dwordx4\n");
        const uint32_t act[] = {64, 48, 32, 16, 8, 4};
        for (uint32_t a : act)
            std::printf("  %12u   %8.2f   %8.2f   %8.2f\n", a, run<1, 4>(table, sizes[s], a, 6, out), run<2, 4>(table, sizes[s], a, 6, out),
                        run<4, 4>(table, sizes[s], a, 6, out));
    }
    std::printf("same record for 4 neighbouring lanes (each reads its own 16-byte quarter), dwordx4, 16 KiB / 4 MiB: %.2f / %.2f\n",
                run<4, 4, true>(table, sizes[0], 64, 6, out), run<4, 4, true>(table, sizes[1], 64, 6, out));
    return 0;
}

// Micro-benchmark of the brute-force intersect loop alone, round 2: where the triangle record comes from and how the
// quotient is formed.  Every lane tests its ray against N prepared triangles, repeated R times; all variants must print
// the same checksum (they compute the same closest hits).
//   A  LDS broadcast reads (ds_read_b128 x4 per record and wave), IEEE divide        -- the round-1 production loop
//   S  scalar loads (s_load_dwordx16 per record and wave; operands straight from SGPRs), IEEE divide
//   P  as S with the next record requested before the current one is used
//   L  LDS reads, lean divide (v_rcp + one Newton step + one Markstein correction; no scale / fixup)
//   SL scalar loads + lean divide
//   S2 scalar loads, 2 rays per lane
// Also counts, over random operands, how often the lean quotient differs from the IEEE one (must be 0 for normal-range
// operands if it is ever to replace it).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize intersect_variants.hip -o intersect_variants
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const v4f *cptr;
struct f3 { float x, y, z; };
__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
__device__ __forceinline__ float lean_div(float a, float b)
{
    float r = __builtin_amdgcn_rcpf(b);
    const float e = fma_(-b, r, 1.0f);
    r = fma_(e, r, r);
    const float q = a * r;
    const float rem = fma_(-b, q, a);
    return fma_(rem, r, q);
}
template <bool LEAN>
__device__ __forceinline__ void test(const v4f q0, const v4f q1, const v4f q2, const v4f q3, const f3 o, const f3 d, unsigned idx, float &closest, unsigned &hit)
{
    const f3 v0{q0.x, q0.y, q0.z}, n{q0.w, q1.x, q1.y}, e0{q1.z, q1.w, q2.x}, e1{q2.y, q2.z, q2.w};
    const f3 q{v0.x - o.x, v0.y - o.y, v0.z - o.z};
    const float num = dot(q, n), den = dot(d, n);
    const float t = LEAN ? lean_div(num, den) : num / den;
    const f3 p0{fma_(d.x, t, o.x) - v0.x, fma_(d.y, t, o.y) - v0.y, fma_(d.z, t, o.z) - v0.z};
    const float b0 = dot(p0, e0), b1 = dot(p0, e1);
    const float u = q3.w * fma_(q3.y, b1, q3.x * b0), v = q3.w * fma_(q3.z, b1, q3.y * b0);
    const bool acc = (0.f < t) & (t < closest) & (0.f < u) & (0.f < v) & (u + v < 1.f);
    closest = acc ? t : closest;
    hit = acc ? idx : hit;
}

enum { A, S, P, L, SL, S2, A2, L2, L4 };
template <int V>
__global__ __launch_bounds__(256) void loop(const v4f *__restrict__ prep, unsigned n, unsigned reps, float *__restrict__ out, float oz)
{
    extern __shared__ __attribute__((aligned(16))) v4f lds[];
    if (V == A || V == L || V == A2 || V == L2 || V == L4) {
        for (unsigned i = threadIdx.x; i < 4 * n; i += 256) lds[i] = prep[i];
        __syncthreads();
    }
    cptr sp = (cptr)prep;
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    f3 o{0.01f * (gid & 63), 0.02f * ((gid >> 6) & 63), oz}, d{0.001f * (gid & 31), 0.002f * ((gid >> 5) & 31), 1.f};
    f3 o2{o.x + 0.3f, o.y - 0.2f, oz}, d2{-d.x, d.y * 0.5f, 1.f};
    float acc = 0.f;
    for (unsigned r = 0; r < reps; ++r) {
        float c1 = __builtin_inff(), c2 = __builtin_inff();
        unsigned h1 = ~0u, h2 = ~0u;
        if (V == A || V == L) {
#pragma unroll 4
            for (unsigned i = 0; i < n; ++i) test<V == L>(lds[4 * i], lds[4 * i + 1], lds[4 * i + 2], lds[4 * i + 3], o, d, i, c1, h1);
        } else if (V == S || V == SL) {
#pragma unroll 4
            for (unsigned i = 0; i < n; ++i) test<V == SL>(sp[4 * i], sp[4 * i + 1], sp[4 * i + 2], sp[4 * i + 3], o, d, i, c1, h1);
        } else if (V == P) {
            v4f q0 = sp[0], q1 = sp[1], q2 = sp[2], q3 = sp[3];
#pragma unroll 2
            for (unsigned i = 0; i < n; ++i) {
                const unsigned j = (i + 1 < n) ? i + 1 : i;
                const v4f n0 = sp[4 * j], n1 = sp[4 * j + 1], n2 = sp[4 * j + 2], n3 = sp[4 * j + 3];
                test<false>(q0, q1, q2, q3, o, d, i, c1, h1);
                q0 = n0; q1 = n1; q2 = n2; q3 = n3;
            }
        } else if (V == S2) {
#pragma unroll 2
            for (unsigned i = 0; i < n; ++i) {
                const v4f q0 = sp[4 * i], q1 = sp[4 * i + 1], q2 = sp[4 * i + 2], q3 = sp[4 * i + 3];
                test<false>(q0, q1, q2, q3, o, d, i, c1, h1);
                test<false>(q0, q1, q2, q3, o2, d2, i, c2, h2);
            }
        } else if (V == A2 || V == L2) {
#pragma unroll 2
            for (unsigned i = 0; i < n; ++i) {
                const v4f q0 = lds[4 * i], q1 = lds[4 * i + 1], q2 = lds[4 * i + 2], q3 = lds[4 * i + 3];
                test<V == L2>(q0, q1, q2, q3, o, d, i, c1, h1);
                test<V == L2>(q0, q1, q2, q3, o2, d2, i, c2, h2);
            }
        } else {  // L4: four rays per lane (rays 3, 4 = mirrored copies)
            const f3 o3{-o.x, o.y, oz}, o4{o.x, -o.y, oz};
            float c3 = __builtin_inff(), c4 = __builtin_inff();
            unsigned h3 = ~0u, h4 = ~0u;
            for (unsigned i = 0; i < n; ++i) {
                const v4f q0 = lds[4 * i], q1 = lds[4 * i + 1], q2 = lds[4 * i + 2], q3 = lds[4 * i + 3];
                test<true>(q0, q1, q2, q3, o, d, i, c1, h1);
                test<true>(q0, q1, q2, q3, o2, d2, i, c2, h2);
                test<true>(q0, q1, q2, q3, o3, d, i, c3, h3);
                test<true>(q0, q1, q2, q3, o4, d2, i, c4, h4);
            }
            acc += (h3 != ~0u ? c3 : 0.f) + (h4 != ~0u ? c4 : 0.f);
        }
        acc += (h1 != ~0u ? c1 : 0.f) + ((V == S2 || V == A2 || V == L2 || V == L4) ? (h2 != ~0u ? c2 : 0.f) : 0.f);
        o.x += 1e-4f;  // keep the compiler from hoisting the loop
        o2.y += 1e-4f;
    }
    out[gid] = acc;
}

// lean vs IEEE quotient on pseudo-random operands of moderate exponent (|x| in 2^-20 .. 2^20) + the mantissa extremes
__global__ void div_check(unsigned long long *mismatch, unsigned rounds)
{
    unsigned s = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
    unsigned long long bad = 0;
    for (unsigned r = 0; r < rounds; ++r) {
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        unsigned ma = s;
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        unsigned mb = s;
        // exponents 107..147 (2^-20..2^20), random sign and mantissa; every 16th denominator mantissa all ones / all zeros / one
        unsigned ea = 107 + (ma >> 23) % 41, eb = 107 + (mb >> 23) % 41;
        unsigned fa = (ma & 0x807fffffu) | (ea << 23), fb = (mb & 0x807fffffu) | (eb << 23);
        if ((r & 15) == 0) fb |= 0x007fffffu;
        if ((r & 15) == 1) fb &= 0xff800000u;
        if ((r & 15) == 2) fb = (fb & 0xff800000u) | 1u;
        if ((r & 15) == 3) fb = (fb & 0xff800000u) | 0x007ffffeu;
        const float a = __uint_as_float(fa), b = __uint_as_float(fb);
        const float q1 = a / b, q2 = lean_div(a, b);
        bad += (__float_as_uint(q1) != __float_as_uint(q2));
    }
    if (bad) atomicAdd(mismatch, bad);
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
    const char *names[9] = {"A  LDS, IEEE div", "S  s_load, IEEE div", "P  s_load prefetch, IEEE", "L  LDS, lean div", "SL s_load, lean div", "S2 s_load, 2 rays/lane",
                            "A2 LDS, IEEE, 2 rays/lane", "L2 LDS, lean, 2 rays/lane", "L4 LDS, lean, 4 rays/lane"};
    std::vector<float> host(blocks * 256);
    for (int v = 0; v < 9; ++v) {
        for (int it = 0; it < 4; ++it) {
            hipEventRecord(e0);
            const float oz = -3.f;
            if (v == 0) hipLaunchKernelGGL(loop<A>, dim3(blocks), dim3(256), n * 64, 0, d_prep, n, reps, d_out, oz);
            if (v == 1) hipLaunchKernelGGL(loop<S>, dim3(blocks), dim3(256), 0, 0, d_prep, n, reps, d_out, oz);
            if (v == 2) hipLaunchKernelGGL(loop<P>, dim3(blocks), dim3(256), 0, 0, d_prep, n, reps, d_out, oz);
            if (v == 3) hipLaunchKernelGGL(loop<L>, dim3(blocks), dim3(256), n * 64, 0, d_prep, n, reps, d_out, oz);
            if (v == 4) hipLaunchKernelGGL(loop<SL>, dim3(blocks), dim3(256), 0, 0, d_prep, n, reps, d_out, oz);
            if (v == 5) hipLaunchKernelGGL(loop<S2>, dim3(blocks), dim3(256), 0, 0, d_prep, n, reps, d_out, oz);
            if (v == 6) hipLaunchKernelGGL(loop<A2>, dim3(blocks), dim3(256), n * 64, 0, d_prep, n, reps, d_out, oz);
            if (v == 7) hipLaunchKernelGGL(loop<L2>, dim3(blocks), dim3(256), n * 64, 0, d_prep, n, reps, d_out, oz);
            if (v == 8) hipLaunchKernelGGL(loop<L4>, dim3(blocks), dim3(256), n * 64, 0, d_prep, n, reps, d_out, oz);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double tests = double(blocks) * 256 * n * reps * (v == 8 ? 4 : (v >= 5 ? 2 : 1));
            if (it == 3) {
                hipMemcpy(host.data(), d_out, host.size() * 4, hipMemcpyDeviceToHost);
                unsigned long long cs = 1469598103934665603ull;
                for (float f : host) { unsigned u; std::memcpy(&u, &f, 4); cs = (cs ^ u) * 1099511628211ull; }
                std::printf("%-26s %8.3f ms  %.3e tests/s  checksum %016llx\n", names[v], ms, tests / (ms * 1e-3), cs);
            }
        }
    }
    unsigned long long *d_bad, bad = 0;
    hipMalloc(&d_bad, 8); hipMemset(d_bad, 0, 8);
    hipLaunchKernelGGL(div_check, dim3(4096), dim3(256), 0, 0, d_bad, 100000u);
    hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost);
    std::printf("lean vs IEEE quotient: %llu mismatches in %.3e pairs\n", bad, 4096.0 * 256 * 100000);
    return 0;
}

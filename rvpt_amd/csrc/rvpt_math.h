// rvpt_math.h — float32 arithmetic of the gfx950 path tracer (device + the few host-side uses).
//
// Implements DESIGN.md "Arithmetic specification": IEEE-754 binary32, round-to-nearest-even, IEEE
// divide / sqrt, and a fused multiply-add exactly where fma_() is written (the translation unit is
// compiled with -ffp-contract=off so the compiler adds none of its own).  GLSL leaves contraction
// and sin/cos/tan/normalize precision to the driver; these are the choices this backend makes.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define RV_HD __host__ __device__ __forceinline__

namespace rv {

// compute_pass.comp:5-12 rounded to float
constexpr float kPi = 3.14159274101257324219f;
constexpr float kTwoPi = 6.28318548202514648438f;
constexpr float kInvPi = 0.31830987334251403809f;
constexpr float kEpsilon = 0.005f;

struct f3 {
    float x, y, z;
};

RV_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
RV_HD f3 mk(float x, float y, float z) { return f3{x, y, z}; }
RV_HD f3 operator+(f3 a, f3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RV_HD f3 operator-(f3 a, f3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RV_HD f3 operator*(f3 a, f3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
RV_HD f3 operator*(f3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
RV_HD f3 operator-(f3 a) { return mk(-a.x, -a.y, -a.z); }
// a*s + b, one fma per component
RV_HD f3 fma3(f3 a, float s, f3 b) { return mk(fma_(a.x, s, b.x), fma_(a.y, s, b.y), fma_(a.z, s, b.z)); }
// a*b + c component-wise
RV_HD f3 fma3(f3 a, f3 b, f3 c) { return mk(fma_(a.x, b.x, c.x), fma_(a.y, b.y, c.y), fma_(a.z, b.z, c.z)); }
// x*x' then fused y, z terms
RV_HD float dot(f3 a, f3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }
RV_HD f3 cross(f3 a, f3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RV_HD f3 normalize(f3 a)
{
    const float len = __builtin_sqrtf(dot(a, a));
#if defined(__HIP_DEVICE_COMPILE__)
    // 1 / len on the device: the refined v_rcp_f32 of div_dots — RN(1 / x) for EVERY x in [2^-126, 2^126] (exhaustive: tools/microbench/rcp_probe.hip), i.e. the very bits of
    // the IEEE divide — wherever every lane's len is a positive normal number (a square root is never above 2^64, so that is the whole of the proven range); a wave with a
    // len of 0, inf or NaN (a zero vector, an overflowed dot product) takes the IEEE expansion as before.  3 instructions instead of 11, in every camera round and every hit.
    float inv;
    if (__builtin_amdgcn_ballot_w64(!__builtin_amdgcn_class(len, 0x100)) == 0) {  // (0x100: positive normal)
        const float r = __builtin_amdgcn_rcpf(len);
        inv = fma_(fma_(-len, r, 1.0f), r, r);
    } else {
        inv = 1.0f / len;
    }
#else
    const float inv = 1.0f / len;
#endif
    return a * inv;
}

// The quotient of the intersect loop, t = dot(v0 - o, n) / dot(d, n) (intersection.glsl:292): Markstein's sequence on
// v_rcp_f32 — 6 VALU instead of the 11 of the scaled IEEE expansion (v_div_scale x2, v_rcp, 5 fma, v_div_fmas, v_div_fixup).
// The refined reciprocal is the correctly rounded 1/b for every b in [2^-126, 2^126] (exhaustive:
// tools/microbench/rcp_probe.hip), which makes the result the correctly rounded a/b whenever nothing leaves the normal
// range; outside, v_rcp_f32 flushes (+-inf for zero / subnormal b, +-0 when 1/b would be subnormal) and the result is NaN
// or 0, which the accept test rejects.  DESIGN.md §2; oracle: o_div_dots; reference shader: shim_fdiv_dots.
__device__ __forceinline__ float div_dots(float a, float b)
{
    float r = __builtin_amdgcn_rcpf(b);
    r = fma_(fma_(-b, r, 1.0f), r, r);
    const float q = a * r;
    return fma_(fma_(-b, q, a), r, q);
}

// Range-reduce by pi/2 in three fused steps, evaluate the odd/even minimax polynomials on
// [-pi/4, pi/4], pick by quadrant.  Valid for the arguments this renderer produces (|x| < ~8).
RV_HD void sincos_det(float x, float &s, float &c)
{
    const float q = __builtin_floorf(fma_(x, 0.636619746685028076171875f, 0.5f));
    float r = fma_(q, -1.5703125f, x);
    r = fma_(q, -4.837512969970703125e-4f, r);
    r = fma_(q, -7.54978995489188216e-8f, r);
    const float r2 = r * r;
    float ps = fma_(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fma_(r2, ps, -1.6666654611e-1f);
    const float sr = fma_(ps * r2, r, r);
    float pc = fma_(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fma_(r2, pc, 4.166664568298827e-2f);
    const float cr = fma_(pc * r2, r2, fma_(r2, -0.5f, 1.0f));
    const int quad = static_cast<int>(q) & 3;
    const float a = (quad & 1) ? cr : sr;  // |sin| source
    const float b = (quad & 1) ? sr : cr;  // |cos| source
    s = (quad & 2) ? -a : a;
    c = ((quad + 1) & 2) ? -b : b;
}
RV_HD float tan_det(float x)
{
    float s, c;
    sincos_det(x, s, c);
    return s / c;
}

// util.glsl:25-33
RV_HD uint32_t wang_hash(uint32_t v)
{
    v = (v ^ 61u) ^ (v >> 16);
    v *= 9u;
    v ^= v >> 4;
    v *= 0x27d4eb2du;
    v ^= v >> 15;
    return v;
}
// util.glsl:38-50: xorshift32 step, then state / 2^32 (RNE conversion, exact scaling)
RV_HD float rand01(uint32_t &state)
{
    state ^= state << 13;
    state ^= state >> 17;
    state ^= state << 5;
    return static_cast<float>(state) * 2.3283064365386962890625e-10f;
}

// samples_mapping.glsl:53-58
RV_HD f3 uniform_sphere(float u, float v)
{
    const float phi = kTwoPi * u;
    const float ct = (1.0f - v) - v;
    const float st = __builtin_sqrtf(fma_(-ct, ct, 1.0f));
    float sp, cp;
    sincos_det(phi, sp, cp);
    return mk(st * cp, st * sp, ct);
}

// material.glsl:223-226; every `x ± a*b` of the shader is one fma (the contraction rule of DESIGN.md §2, the one the
// reference's compiled shader is executed under in oracle/ref_spv)
RV_HD float fresnel(float cos_in, float cos_out, float eta)
{
    const float rs = fma_(eta, cos_in, -cos_out) / fma_(eta, cos_in, cos_out);
    const float rp = fma_(-eta, cos_out, cos_in) / fma_(eta, cos_out, cos_in);
    return 0.5f * fma_(rp, rp, rs * rs);
}

}  // namespace rv

/*
 * spv_shim.h — the hand-written part of oracle/_ref/libref_spv.so (TEST INFRASTRUCTURE, authoring container only).
 *
 * tools/spv2c.py turns every instruction of the reference's compiled shader (assets/shaders/compute_pass.comp.spv, the
 * binary rvpt.cpp:676-681 loads) into one C statement.  What SPIR-V itself delegates to the Vulkan implementation cannot
 * be generated; it is defined here, and this list is complete:
 *
 *   1. GLSL.std.450 extended instructions used by the module: FSign Sin Cos Tan Sqrt FMin UMin FMax FClamp FMix
 *      (component-wise, this file) and Length Cross Normalize (spv_shim_vec.h);
 *   2. OpDot and OpMatrixTimesVector (evaluation order is implementation-defined; spv_shim_vec.h);
 *   3. float -> int conversion (OpConvertFToS), constants from bit patterns; OpFDiv of two OpDot results (see
 *      shim_fdiv_dots below; every other OpFDiv is C's correctly rounded `/`);
 *   4. the storage-image model behind OpImageRead / OpImageWrite / OpImageQuerySize (rgba8 UNORM conversion of the
 *      Vulkan spec, or a float image so that radiance can be compared before quantisation);
 *   5. the resource bindings handed to the entry point.
 *
 * Precision choices are those of DESIGN.md §2 (the arithmetic specification both the CPU oracle and the HIP kernels
 * implement): IEEE-754 binary32 RNE; correctly rounded divide and sqrt; min/max = IEEE minNum/maxNum; sin/cos by 3-term
 * Cody-Waite reduction + fixed minimax polynomials (fused Horner steps inside the builtin); tan = sin/cos.  With
 * REF_SPV_FUSED undefined (the default) no other operation here or in the generated code is contracted.
 */
#ifndef SPV_SHIM_H
#define SPV_SHIM_H
#include <math.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float shim_f32_bits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline double shim_f64_bits(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }
static inline int32_t shim_f2i(float x) { return (int32_t)x; }  /* OpConvertFToS: round toward zero */
static inline void shim_unreachable(void) { abort(); }

/* OpFDiv of two OpDot results — there is exactly one in the module: t = dot(v0 - o, n) / dot(d, n), the ray/plane distance
 * of intersect_triangle_fast, the innermost operation of the path.  Vulkan asks 2.5 ULP of a quotient for a divisor in
 * [2^-126, 2^126] and nothing outside; this build forms it as the gfx950 kernel does: Markstein's sequence on the
 * hardware reciprocal,
 *     r0 = v_rcp_f32(b);  r = fma(fma(-b, r0, 1), r0, r0);  q = a*r;  result = fma(fma(-b, q, a), r, q)
 * v_rcp_f32 returns copysign(inf, b) for zero / subnormal b, copysign(0, b) when 1/b would be subnormal (|b| > 2^126) or b
 * is infinite, and is within 1 ulp otherwise; its refinement r equals the correctly rounded 1/b for EVERY b with
 * 2^-126 <= |b| <= 2^126 (all 2^23 mantissas x 253 exponents compared on the GPU: tools/microbench/rcp_probe.hip,
 * tests/test_gpu_parity.py::test_fast_division_model), which is how r is obtained here.  The result is the correctly rounded
 * a/b whenever no intermediate leaves the normal range (Markstein 1990; 10^11 random pairs on the GPU: 0 differences) — on
 * all of Vulkan's specified domain — and NaN / 0 where the hardware reciprocal flushes.  The fmas are part of the builtin
 * (as in sin/cos) and stay fused in both builds. */
static inline float shim_fdiv_dots(float a, float b)
{
#ifdef REF_SPV_IEEE_FDIV
    /* the strict reading: C's correctly rounded quotient (what OpFDiv means on an IEEE machine).  Built as a THIRD variant
     * (libref_spv_ieee.so / libref_spv_fused_ieee.so) only to prove that the choice above changes nothing observable: every
     * committed fixture re-rendered with it is bit-identical (tests/test_ref_spv.py::test_strict_ieee_quotient_changes_no_fixture) */
    return a / b;
#endif
    float r;
    if (b != b || (fabsf(b) >= 0x1p-126f && fabsf(b) <= 0x1p126f)) {
        r = 1.0f / b;
    } else {
        const float r0 = fabsf(b) < 0x1p-126f ? copysignf(INFINITY, b) : copysignf(0.0f, b);
        r = fmaf(fmaf(-b, r0, 1.0f), r0, r0);
    }
    const float q = a * r;
    return fmaf(fmaf(-b, q, a), r, q);
}

/* ---- 1. component-wise GLSL.std.450 ---- */
static inline float shim_fmin_f(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }
static inline float shim_fmax_f(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }
static inline double shim_fmin_d(double a, double b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }
static inline double shim_fmax_d(double a, double b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }
static inline uint32_t shim_umin_u(uint32_t a, uint32_t b) { return a < b ? a : b; }
static inline float shim_fclamp_f(float x, float lo, float hi) { return shim_fmin_f(shim_fmax_f(x, lo), hi); }
static inline float shim_fsign_f(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
static inline float shim_sqrt_f(float x) { return sqrtf(x); }
/* FMix: x*(1-a) + y*a, the GLSL specification's formula, left to right */
static inline float shim_fmix_f(float x, float y, float a)
{
#ifdef REF_SPV_FUSED
    return fmaf(y, a, x * (1.0f - a));
#else
    return x * (1.0f - a) + y * a;
#endif
}

/* sin / cos: q = floor(x*2/pi + 1/2); r = x - q*pi/2 in three fused steps (pi/2 split 1.5703125 + 4.837512969970703125e-4
 * + 7.54978995489188216e-8); degree-7 / degree-6 polynomials in r on [-pi/4, pi/4]; quadrant select. */
static inline void shim_sincos(float x, float* s, float* c)
{
    const float q = floorf(fmaf(x, 0.636619746685028076171875f, 0.5f));
    float r = fmaf(q, -1.5703125f, x);
    r = fmaf(q, -4.837512969970703125e-4f, r);
    r = fmaf(q, -7.54978995489188216e-8f, r);
    const float z = r * r;
    float ps = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = fmaf(z, ps, -1.6666654611e-1f);
    ps = fmaf(ps * z, r, r);
    float pc = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = fmaf(z, pc, 4.166664568298827e-2f);
    pc = fmaf(pc * z, z, fmaf(z, -0.5f, 1.0f));
    const int quadrant = (int)q & 3;
    *s = quadrant == 0 ? ps : quadrant == 1 ? pc : quadrant == 2 ? -ps : -pc;
    *c = quadrant == 0 ? pc : quadrant == 1 ? -ps : quadrant == 2 ? -pc : ps;
}
#ifdef REF_SPV_LIBM
/* REF_SPV_LIBM: another admissible driver — the C library's (correctly rounded or nearly so) sinf / cosf / tanf */
static inline float shim_sin_f(float x) { return sinf(x); }
static inline float shim_cos_f(float x) { return cosf(x); }
static inline float shim_tan_f(float x) { return tanf(x); }
#else
static inline float shim_sin_f(float x) { float s, c; shim_sincos(x, &s, &c); return s; }
static inline float shim_cos_f(float x) { float s, c; shim_sincos(x, &s, &c); return c; }
static inline float shim_tan_f(float x) { float s, c; shim_sincos(x, &s, &c); return s / c; }
#endif

/* ---- 4. storage images (compute_pass.comp:41-42 declares both rgba8) ---- */
typedef struct {
    int32_t width, height;
    float* texels; /* RGBA, row-major, top row first; always holds the value a later imageLoad returns */
    int32_t unorm8; /* 1: VK_FORMAT_R8G8B8A8_UNORM store conversion (clamp, *255, round to nearest, NaN -> 0) */
} shim_image;

static inline float shim_unorm8(float x)
{
    if (!(x == x)) return 0.0f;
    const float cl = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
    return floorf(cl * 255.0f + 0.5f) / 255.0f;
}

/* ---- 5. descriptor set 0 ---- */
typedef struct {
    const void* binding[8]; /* 0 RenderSettings, 1 result image, 2 temporal image, 3 Random, 4 Camera, 5 nodes, 6 triangles, 7 materials */
    uint32_t length[8];     /* element counts of the runtime arrays (OpArrayLength) */
} shim_bindings;

#endif

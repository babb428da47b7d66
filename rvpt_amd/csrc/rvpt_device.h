// rvpt_device.h — device-side building blocks shared by the frame kernels (rvpt_kernels.hip, rvpt_packets.hip,
// rvpt_bvh4.hip, rvpt_bvh8.hip): wave intrinsics, the prepared-triangle test (intersection.glsl:267-323), the slab test
// (intersection.glsl:327-357), cameras and integrators (compute_pass.comp, camera.glsl, integrators.glsl), the pixel epilogue
// and the per-wave work pool.  Everything is internal linkage (anonymous namespace): each translation unit gets its own copy.
// Arithmetic follows DESIGN.md "Arithmetic specification" (rvpt_math.h); compile with -ffp-contract=off -fno-slp-vectorize.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rvpt_kernels.h"
#include "rvpt_math.h"

#ifndef RV_REPORT_STACK_OVERFLOW
#define RV_REPORT_STACK_OVERFLOW 0  // 1 in the debug build of the library (rvpt_amd/build.py: build_native_debug -> librvpt_hip_debug.so): measured -3.4 % on C3 / -1.5 % on
                                    // C4 geometry in the release kernels for a branch that is never taken (profiles/r05_bvh4_riders.txt), so the release build only clamps
#endif
#ifndef RV_PREFETCH_CLAIM
#define RV_PREFETCH_CLAIM 1
#endif

namespace rv {

namespace {

constexpr float kInf = __builtin_inff();

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
// number of set bits of `mask` below this lane
__device__ __forceinline__ uint32_t prefix_rank(uint64_t mask)
{
    return __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
}
__device__ __forceinline__ uint64_t ballot(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// OR of `v` over the 64 lanes of the wave (every lane must be active), as a wave-uniform value: four DPP steps leave the OR of each 16-lane row in all of
// its lanes (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror), row_bcast:15 / row_bcast:31 carry it down the rows, lane 63 holds the whole
__device__ __forceinline__ uint32_t wave_or(uint32_t v)
{
    v |= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0xB1, 0xF, 0xF, false));
    v |= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x4E, 0xF, 0xF, false));
    v |= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x141, 0xF, 0xF, false));
    v |= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x140, 0xF, 0xF, false));
    v |= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x142, 0xA, 0xF, false));
    v |= static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0x143, 0xC, 0xF, false));
    return __builtin_amdgcn_readlane(v, 63);
}

// One prepared triangle = 4 x float4:
//   q0 = (v0.x, v0.y, v0.z, n.x)   q1 = (n.y, n.z, e0.x, e0.y)
//   q2 = (e0.z, e1.x, e1.y, e1.z)  q3 = (a00, a01, a11, inv_det)
struct PrepTri {
    f3 v0, n, e0, e1;
    float a00, a01, a11, inv_det;
};
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ PrepTri unpack(const v4f q0, const v4f q1, const v4f q2, const v4f q3)
{
    PrepTri t;
    t.v0 = mk(q0.x, q0.y, q0.z);
    t.n = mk(q0.w, q1.x, q1.y);
    t.e0 = mk(q1.z, q1.w, q2.x);
    t.e1 = mk(q2.y, q2.z, q2.w);
    t.a00 = q3.x;
    t.a01 = q3.y;
    t.a11 = q3.z;
    t.inv_det = q3.w;
    return t;
}

// Ray-dependent half of intersect_triangle_fast (intersection.glsl:290-312) against the shrinking
// interval (0, closest).
__device__ __forceinline__ void test_triangle(const PrepTri &t, const f3 o, const f3 d, const uint32_t index,
                                              float &closest, uint32_t &hit)
{
    const float tt = div_dots(dot(t.v0 - o, t.n), dot(d, t.n));
    const f3 p0 = fma3(d, tt, o) - t.v0;
    const float b0 = dot(p0, t.e0);
    const float b1 = dot(p0, t.e1);
    const float u = t.inv_det * fma_(t.a01, b1, t.a00 * b0);
    const float v = t.inv_det * fma_(t.a11, b1, t.a01 * b0);
    const bool accept = (0.0f < tt) & (tt < closest) & (0.0f < u) & (0.0f < v) & (u + v < 1.0f);
    closest = accept ? tt : closest;
    hit = accept ? index : hit;
}

// The same test in two steps for the unrolled loops.  First everything that does not depend on closest_t:
// tt, m = min(tt, u, v) and s = u + v.  (0 < tt) & (0 < u) & (0 < v) is m > 0: a NaN among them is ignored by the
// minimum, but a NaN u or v makes s NaN and a NaN tt fails `tt < closest_t`, both part of the conjunction.
struct OpenTest {
    float tt, m, s;
};
__device__ __forceinline__ OpenTest test_triangle_open(const PrepTri &t, const f3 o, const f3 d)
{
    OpenTest r;
    r.tt = div_dots(dot(t.v0 - o, t.n), dot(d, t.n));
    const f3 p0 = fma3(d, r.tt, o) - t.v0;
    const float b0 = dot(p0, t.e0);
    const float b1 = dot(p0, t.e1);
    const float u = t.inv_det * fma_(t.a01, b1, t.a00 * b0);
    const float v = t.inv_det * fma_(t.a11, b1, t.a01 * b0);
    r.m = __builtin_fminf(__builtin_fminf(r.tt, u), v);
    r.s = u + v;
    return r;
}
// ... then the interval test and the update, skipped for the whole packet when no lane accepts (the common case:
// most triangles are not a new closest hit for any of the 64 rays)
__device__ __forceinline__ void accept_hit(const OpenTest r, const uint32_t index, float &closest, uint32_t &hit)
{
    const bool accept = (r.m > 0.0f) & (r.s < 1.0f) & (r.tt < closest);
    if (ballot(accept) != 0) {
        asm volatile("" ::: "memory");  // keep this a (wave-uniform) branch: if-converted it is two selects per test again
        closest = accept ? r.tt : closest;
        hit = accept ? index : hit;
    }
}

// `count` consecutive prepared triangles starting at record `src` (LDS), indices index0, index0 + 1, ...: U tests'
// arithmetic is scheduled together (the empty asm consumes all their results, so none of it sinks behind the first
// update branch), then the U updates follow in order.
template <int U>
__device__ __forceinline__ void intersect_run(const v4f *src, const uint32_t index0, const uint32_t count, const f3 o, const f3 d,
                                              float &closest, uint32_t &hit)
{
    static_assert(U == 2 || U == 4, "the scheduling barrier below is written out for 2 and 4");
    uint32_t i = 0;
    for (; i + U <= count; i += U) {
        OpenTest r[U];
#pragma unroll
        for (uint32_t k = 0; k < U; ++k) {
            const uint32_t j = i + k;
            r[k] = test_triangle_open(unpack(src[4 * j + 0], src[4 * j + 1], src[4 * j + 2], src[4 * j + 3]), o, d);
        }
        if (U == 4)
            asm volatile("" ::"v"(r[0].tt), "v"(r[0].m), "v"(r[0].s), "v"(r[1].tt), "v"(r[1].m), "v"(r[1].s), "v"(r[U - 2].tt), "v"(r[U - 2].m),
                         "v"(r[U - 2].s), "v"(r[U - 1].tt), "v"(r[U - 1].m), "v"(r[U - 1].s));
        else
            asm volatile("" ::"v"(r[0].tt), "v"(r[0].m), "v"(r[0].s), "v"(r[1].tt), "v"(r[1].m), "v"(r[1].s));
#pragma unroll
        for (uint32_t k = 0; k < U; ++k) accept_hit(r[k], index0 + i + k, closest, hit);
    }
    for (; i < count; ++i)
        accept_hit(test_triangle_open(unpack(src[4 * i + 0], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]), o, d), index0 + i, closest, hit);
}

struct Lane {
    f3 o, d;          // current segment
    f3 thr, col;      // path throughput / radiance so far
    f3 sum;           // sum of finished samples of this pixel
    uint32_t rng;
    uint32_t work;    // tile-linear accumulator index of the pixel
    uint32_t gx, gy;  // pixel coordinates
    int sample;       // finished samples
    int bounce;       // finished segments of the current path
    uint32_t nseg;    // segments traced by this lane so far (statistics)
    // GENERIC kernels only (render modes other than Kajiya): integrator of this pixel and its continuation state
    int mode, phase;
    int ao_i;
    float ao_acc;
    f3 aux_o, aux_n;
};

enum Phase { PH_MAIN = 0, PH_SHADOW = 1, PH_AO = 2, PH_COOK_LAST = 3 };

// compute_pass.comp:151-156 + camera.glsl:29-51
__device__ __forceinline__ void begin_sample(Lane &L, const FrameParams &p)
{
    const float r0 = rand01(L.rng);
    const float r1 = rand01(L.rng);
    const float cx = (static_cast<float>(L.gx) + r0) * p.inv_w;
    const float cy = 1.0f - (static_cast<float>(L.gy) + r1) * p.inv_h;
    const float u = p.aspect * ((cx + cx) - 1.0f);
    const float v = (cy + cy) - 1.0f;
    const f3 c0 = mk(p.cam[0], p.cam[1], p.cam[2]);
    const f3 c1 = mk(p.cam[3], p.cam[4], p.cam[5]);
    const f3 c2 = mk(p.cam[6], p.cam[7], p.cam[8]);
    const f3 c3 = mk(p.cam[9], p.cam[10], p.cam[11]);
    // M * vec4(u, v, w, 0): all four column terms, the last one `+ c3*0` included (it turns a -0 sum into +0, as the
    // reference's OpMatrixTimesVector does)
    L.d = normalize(fma3(c3, 0.0f, fma3(c2, p.cam_w, fma3(c1, v, c0 * u))));
    L.o = c3;
    L.thr = mk(1.0f, 1.0f, 1.0f);
    L.col = mk(0.0f, 0.0f, 0.0f);
    L.bounce = 0;
}

// compute_pass.comp:102-118 — all three cameras (camera.glsl:29-99) + per-integrator initial state
__device__ __forceinline__ void begin_sample_generic(Lane &L, const FrameParams &p)
{
    if (p.camera_mode == 0) {
        begin_sample(L, p);
    } else {
        const float r0 = rand01(L.rng);
        const float r1 = rand01(L.rng);
        const float cx = (static_cast<float>(L.gx) + r0) * p.inv_w;
        const float cy = 1.0f - (static_cast<float>(L.gy) + r1) * p.inv_h;
        const f3 c0 = mk(p.cam[0], p.cam[1], p.cam[2]);
        const f3 c1 = mk(p.cam[3], p.cam[4], p.cam[5]);
        const f3 c2 = mk(p.cam[6], p.cam[7], p.cam[8]);
        const f3 c3 = mk(p.cam[9], p.cam[10], p.cam[11]);
        if (p.camera_mode == 1) {  // ortho: origin = M*(s*u, s*v, 0, 1), direction = M[2].xyz (not normalised)
            const float u = p.aspect * ((cx + cx) - 1.0f);
            const float v = (cy + cy) - 1.0f;
            L.o = fma3(c3, 1.0f, fma3(c2, 0.0f, fma3(c1, p.ortho_scale * v, c0 * (p.ortho_scale * u))));
            L.d = c2;
        } else {  // spherical: direction = M*(unit_spherical(phi,theta).xzy, 0) (not normalised)
            float sp, cp, st, ct;
            sincos_det(cx * kTwoPi, sp, cp);
            sincos_det(cy * kPi, st, ct);
            const f3 l = mk(st * cp, ct, st * sp);
            L.o = c3;
            L.d = fma3(c3, 0.0f, fma3(c2, l.z, fma3(c1, l.y, c0 * l.x)));
        }
        L.thr = mk(1.0f, 1.0f, 1.0f);
        L.bounce = 0;
    }
    L.phase = PH_MAIN;
    const float ambient = (L.mode == 7) ? 0.1f : 0.0f;  // integrator_Whitted starts from ambient (integrators.glsl:293)
    L.col = mk(ambient, ambient, ambient);
}

// compute_pass.comp:134-144
__device__ __forceinline__ int select_mode(const FrameParams &p, const uint32_t gx, const uint32_t gy)
{
    const float psx = static_cast<float>(gx) * p.inv_w, psy = static_cast<float>(gy) * p.inv_h;
    int idx = p.modes[0];
    if (psy > p.split_y)
        idx = (psx < p.split_x) ? p.modes[2] : p.modes[3];
    else if (psx > p.split_x)
        idx = p.modes[1];
    return idx;
}

// Where shade() finds the hit triangle's normal and material: HBM/L2 (streamed / large-scene BVH kernels) or the
// LDS copies of the resident kernels.
struct ShadeSrc {
    const float4 *prep;
    const uint32_t *mat_index;
    const float4 *mats;
    const float4 *unit_n;  // FrameParams::unit_n (HBM; a 16-byte gather per hit through the vector cache — the table is n_tris x 16 B)
};

// One iteration of integrator_Kajiya's loop body after the closest hit is known
// (integrators.glsl:576-671, intersect_scene's normalisation intersection.glsl:511-513).
// Returns true when the path ended; `radiance` is then its value.

// `leave` (optional): for the packet kernel's bounce cull — where the NEXT segment leaves from: 2 * hit + s, s = 0 when it leaves the hit triangle on the side
// its record's normal n = cross(e0, e1) points to, 1 on the other side.  Lambert, mirror and a reflecting dielectric leave on the side the ray came from
// (origin pos + eps N', direction N' + S resp. dir_in + 2 cos_in N': dot(direction, N') = 1 + dot(S, N') >= 0 resp. cos_in >= 0, N' = the normal turned against the incoming ray);
// a refracting dielectric leaves on the other (origin pos - eps N', dot(direction, N') = -cos_out <= 0).
__device__ __forceinline__ bool shade(Lane &L, const FrameParams &p, const ShadeSrc src, const uint32_t hit, const float t_hit,
                                      f3 &radiance, uint32_t *leave = nullptr)
{
    if (hit == 0xFFFFFFFFu) {
        const float s = fma_(L.d.y, 0.5f, 0.5f);
        const float oms = 1.0f - s;
        const f3 bg = mk(fma_(0.2f, s, oms), fma_(0.3f, s, oms), fma_(0.7f, s, oms));
        radiance = fma3(L.thr, bg, L.col);
        return true;
    }
    // the hit triangle's unit normal: normalize(n) of its prepared record, computed by prepare_triangles (one IEEE sqrt and divide per TRIANGLE instead of per hit: the
    // two expansions were 25 of this function's ~90 VALU instructions, most of them of the classes that issue at half rate — profiles/r06_valu_issue_probe.txt)
    const float4 un = src.unit_n[hit];
    const uint32_t mi = src.mat_index[hit];
    const float4 albedo = src.mats[3 * mi + 0];
    const float4 emission = src.mats[3 * mi + 1];
    const float4 data = src.mats[3 * mi + 2];

    f3 normal = mk(un.x, un.y, un.z);
    const f3 pos = fma3(L.d, t_hit, L.o);
    L.col = fma3(L.thr, mk(emission.x, emission.y, emission.z), L.col);

    const f3 dir_in = normalize(L.d);
    const float cos_view = dot(dir_in, normal);
    float cos_in;
    float eta = albedo.w;
    if (cos_view > 0.0f) {
        cos_in = cos_view;
        normal = -normal;
    } else {
        cos_in = -cos_view;
        eta = data.w;  // 1.0f / eta, divided once per material (prepare_materials)
    }
    const f3 base = mk(albedo.x, albedo.y, albedo.z);
    const int type = static_cast<int>(data.x);
    f3 pos_out, dir_out;
    bool other_side = cos_view > 0.0f;  // N' = -normalize(n): the ray came from the side n points away from
    bool side_known = true;             // ... and the side the next segment leaves on is provable from the arithmetic below (else: no bounce cull for it)
    if (type == 0) {
        pos_out = fma3(normal, kEpsilon, pos);
        const float u = rand01(L.rng);
        const float v = rand01(L.rng);
        dir_out = normal + uniform_sphere(u, v);
        // dot(dir_out, N') = |dir_out|^2 / 2 + (|N'|^2 - |S|^2) / 2 and both are unit only to ~1e-7: when S all but cancels N' the sum's direction is rounding noise
        // and may point below the plane.  |dir_out|^2 >= 2^-16 keeps the normal component >= 2^-17 - 2e-7 > 0; one path in 2^18 gives up its cull.
        side_known = dot(dir_out, dir_out) >= 0x1p-16f;  // (NaN: false)
        L.thr = L.thr * ((base * kInvPi) * kPi);
    } else if (type == 1) {
        pos_out = fma3(normal, kEpsilon, pos);
        dir_out = fma3(normal, cos_in + cos_in, dir_in);
        L.thr = L.thr * base;
    } else if (type == 2) {
        const float k = fma_(-cos_in, cos_in, 1.0f);
        const float c2 = fma_(-(eta * eta), k, 1.0f);
        float cos_out = 0.0f;
        bool refl = (c2 <= 0.0f);
        if (!refl) {
            cos_out = __builtin_sqrtf(__builtin_fmaxf(0.0f, c2));
            const float f = fresnel(cos_in, cos_out, eta);
            refl = rand01(L.rng) < f;
        }
        if (refl) {
            pos_out = fma3(normal, kEpsilon, pos);
            dir_out = fma3(normal, cos_in + cos_in, dir_in);
        } else {
            pos_out = fma3(normal, -kEpsilon, pos);
            dir_out = fma3(normal, fma_(eta, cos_in, -cos_out), dir_in * eta);
            other_side = !other_side;
            // dot(dir_out, N') = -cos_out + eta (cos_in - dot(-dir_in, N')): the bracket is a rounding error of ~2^-22, harmless while eta is a refractive index
            // and not 1 / (an ior of 1e-6) (NaN / inf eta: false)
            side_known = eta <= 16.0f;
        }
        L.thr = L.thr * base;
    } else {
        radiance = mk(0.0f, 0.0f, 0.0f);
        return true;
    }
    L.o = pos_out;
    L.d = dir_out;
    L.bounce += 1;
    if (leave != nullptr) *leave = side_known ? 2u * hit + (other_side ? 1u : 0u) : 0xFFFFFFFFu;
    if (L.bounce >= p.max_bounces) {
        radiance = mk(0.0f, 0.0f, 0.0f);
        return true;
    }
    return false;
}

// ---- the other nine integrators (integrators.glsl:24-543), as continuations of the same closest-hit query ----
struct SurfaceHit {
    float t;
    f3 pos, normal, base, emissive;
    float ior;
    int type;
};
// intersect_scene's outputs (intersection.glsl:489-517); all zero (t = inf) on a miss
__device__ __forceinline__ SurfaceHit surface_at(const Lane &L, const ShadeSrc src, const uint32_t hit, const float t_hit)
{
    SurfaceHit h{};
    h.t = t_hit;
    if (hit != 0xFFFFFFFFu) {
        const float4 q0 = src.prep[4 * hit + 0];
        const float4 q1 = src.prep[4 * hit + 1];
        const uint32_t mi = src.mat_index[hit];
        const float4 albedo = src.mats[3 * mi + 0];
        const float4 emission = src.mats[3 * mi + 1];
        const float4 data = src.mats[3 * mi + 2];
        h.normal = normalize(mk(q0.w, q1.x, q1.y));
        h.pos = fma3(L.d, t_hit, L.o);
        h.base = mk(albedo.x, albedo.y, albedo.z);
        h.emissive = mk(emission.x, emission.y, emission.z);
        h.ior = albedo.w;
        h.type = static_cast<int>(data.x);
    }
    return h;
}
__device__ __forceinline__ f3 splat(const float x) { return mk(x, x, x); }
__device__ __forceinline__ f3 sky_mix(const float s)  // mix(white, blue, s), s unclamped
{
    const float oms = 1.0f - s;
    return mk(fma_(0.2f, s, oms), fma_(0.3f, s, oms), fma_(0.7f, s, oms));
}
// integrators.glsl:124,243,294: normalize(vec3(0.5, 1.0, 0.3)) is folded by glslang (in double precision); these are the
// three floats in the compiled shader's constant pool (0x3edd267b, 0x3f5d267b, 0x3e84b0b0)
__device__ __forceinline__ f3 light_direction() { return mk(0.4319342076778412f, 0.8638684153556824f, 0.25916051864624023f); }

// distance_functions.glsl:27-60 (distance from a point to a triangle) — sign() is 1/-1/0 (0 for NaN), clamp is
// min(max(x,0),1) with IEEE minNum/maxNum, `e*k - q` is fused per component
__device__ __forceinline__ float sign_(const float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
__device__ __forceinline__ float edge_dist2(const f3 e, const f3 q)
{
    const float k = __builtin_fminf(__builtin_fmaxf(dot(e, q) / dot(e, e), 0.0f), 1.0f);
    const f3 w = mk(fma_(e.x, k, -q.x), fma_(e.y, k, -q.y), fma_(e.z, k, -q.z));
    return dot(w, w);
}
__device__ __forceinline__ float distance_triangle(const f3 pt, const f3 a, const f3 b, const f3 c)
{
    const f3 ba = b - a, pa = pt - a;
    const f3 cb = c - b, pb = pt - b;
    const f3 ac = a - c, pc = pt - c;
    const f3 nor = cross(ba, ac);
    const float s = (sign_(dot(cross(ba, nor), pa)) + sign_(dot(cross(cb, nor), pb))) + sign_(dot(cross(ac, nor), pc));
    float m;
    if (s < 2.0f) {
        m = __builtin_fminf(__builtin_fminf(edge_dist2(ba, pa), edge_dist2(cb, pb)), edge_dist2(ac, pc));
    } else {
        const float dn = dot(nor, pa);
        m = (dn * dn) / dot(nor, nor);
    }
    return __builtin_sqrtf(m);
}
// integrator_Hart (integrators.glsl:681-693) over intersect_scene_st (distance_functions.glsl:70-116): sphere tracing
// of the primary ray against all triangles, MARCH_ITER 32, MARCH_EPS 0.1; the view is iterations / 31.  A debug
// heat map: it ignores the closest-hit query and marches per lane in global memory order.
__device__ __forceinline__ f3 hart(const Lane &L, const FrameParams &p)
{
    float t = 0.0f;
    f3 pt = fma3(L.d, t, L.o);
    int i = 0;
    for (; i < 32; ++i) {
        float best = kInf;
        for (uint32_t j = 0; j < p.n_tris; ++j) {
            const float4 a = p.tris[4 * j + 0], b = p.tris[4 * j + 1], c = p.tris[4 * j + 2];
            const float dist = distance_triangle(pt, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z));
            best = (best < dist) ? best : dist;  // min_idx keeps the old value only if it is smaller (:64-67)
        }
        const float min_radius = __builtin_fminf(kInf, best);
        if (min_radius < 0.1f || min_radius > kInf) break;
        t += min_radius;
        pt = fma3(L.d, min_radius, pt);
    }
    return splat(static_cast<float>(i) / 31.0f);
}

// Returns true when the sample is finished (`radiance` = its value); otherwise L.o/L.d hold the next query.
// A shadow / occlusion query only needs "was anything hit", which the closest-hit query answers identically to
// intersect_scene_any (both accept the same first triangle in traversal order before any interval shrinking).
__device__ __forceinline__ bool shade_generic(Lane &L, const FrameParams &p, const ShadeSrc src, const uint32_t hit, const float t_hit,
                                              f3 &radiance)
{
    const bool any = hit != 0xFFFFFFFFu;
    if (L.phase == PH_MAIN && L.mode == 9) return shade(L, p, src, hit, t_hit, radiance);
    if (L.phase == PH_SHADOW) {  // Appel :251-259, Whitted :341-349
        radiance = any ? L.col : L.thr;
        return true;
    }
    if (L.phase == PH_AO) {  // integrator_ao :191-205
        L.ao_acc += any ? 1.0f : 0.0f;
        L.ao_i += 1;
        if (L.ao_i < p.max_bounces) {
            const float u = rand01(L.rng);
            const float v = rand01(L.rng);
            L.o = L.aux_o;
            L.d = L.aux_n + uniform_sphere(u, v);
            return false;
        }
        radiance = splat(1.0f - L.ao_acc / static_cast<float>(p.max_bounces));
        return true;
    }
    const SurfaceHit h = surface_at(L, src, hit, t_hit);
    if (L.phase == PH_COOK_LAST) {  // integrator_Cook :473-479
        radiance = any ? fma3(L.thr, h.emissive, L.col) : fma3(L.thr, sky_mix(L.d.y), L.col);
        return true;
    }
    switch (L.mode) {
    case 0:  // binary :24-38
        radiance = splat(any ? 1.0f : 0.0f);
        return true;
    case 1:  // color :42-60
        radiance = any ? h.base : splat(0.0f);
        return true;
    case 2:  // depth :64-84
        radiance = splat(1.0f / (__builtin_sqrtf(dot(L.d, L.d)) * h.t));
        return true;
    case 3: {  // normal :88-105
        const float half_isect = 0.5f * (any ? 1.0f : 0.0f);
        radiance = mk(fma_(0.5f, h.normal.x, half_isect), fma_(0.5f, h.normal.y, half_isect), fma_(0.5f, h.normal.z, half_isect));
        return true;
    }
    case 4: {  // Utah :109-155
        if (!any) {
            radiance = sky_mix(L.d.y);
            return true;
        }
        const f3 col = splat(0.1f) + h.emissive;
        const f3 n = (dot(L.d, h.normal) < 0.0f) ? h.normal : -h.normal;
        const float cos_light = __builtin_fmaxf(0.0f, dot(light_direction(), n));
        radiance = fma3(h.base, cos_light, col);
        return true;
    }
    case 5: {  // ambient occlusion :159-208
        if (!any) {
            radiance = splat(0.0f);
            return true;
        }
        const f3 n = (dot(L.d, h.normal) < 0.0f) ? h.normal : -h.normal;
        L.aux_n = n;
        L.aux_o = fma3(n, kEpsilon, h.pos);
        L.ao_acc = 0.0f;
        L.ao_i = 0;
        if (p.max_bounces <= 0) {  // the loop never runs: 1 - 0/0
            radiance = splat(1.0f - L.ao_acc / static_cast<float>(p.max_bounces));
            return true;
        }
        const float u = rand01(L.rng);
        const float v = rand01(L.rng);
        L.o = L.aux_o;
        L.d = n + uniform_sphere(u, v);
        L.phase = PH_AO;
        return false;
    }
    case 6: {  // Appel :212-263
        if (!any) {
            radiance = splat(1.0f);
            return true;
        }
        const f3 dir_in = normalize(L.d);
        const f3 n = (dot(dir_in, h.normal) > 0.0f) ? -h.normal : h.normal;
        const f3 l = light_direction();
        L.col = splat(0.0f);                                   // result if the light is blocked
        L.thr = splat(__builtin_fmaxf(0.0f, dot(l, n)));       // result if it is visible
        L.o = fma3(n, kEpsilon, h.pos);
        L.d = l;
        L.phase = PH_SHADOW;
        return false;
    }
    case 7:    // Whitted :267-403
    case 8: {  // Cook :407-543
        if (!any) {
            radiance = fma3(L.thr, sky_mix(L.d.y), L.col);
            return true;
        }
        L.col = fma3(L.thr, h.emissive, L.col);
        const f3 dir_in = normalize(L.d);
        f3 normal = h.normal;
        const float cos_view = dot(dir_in, normal);
        float cos_in, eta = h.ior;
        if (cos_view > 0.0f) {
            cos_in = cos_view;
            normal = -normal;
        } else {
            cos_in = -cos_view;
            eta = 1.0f / eta;
        }
        f3 pos_out, dir_out;
        if (h.type == 0) {
            if (L.mode == 7) {  // direct Lambert: shadow ray towards the directional light
                const f3 l = light_direction();
                const float cos_light = __builtin_fmaxf(0.0f, dot(l, normal));
                L.thr = fma3(L.thr * h.base, cos_light, L.col);  // value if lit; L.col is the value if shadowed
                L.o = fma3(normal, kEpsilon, h.pos);
                L.d = l;
                L.phase = PH_SHADOW;
            } else {  // one more diffuse bounce, then stop
                const float u = rand01(L.rng);
                const float v = rand01(L.rng);
                L.o = fma3(normal, kEpsilon, h.pos);
                L.d = normal + uniform_sphere(u, v);
                L.thr = L.thr * ((h.base * kInvPi) * kPi);
                L.phase = PH_COOK_LAST;
            }
            return false;
        } else if (h.type == 1) {
            pos_out = fma3(normal, kEpsilon, h.pos);
            dir_out = fma3(normal, cos_in + cos_in, dir_in);
            L.thr = L.thr * h.base;
        } else if (h.type == 2) {
            const float k = fma_(-cos_in, cos_in, 1.0f);
            const float c2 = fma_(-(eta * eta), k, 1.0f);
            float cos_out = 0.0f;
            bool refl = (c2 <= 0.0f);
            if (!refl) {
                cos_out = __builtin_sqrtf(__builtin_fmaxf(0.0f, c2));
                refl = rand01(L.rng) < fresnel(cos_in, cos_out, eta);
            }
            if (refl) {
                pos_out = fma3(normal, kEpsilon, h.pos);
                dir_out = fma3(normal, cos_in + cos_in, dir_in);
            } else {
                pos_out = fma3(normal, -kEpsilon, h.pos);
                dir_out = fma3(normal, fma_(eta, cos_in, -cos_out), dir_in * eta);
            }
            L.thr = L.thr * h.base;
        } else {
            radiance = splat(0.0f);
            return true;
        }
        L.o = pos_out;
        L.d = dir_out;
        L.bounce += 1;
        if (L.bounce >= p.max_bounces) {
            radiance = splat(0.0f);
            return true;
        }
        return false;
    }
    default:  // eval_integrator's default branch (compute_pass.comp:96-97)
        radiance = hart(L, p);
        return true;
    }
}

// dispatchers: the Kajiya/pinhole kernels (GENERIC = false) keep the lean code path
template <bool GENERIC>
__device__ __forceinline__ void begin_sample_t(Lane &L, const FrameParams &p)
{
    if (GENERIC)
        begin_sample_generic(L, p);
    else
        begin_sample(L, p);
}
template <bool GENERIC>
__device__ __forceinline__ bool shade_t(Lane &L, const FrameParams &p, const ShadeSrc src, const uint32_t hit, const float t_hit, f3 &radiance)
{
    return GENERIC ? shade_generic(L, p, src, hit, t_hit, radiance) : shade(L, p, src, hit, t_hit, radiance);
}
// does this lane need an intersection query this round?  Loop integrators (Whitted, Cook, Kajiya) with a
// non-positive bounce budget return black without tracing (integrators.glsl:298,440,574)
template <bool GENERIC>
__device__ __forceinline__ bool wants_trace(const Lane &L, const FrameParams &p)
{
    return GENERIC ? (L.mode < 7 || L.mode > 9 || p.max_bounces > 0) : (p.max_bounces > 0);
}

// rgba8 UNORM store followed by the next frame's load (compute_pass.comp:41-42): clamp to [0,1] (NaN -> 0),
// scale by 255, round half up, back to float as q/255.
__device__ __forceinline__ float unorm8_roundtrip(float f)
{
    f = (f > 0.0f) ? f : 0.0f;
    f = (f > 1.0f) ? 1.0f : f;
    return __builtin_floorf(fma_(f, 255.0f, 0.5f)) / 255.0f;
}
__device__ __forceinline__ f3 store_format(const f3 v, const uint32_t quantize)
{
    return quantize ? mk(unorm8_roundtrip(v.x), unorm8_roundtrip(v.y), unorm8_roundtrip(v.z)) : v;
}

// compute_pass.comp:161-166 on the FP32 tile-linear accumulator
__device__ __forceinline__ void finish_pixel(const Lane &L, const FrameParams &p)
{
    const float faa = static_cast<float>(p.aa);
    // sampled /= aa (compute_pass.comp:161); x / 1.0f is x for every x, NaN and infinities included, so one sample per pixel skips the three divisions
    const f3 sampled = (p.aa == 1) ? L.sum : mk(L.sum.x / faa, L.sum.y / faa, L.sum.z / faa);
    if (p.sample_out != nullptr) {  // decoupled: blend_accumulate finishes compute_pass.comp:162-166
        p.sample_out[L.work] = SampleRGB{sampled.x, sampled.y, sampled.z};
        return;
    }
    f3 prev = mk(0.0f, 0.0f, 0.0f);
    if (p.frame != 0u) {
        const float4 a = p.accum[L.work];
        prev = mk(a.x, a.y, a.z);
    }
    const f3 out = store_format(fma3(prev, p.cf, sampled) * p.inv_cf, p.quantize);
    p.accum[L.work] = make_float4(out.x, out.y, out.z, 0.0f);
}

// tile-linear work index -> pixel; false if the pixel lies outside the image (partial edge tiles)
__device__ __forceinline__ bool decode_work(const FrameParams &p, const uint32_t work, uint32_t &gx, uint32_t &gy)
{
    const uint32_t local_tile = work >> 8;
    const uint32_t in_tile = work & 255u;
    uint32_t tile_x, tile_y;
    // slot_tile (rvpt_kernels.h) with its three divisions by tiles_x as multiplications (FastDiv)
    const uint32_t slot = local_tile * p.tile_world + p.tile_rank;
    tile_y = fast_div(slot, p.div_tiles_x);
    const uint32_t rotated = slot - tile_y * p.tiles_x;
    const uint32_t shift = kTileShift * tile_y;
    const uint32_t shift_mod = shift - fast_div(shift, p.div_tiles_x) * p.tiles_x;
    const uint32_t unrot = rotated + p.tiles_x - shift_mod;  // in [1, 2 tiles_x)
    tile_x = unrot >= p.tiles_x ? unrot - p.tiles_x : unrot;
    gx = tile_x * 16u + (in_tile & 15u);
    gy = tile_y * 16u + (in_tile >> 4);
    return (gx < p.width) & (gy < p.height);
}

// Per-wave pool of claimed work indices + the ballot/mbcnt hand-out to lanes that need a pixel.
//
// Work is dealt in units of kUnit consecutive tile-linear indices (one 16-pixel tile row):
//   * every wave owns a static first chunk (no atomic, no thundering herd at kernel start);
//   * the rest is split evenly over kClaimShards counters (one L2 atomic word sustains only ~90 claims/us
//     chip-wide, one shared head would serialise 4096 waves).  A wave claims from its home shard and moves
//     on round-robin when a shard runs dry;
//   * the host sizes the static chunk and the claims from the work per wave (rvpt_abi.hip: plan_work); with
//     little work per wave everything is static (claims that shrink as a shard drains were measured no better);
//   * the next claim is issued one round ahead (lane 0's returning atomic stays in flight during the
//     intersect loop), so its latency is never waited for.
struct WavePool {
    uint32_t next = 0, end = 0;       // claimed work indices not yet handed to a lane
    uint32_t shard = 0, shards_dry = 0;
    uint32_t asked = 0;               // units requested by the in-flight claim
    bool exhausted = false;
    bool pending = false;
    bool first = true;
    unsigned long long ticket = 0;    // lane 0: value returned by the in-flight claim
};

__device__ __forceinline__ void claim_async(WavePool &pool, const FrameParams &p, const uint32_t lane)
{
    pool.asked = p.claim_units;
    if (lane == 0) pool.ticket = atomicAdd(&p.counter[kShardStride * pool.shard], static_cast<unsigned long long>(pool.asked));
    pool.pending = true;
}

// returns false when every shard is dry
template <bool REGEN>
__device__ __forceinline__ bool next_chunk(WavePool &pool, const FrameParams &p, const uint32_t lane, const uint32_t wave_id)
{
    const FrameParams &c = p;
    uint32_t unit0, units;
    if (pool.first) {
        pool.first = false;
        unit0 = wave_id * c.first_units;
        units = c.first_units;
        if (!REGEN) pool.exhausted = true;  // one-pixel-per-lane kernel: exactly one chunk per wave
        if (unit0 >= c.n_units) {
            pool.exhausted = true;
            return false;
        }
    } else {
        for (;;) {
            if (!pool.pending) claim_async(pool, p, lane);
            const uint32_t pos = uniform(static_cast<uint32_t>(pool.ticket));
            pool.pending = false;
            const uint32_t shard_begin = c.dyn_base + pool.shard * c.shard_len;
            const uint32_t shard_end = min(c.n_units, shard_begin + c.shard_len);
            unit0 = shard_begin + pos;
            units = pool.asked;
            if (pos < c.shard_len && unit0 < shard_end) {
                units = min(units, shard_end - unit0);
                break;
            }
            pool.shard = (pool.shard + 1u) % kClaimShards;
            if (++pool.shards_dry >= kClaimShards) {
                pool.exhausted = true;
                return false;
            }
        }
    }
    pool.next = unit0 * kUnit;
    pool.end = min(p.n_work, (unit0 + units) * kUnit);
    return true;
}

template <bool REGEN, bool GENERIC>
__device__ __forceinline__ void regenerate(WavePool &pool, const FrameParams &p, const uint32_t lane, const uint32_t wave_id,
                                           bool &have_pixel, bool &need_sample, Lane &L)
{
    bool need = !have_pixel;
    for (;;) {
        const uint64_t mask = ballot(need);
        if (mask == 0) break;
        uint32_t avail = pool.end - pool.next;
        if (avail == 0) {
            if (pool.exhausted || !next_chunk<REGEN>(pool, p, lane, wave_id)) break;
            avail = pool.end - pool.next;
        }
        const uint32_t rank = prefix_rank(mask);
        const uint32_t wanted = static_cast<uint32_t>(__builtin_popcountll(mask));
        if (need && rank < avail) {
            const uint32_t work = pool.next + rank;
            // a launch may cover several consecutive frames (rvpt_hip_dispatch_frames): work = frame offset * n_work_frame + pixel
            uint32_t frame_offset = 0, pixel = work;
            if (p.n_work_frame != p.n_work) {
                frame_offset = fast_div(work, p.div_work_frame);
                pixel = work - frame_offset * p.n_work_frame;
            }
            uint32_t gx, gy;
            if (decode_work(p, pixel, gx, gy)) {
                L.work = work;
                L.gx = gx;
                L.gy = gy;
                L.rng = wang_hash(gx + gy * p.width) + (p.frame + frame_offset);  // util.glsl:35-36
                L.sample = 0;
                L.sum = mk(0.0f, 0.0f, 0.0f);
                if (GENERIC) L.mode = select_mode(p, gx, gy);
                have_pixel = true;
                need_sample = true;
                need = false;
            }
        }
        pool.next += min(wanted, avail);
    }
    if (REGEN && RV_PREFETCH_CLAIM && !pool.pending && !pool.exhausted && !pool.first && (pool.end - pool.next) < 64u)
        claim_async(pool, p, lane);
}

// After a segment: fold a finished path into the pixel, finish the pixel after `aa` samples.
__device__ __forceinline__ void retire(Lane &L, const FrameParams &p, const bool path_done, const f3 radiance,
                                       bool &have_pixel, bool &need_sample)
{
    if (path_done) {
        L.sum = L.sum + radiance;
        L.sample += 1;
        if (L.sample < p.aa) {
            need_sample = true;
        } else {
            finish_pixel(L, p);
            have_pixel = false;
        }
    }
}

// The BVH kernels clamp a push at the top of the stack the host sized (upload_scene: from the tree, build_wide_nodes: from the wide tree) — as the reference's
// 64-entry stack would silently overflow (intersection.glsl:367).  If that bound were ever wrong the image would be wrong without a word: a wave that saw a lane
// push past the top says so in the word after the exited-wave counter (never reset by the kernels; rvpt_hip_wait reads and clears it when RVPT_HIP_DEBUG is set).
// (Reported at the clamp itself — a branch that is never taken — rather than carried in a flag to the wave's exit: the walk loops have no register to spare.
// DEBUG BUILD ONLY: RV_REPORT_STACK_OVERFLOW above.)
__device__ __forceinline__ void report_stack_overflow(const FrameParams &p, const bool overflowed)
{
#if RV_REPORT_STACK_OVERFLOW
    if (overflowed) atomicOr(&p.counter[kShardStride * kClaimShards + 1u], 1ull);
#endif
}

// Wave epilogue: optional statistics, then the exit ticket.  The last wave of the launch to leave
// zeroes every counter, so the next launch on the stream starts from 0 without a memset in between
// (no wave can still be claiming work once every wave has taken its exit ticket).
__device__ __forceinline__ void wave_exit(const FrameParams &p, const uint32_t lane, uint32_t nseg, uint32_t nsmp)
{
    if (p.stats != nullptr) {
        for (int off = 32; off > 0; off >>= 1) {
            nseg += __shfl_down(nseg, off, 64);
            nsmp += __shfl_down(nsmp, off, 64);
        }
        if (lane == 0) {
            unsigned long long *stripe = p.stats + kStatStride * ((blockIdx.x * (kBlock / 64u) + (threadIdx.x >> 6)) % kStatStripes);  // (rvpt_kernels.h: kStatStripes)
            atomicAdd(&stripe[0], static_cast<unsigned long long>(nseg));
            atomicAdd(&stripe[1], static_cast<unsigned long long>(nsmp));
        }
    }
    if (lane == 0) {
        const unsigned long long ticket = atomicAdd(&p.counter[kShardStride * kClaimShards], 1ull);
        if (ticket + 1ull == static_cast<unsigned long long>(p.n_waves)) {
            for (uint32_t s = 0; s <= kClaimShards; ++s) atomicExch(&p.counter[kShardStride * s], 0ull);
        }
    }
}


// intersect_aabb (intersection.glsl:327-357) against the interval (0, closest): true when the box is hit; `entry` = max(t_near, 0)
__device__ __forceinline__ bool slab_entry(const f3 o, const f3 inv, const float4 n0, const float4 n1, const float closest, float &entry)
{
    // bounds = {minx,maxx,miny,maxy,minz,maxz}: n0.zw = x, n1.xy = y, n1.zw = z  (intersection.glsl:341-355)
    const f3 f = mk((n0.w - o.x) * inv.x, (n1.y - o.y) * inv.y, (n1.w - o.z) * inv.z);
    const f3 n = mk((n0.z - o.x) * inv.x, (n1.x - o.y) * inv.y, (n1.z - o.z) * inv.z);
    const float t1 = __builtin_fminf(__builtin_fmaxf(f.x, n.x), __builtin_fminf(__builtin_fmaxf(f.y, n.y), __builtin_fmaxf(f.z, n.z)));
    const float t0 = __builtin_fmaxf(__builtin_fminf(f.x, n.x), __builtin_fmaxf(__builtin_fminf(f.y, n.y), __builtin_fminf(f.z, n.z)));
    entry = __builtin_fmaxf(t0, 0.0f);
    return __builtin_fminf(t1, closest) >= entry;
}

// slab test of one child of a wide node, its six bounds given one by one (rvpt_bvh4.hip, rvpt_bvh8.hip; intersect_aabb, intersection.glsl:327-357)
__device__ __forceinline__ bool slab_child(const f3 o, const f3 inv, const float minx, const float maxx, const float miny, const float maxy, const float minz,
                                           const float maxz, const float closest, float &entry)
{
    const f3 f = mk((maxx - o.x) * inv.x, (maxy - o.y) * inv.y, (maxz - o.z) * inv.z);
    const f3 n = mk((minx - o.x) * inv.x, (miny - o.y) * inv.y, (minz - o.z) * inv.z);
    const float t1 = __builtin_fminf(__builtin_fmaxf(f.x, n.x), __builtin_fminf(__builtin_fmaxf(f.y, n.y), __builtin_fmaxf(f.z, n.z)));
    const float t0 = __builtin_fmaxf(__builtin_fminf(f.x, n.x), __builtin_fmaxf(__builtin_fminf(f.y, n.y), __builtin_fminf(f.z, n.z)));
    entry = __builtin_fmaxf(t0, 0.0f);
    return __builtin_fminf(t1, closest) >= entry;
}

// ---- the leaf boxes of the packet kernel's bounce rounds (rvpt_vis.h has the construction and why the test is a superset test) ----
struct LeafRay {
    f3 inv, oi;
};
__device__ __forceinline__ float leaf_inv(const float x)
{
    const float r = __builtin_amdgcn_rcpf(x);  // 1 ulp; +-inf for zero and denormal x
    return __builtin_copysignf(__builtin_fminf(__builtin_fabsf(r), 0x1p60f), r);  // (minNum: a NaN becomes 2^60)
}
__device__ __forceinline__ LeafRay leaf_ray(const f3 o, const f3 d)
{
    LeafRay r;
    r.inv = mk(leaf_inv(d.x), leaf_inv(d.y), leaf_inv(d.z));
    r.oi = mk(o.x * r.inv.x, o.y * r.inv.y, o.z * r.inv.z);
    return r;
}
// box = (lo.xyz, hi.x) (hi.yz, -, -); true when the ray o + t d, t > 0, may come within the box
__device__ __forceinline__ bool leaf_slab(const LeafRay &r, const float4 b0, const float4 b1)
{
    const float ax = fma_(b0.x, r.inv.x, -r.oi.x), bx = fma_(b0.w, r.inv.x, -r.oi.x);
    const float ay = fma_(b0.y, r.inv.y, -r.oi.y), by = fma_(b1.x, r.inv.y, -r.oi.y);
    const float az = fma_(b0.z, r.inv.z, -r.oi.z), bz = fma_(b1.y, r.inv.z, -r.oi.z);
    const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fminf(az, bz));
    const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fmaxf(az, bz));
    return tf >= __builtin_fmaxf(tn, 0.0f);
}

// ---- the conservative slab test of the quantised wide walk (rvpt_bvh4.hip: trace_bvh4q; nodes: bvh_wide.cpp build_quant_nodes) ----
// The reference's test (intersect_aabb, intersection.glsl:327-357) computes per plane t_ref = fl(fl(b - o) * inv) on a child's exact bound b.  Under
// containment an INNER box only culls (bvh_wide.cpp), so for the children of a wide node any test that accepts whenever that one does will do — provided a
// leaf's own box is then tested exactly at its visit (slab_leaf).  This one knows a child's box as B = origin + q scale (q an 8-bit integer, scale a power of
// two, B_min <= b_min and B_max >= b_max exactly: the host verified every q), takes per axis the plane on the ray's near side by the SIGN of inv (no min / max
// pair) and evaluates in ray space
//     s' = scale inv (exact: a power of two),   c0 = fma(origin, inv, cn),   t' = fma(float(q), s', c0),   cn = c - m, cf = c + m, c = fl(-(o inv)),
//     m = 2^-19 (|o| + extent) |inv|,   extent >= every |origin| and |B| of the tree.
// Why near' <= near_ref and far' >= far_ref (u = 2^-24, M := (extent + |o|) |inv|): in real arithmetic T = (B - o) inv lies on the accepting side of
// (b - o) inv, because B does and the plane was picked by inv's sign; t_ref is within 2.01 u M of (b - o) inv (two roundings of a value of magnitude <= M);
// t' differs from T - m by four roundings (c, cn, c0, t') of values of magnitude <= 2 M + m each: at most 8.1 u M + 3 u m; and the computed m is >=
// 32 u M (1 - 3 u).  So near' <= T - m + 8.1 u M + 3 u m <= T - 23 u M <= near_ref, the far side is the mirror image.  Infinities: a direction component
// so small that scale inv could overflow (|inv| > 2^60; the host keeps scale <= 2^60) is treated as
// zero, inv_q = +-inf: then m = inf, every t' of the axis is +-inf on the accepting side or NaN, v_max3 / v_min3 drop a NaN operand, and the axis does
// not cull — still a superset.  The same happens when a product overflows.  (The exact tests — the root's box, a leaf's box — use the true inv.)
struct QuantRay {
    f3 cn, cf;
    bool inf_x, inf_y, inf_z;  // the axis' inv is treated as +-inf
    bool neg_x, neg_y, neg_z;
};
__device__ __forceinline__ f3 inv_q(const QuantRay &r, const f3 inv)
{
    return mk(r.inf_x ? __builtin_copysignf(kInf, inv.x) : inv.x, r.inf_y ? __builtin_copysignf(kInf, inv.y) : inv.y, r.inf_z ? __builtin_copysignf(kInf, inv.z) : inv.z);
}
__device__ __forceinline__ QuantRay quant_slab_setup(const f3 o, const f3 inv, const float extent)
{
    QuantRay r;
    const float big = 1.152921504606846976e+18f;  // 2^60
    r.inf_x = __builtin_fabsf(inv.x) > big, r.inf_y = __builtin_fabsf(inv.y) > big, r.inf_z = __builtin_fabsf(inv.z) > big;
    const f3 iq = inv_q(r, inv);
    const f3 c = mk(-(o.x * iq.x), -(o.y * iq.y), -(o.z * iq.z));
    const float k = 1.9073486328125e-06f;  // 2^-19
    const f3 m = mk(k * (__builtin_fabsf(o.x) + extent) * __builtin_fabsf(iq.x), k * (__builtin_fabsf(o.y) + extent) * __builtin_fabsf(iq.y),
                    k * (__builtin_fabsf(o.z) + extent) * __builtin_fabsf(iq.z));
    r.cn = mk(c.x - m.x, c.y - m.y, c.z - m.z);
    r.cf = mk(c.x + m.x, c.y + m.y, c.z + m.z);
    r.neg_x = (__float_as_uint(inv.x) >> 31) != 0u;  // the sign BIT: inv = -inf for d = -0 too
    r.neg_y = (__float_as_uint(inv.y) >> 31) != 0u;
    r.neg_z = (__float_as_uint(inv.z) >> 31) != 0u;
    return r;
}
// a node in ray space: per axis the scaled step, the near / far constants and the near / far byte words (byte k = child k)
struct QuantNode {
    f3 s, cn, cf;
    uint32_t wnx, wfx, wny, wfy, wnz, wfz;
};
__device__ __forceinline__ QuantNode quant_slab_node(const QuantRay &r, const f3 iq, const float4 qa, const float4 qb, const float4 qc)
{
    QuantNode n;
    n.s = mk(qa.w * iq.x, qb.x * iq.y, qb.y * iq.z);
    n.cn = mk(__builtin_fmaf(qa.x, iq.x, r.cn.x), __builtin_fmaf(qa.y, iq.y, r.cn.y), __builtin_fmaf(qa.z, iq.z, r.cn.z));
    n.cf = mk(__builtin_fmaf(qa.x, iq.x, r.cf.x), __builtin_fmaf(qa.y, iq.y, r.cf.y), __builtin_fmaf(qa.z, iq.z, r.cf.z));
    const uint32_t minx = __float_as_uint(qb.z), maxx = __float_as_uint(qb.w), miny = __float_as_uint(qc.x), maxy = __float_as_uint(qc.y),
                   minz = __float_as_uint(qc.z), maxz = __float_as_uint(qc.w);
    n.wnx = r.neg_x ? maxx : minx, n.wfx = r.neg_x ? minx : maxx;
    n.wny = r.neg_y ? maxy : miny, n.wfy = r.neg_y ? miny : maxy;
    n.wnz = r.neg_z ? maxz : minz, n.wfz = r.neg_z ? minz : maxz;
    return n;
}
template <int K> __device__ __forceinline__ float quant_byte(const uint32_t w)
{
    return static_cast<float>((w >> (8 * K)) & 0xFFu);  // v_cvt_f32_ubyteK
}
template <int K> __device__ __forceinline__ bool quant_slab_child(const QuantNode &n, const float closest, float &entry)
{
    const float t0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(quant_byte<K>(n.wnx), n.s.x, n.cn.x), __builtin_fmaf(quant_byte<K>(n.wny), n.s.y, n.cn.y)),
                                     __builtin_fmaf(quant_byte<K>(n.wnz), n.s.z, n.cn.z));
    const float t1 = __builtin_fminf(__builtin_fminf(__builtin_fmaf(quant_byte<K>(n.wfx), n.s.x, n.cf.x), __builtin_fmaf(quant_byte<K>(n.wfy), n.s.y, n.cf.y)),
                                     __builtin_fmaf(quant_byte<K>(n.wfz), n.s.z, n.cf.z));
    entry = __builtin_fmaxf(t0, 0.0f);
    return __builtin_fminf(t1, closest) >= entry;
}
// the exact test of a leaf's own box at its visit: q0 = (minx, maxx, miny, maxy), q1 = (minz, maxz, -, -) (bvh_wide.cpp: build_leaf_boxes)
__device__ __forceinline__ bool slab_leaf(const f3 o, const f3 inv, const float4 q0, const float4 q1, const float closest)
{
    float entry;
    return slab_child(o, inv, q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, closest, entry);
}

}  // namespace

}  // namespace rv

// rvpt_vis.h — the bounce cull's table (the packet kernel's bounce rounds, rvpt_packets.hip): one word of one row.  Host + device, double
// precision on the float records taken as exact, no GPU needed (rvpt_bounce_rows, tests/test_bounce_rows.py).
//
// Row 2 A + s, bit B = 0 only when triangle B lies WHOLLY behind the plane of triangle A as seen from side s (s = 0: the side A's normal n = cross(e0, e1)
// points to) by more than `margin`, and both triangles are well shaped (sin^2 of the angle between the edges >= 2^-6, finite, non-degenerate).  A segment that
// leaves A on side s — origin on A's plane up to the float error of a position, pushed EPSILON towards s, direction with a non-negative component towards s
// (rvpt_device.h: shade, which says so in `leave` and gives up the cull where it cannot prove it) — cannot be accepted by the float test against such a B: the
// accepted point lies within (33 eps / kappa_B)(t + S) of B (rvpt_rect.h has the bound), far less than the margin of 2^-10 scene scales upload_scene passes.
#pragma once

#include <stdint.h>

#include "rvpt_math.h"

namespace rv {

// prep: n x 16 floats, the prepared records q0 = (v0, n.x) q1 = (n.yz, e0.xy) q2 = (e0.z, e1) q3 = Gram terms (rvpt_device.h)
RV_HD uint32_t bounce_row_word(const float *prep, const uint32_t n, const uint32_t row, const uint32_t w, const double margin)
{
    const uint32_t A = row >> 1;
    const double side = (row & 1u) ? -1.0 : 1.0;
    auto edges_ok = [](const double *e0, const double *e1) {  // sin^2 of the angle between the edges >= 2^-6 (NaN / degenerate: false)
        const double a00 = e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2], a11 = e0[0] * e0[0] + e0[1] * e0[1] + e0[2] * e0[2];
        const double a01 = e0[0] * e1[0] + e0[1] * e1[1] + e0[2] * e1[2];
        return (a00 * a11 - a01 * a01) >= 0x1p-6 * (a00 * a11) && a00 * a11 > 0.0;
    };
    const float *a = prep + 16u * A;
    const double av0[3] = {a[0], a[1], a[2]}, an[3] = {a[3], a[4], a[5]}, ae0[3] = {a[6], a[7], a[8]}, ae1[3] = {a[9], a[10], a[11]};
    const double nn = __builtin_sqrt(an[0] * an[0] + an[1] * an[1] + an[2] * an[2]);
    const bool a_ok = edges_ok(ae0, ae1) && nn > 0.0 && margin > 0.0;
    uint32_t bits = 0u;
    for (uint32_t b = 0; b < 32u; ++b) {
        const uint32_t B = 32u * w + b;
        if (B >= n) break;
        const float *q = prep + 16u * B;
        const double v0[3] = {q[0], q[1], q[2]}, e0[3] = {q[6], q[7], q[8]}, e1[3] = {q[9], q[10], q[11]};
        bool behind = a_ok && edges_ok(e0, e1);
        for (int k = 0; k < 3 && behind; ++k) {  // the three vertices of the record's triangle: v0, v0 + e0, v0 + e1
            const double p[3] = {v0[0] + (k == 1 ? e0[0] : (k == 2 ? e1[0] : 0.0)) - av0[0], v0[1] + (k == 1 ? e0[1] : (k == 2 ? e1[1] : 0.0)) - av0[1],
                                 v0[2] + (k == 1 ? e0[2] : (k == 2 ? e1[2] : 0.0)) - av0[2]};
            const double dist = side * (p[0] * an[0] + p[1] * an[1] + p[2] * an[2]) / nn;
            behind = dist <= -margin;  // (NaN: false -> the triangle stays in the row)
        }
        if (!behind) bits |= 1u << b;
    }
    return bits;
}

// what float errors of positions scale with: the largest |coordinate| + the largest extent of the uploaded triangles (reference Triangle records: 16 floats,
// vertices at [0..2], [4..6], [8..10]); 0 when a coordinate is not finite or the scale is absurd (then there is no table)
inline double bounce_scene_scale(const float *tris, const size_t n_tris)
{
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, amax = 0.0;
    bool finite = true;
    for (size_t i = 0; i < n_tris; ++i)
        for (int v = 0; v < 3; ++v)
            for (int k = 0; k < 3; ++k) {
                const double x = tris[16 * i + 4 * v + k];
                finite = finite && (x - x == 0.0);
                lo[k] = x < lo[k] ? x : lo[k], hi[k] = x > hi[k] ? x : hi[k];
                amax = (x < 0.0 ? -x : x) > amax ? (x < 0.0 ? -x : x) : amax;
            }
    double ext = hi[0] - lo[0];
    ext = (hi[1] - lo[1]) > ext ? (hi[1] - lo[1]) : ext;
    ext = (hi[2] - lo[2]) > ext ? (hi[2] - lo[2]) : ext;
    const double scale = amax + ext;
    return (n_tris > 0 && finite && scale > 0x1p-60 && scale < 0x1p60) ? scale : 0.0;
}
// ---- the leaf boxes of the bounce rounds (round 6) -------------------------------------------------------------------------------------------------------
// Triangles kLeafTris k .. kLeafTris k + kLeafTris - 1 of the uploaded buffer (the caller's order: BVH-leaf order at the reference's seam, rvpt.cpp:83-86 — spatially
// coherent; any other order only makes the boxes loose) share one axis-aligned box: the bounds of their records' triangles (v0, v0 + e0, v0 + e1 and the uploaded
// vertices themselves), widened by M = 2^-9 (scene scale + 2 EPSILON) on every side.  In a bounce round the wave keeps a group of kLeafTris candidate triangles only
// if SOME lane's ray passes the conservative slab test of the group's box (rvpt_device.h: leaf_slab).  A superset test — why an accepted pair always passes:
// the float test accepts only when the exact ray comes within E <= (33 eps / kappa)(t + S) of the record's triangle (rvpt_rect.h); for kappa >= 2^-6 — a leaf with a
// triangle below that, degenerate or non-finite gets the infinite box — E <= 2^-13 (t + S) <= 2^-10.4 (scale + EPSILON) (t <= twice the scene, S = four 1-norms of
// points of the scene or EPSILON off it), i.e. a point Y of the exact ray with t_Y > 0 lies inside the box widened by 0.4 M; the slab test evaluates
// fma(b, inv, -fl(o inv)) with inv = v_rcp_f32(d) (1 ulp; |inv| clamped to 2^60: an axis the ray does not move along by more than 2^-51 of the scene), whose error is
// <= 2^-21 (|b| + |o|) |inv| — 2^-12 of the remaining slack 0.6 M |inv|; so every axis' computed interval contains t_Y and the test passes.  Lanes whose segment is not
// covered by a proof (leave == all ones: the aa loop's camera rays, a Lambert direction that all but cancelled, eta > 16) vote for every box.
#ifndef RV_LEAF_TRIS
#define RV_LEAF_TRIS 8
#endif
constexpr uint32_t kLeafTris = RV_LEAF_TRIS;  // 4 or 8: a nibble or a byte of a row word
static_assert(kLeafTris == 4 || kLeafTris == 8, "a leaf is a nibble or a byte of a 32-triangle row word");
constexpr double kLeafBoxMarginScales = 0x1p-9;

// out: 8 floats per group of `group` consecutive triangles (lo.xyz, hi.xyz, 0, 0); tris: reference Triangle records (16 floats each), n_tris <= kResidentMaxTris;
// scale = bounce_scene_scale(tris).  group = kLeafTris: the leaf boxes; group = 1: every triangle's own box, tested (second level) for the triangles of a leaf that
// some ray came near, before the triangle itself — 16 VALU instead of 38 for the three in four that no ray of the round comes near
inline void bounce_group_boxes(const float *tris, const size_t n_tris, const double scale, const size_t group, float *out)
{
    const size_t kLeafTris = group;  // (shadows the constant: the body below is written for "a leaf")
    const size_t n_leaves = (n_tris + kLeafTris - 1) / kLeafTris;
    const double M = kLeafBoxMarginScales * (scale + 2.0 * 0.005);
    const float inf = __builtin_inff();
    for (size_t l = 0; l < n_leaves; ++l) {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        bool ok = true;
        for (size_t i = l * kLeafTris; i < n_tris && i < (l + 1) * kLeafTris; ++i) {
            const float *t = tris + 16 * i;
            const float v0[3] = {t[0], t[1], t[2]};
            float e0[3], e1[3];
            for (int k = 0; k < 3; ++k) e0[k] = t[4 + k] - v0[k], e1[k] = t[8 + k] - v0[k];  // prepare_triangles' float edges (rvpt_kernels.hip)
            const double a00 = double(e1[0]) * e1[0] + double(e1[1]) * e1[1] + double(e1[2]) * e1[2], a11 = double(e0[0]) * e0[0] + double(e0[1]) * e0[1] + double(e0[2]) * e0[2];
            const double a01 = double(e0[0]) * e1[0] + double(e0[1]) * e1[1] + double(e0[2]) * e1[2];
            ok = ok && (a00 * a11 - a01 * a01) >= 0x1p-6 * (a00 * a11) && a00 * a11 > 0.0;  // (NaN: false)
            for (int k = 0; k < 3; ++k) {
                const double c[5] = {v0[k], t[4 + k], t[8 + k], double(v0[k]) + e0[k], double(v0[k]) + e1[k]};
                for (double x : c) {
                    ok = ok && (x - x == 0.0);
                    lo[k] = x < lo[k] ? x : lo[k], hi[k] = x > hi[k] ? x : hi[k];
                }
            }
        }
        float *b = out + 8 * l;
        for (int k = 0; k < 3; ++k) {
            // rounded OUTWARD to float (nextafter on the double-to-float conversion's wrong side is covered by the margin: 2^-24 of a coordinate against 2^-9 of the scale)
            b[k] = ok ? static_cast<float>(lo[k] - M) : -inf;
            b[3 + k] = ok ? static_cast<float>(hi[k] + M) : inf;
        }
        b[6] = b[7] = 0.0f;
    }
}

inline void bounce_leaf_boxes(const float *tris, const size_t n_tris, const double scale, float *out) { bounce_group_boxes(tris, n_tris, scale, kLeafTris, out); }

constexpr double kBounceMarginScales = 0x1p-10;  // the table's margin in scene scales: eight times the float error a position can carry under the launch-time premise
constexpr double kBounceCameraScales = 64.0;     // ... which is: the camera (the first segment's origin) no further than this many scene scales from the world origin

}  // namespace rv

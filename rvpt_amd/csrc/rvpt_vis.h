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
constexpr double kBounceMarginScales = 0x1p-10;  // the table's margin in scene scales: eight times the float error a position can carry under the launch-time premise
constexpr double kBounceCameraScales = 64.0;     // ... which is: the camera (the first segment's origin) no further than this many scene scales from the world origin

}  // namespace rv

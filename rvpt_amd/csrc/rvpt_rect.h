// rvpt_rect.h — the screen footprint of a triangle for the camera rounds of the packet kernel (rvpt_packets.hip): a conservative rectangle, in
// units of 16 x 4 pixel blocks, outside which NO camera ray of the launch can be accepted by the triangle test (intersection.glsl:267-323 as
// rvpt_device.h evaluates it).  All 64 camera rays of a round come from one block and one origin, so a triangle whose rectangle does not hold the
// block is skipped for the whole wave — a superset test: the image cannot change.  Host + device, double precision (the rectangle's own
// arithmetic then carries relative errors of 2^-50: nothing beside the margins below), no GPU needed (rvpt_camera_rects, tests/test_host_utils.py).
//
// What is bounded.  The float test accepts a pair (ray, triangle) when t = a / den', p = o + t d, u, v (barycentrics of the orthogonal projection
// of p onto the triangle's plane, through the Gram adjugate) satisfy 0 < t, 0 < u, 0 < v, u + v < 1 as COMPUTED.  With eps = 2^-24, the triangle
// T = {v0 + u e0 + v e1} of the float record taken as exact reals, S = |o|_1 + |v0 - o|_1 + |e0|_1 + |e1|_1, h = the camera's distance from the
// triangle's plane and kappa = det / (a00 a11) = sin^2 of the angle between the edges:
//   * p lies on the ray up to the rounding of the three fmas: |p - (o + t d)| <= sqrt(3) eps (|o| + t);
//   * p lies near the plane: dot(p - v0, n) = t den* - num* with the exact dot products of the float vectors; t = RN(a / den') and the float dot
//     products carry 3-4 roundings each, so the off-plane distance is <= eps (5 t + 5 |v0 - o| + 2 |o|);
//   * the computed (u, v) differ from the exact barycentrics of p's projection by <= c eps (|p - v0| + |e|) / (kappa |e|) — the cancellation in
//     a00 b0 + a01 b1 against det = kappa |e0|^2 |e1|^2 — i.e. by <= 32 eps (|e0| + |e1| + off-plane) / kappa in world units (c <= 16 generously;
//     the float normal n = cross(e0, e1) is off the true one by an angle <= 3 eps / sqrt(kappa), which moves p's distance from T's own plane by
//     <= 3 eps (|e0| + |e1|) / sqrt(kappa): inside the same term);
//   so the ray passes within E <= (33 eps / kappa) (t + S) of a point X of T.  X is at least max(h, t - E) from the camera, hence the angle between the
//   ray and the direction to X has sine <= THETA = 2^-18 (2 + S / h) / kappa  (33 eps < 2^-18; (t + S) / max(h, t / 2) <= 2 + S / h).
// Directions on screen make an angle alpha with the optical axis with cos alpha >= c_s = w / sqrt(w^2 + 1.01 aspect^2 + 1.01) (w = 1 / tan(fov / 2); the
// 1.01 covers the jitter reaching the next pixel).  For THETA <= c_s / 4 the slopes (x / z, y / z) of the two directions differ by <= 2 THETA / c_s^2,
// and a point of T behind the camera plane (angle >= 90 degrees) cannot be X at all.  The rectangle is therefore the bounding box of T's projection
// (clipped at the camera plane: an edge that crosses it runs off to infinity on the side the projected edge moves towards — the sign of
// d(x / z) / ds = (x' z - x z') / z^2 is constant along an edge), widened by 2 THETA / c_s^2 in slope, converted to pixels, rounded outward to
// blocks and widened by ONE more block on every side (the float direction of a pixel sample differs from the exact one by < 0.01 pixel; jitter of
// exactly 1.0).  Anything that breaks a premise — a record not `safe` (rvpt_early_out.h: camera_record), kappa < 2^-12, THETA > c_s / 4, a camera
// matrix that is not invertible in float range, NaN anywhere — gives the WHOLE SCREEN.  The claim is checked on the device for every pixel x every
// triangle x several jitters of every fixture scene and camera (rvpt_hip_selftest_camera_rects: accepted by the float test => inside the rectangle).
#pragma once

#include <stdint.h>

#include "rvpt_math.h"

namespace rv {

constexpr uint32_t kRectAll_lo = 0xFFFF0000u;  // x0 = 0, x1 = 65535
constexpr uint32_t kRectAll_hi = 0xFFFF0000u;  // y0 = 0, y1 = 65535
constexpr uint32_t kRectNone_lo = 0x00000001u;  // x0 = 1 > x1 = 0: no block
constexpr uint32_t kRectNone_hi = 0x00000001u;

// The numerator of a camera record (rvpt_early_out.h: camera_record): |dot(v0 - o, n)| when the record is SAFE for the division-free pre-test — and for the
// rectangle below — and NaN when it is not (the camera in the triangle's plane, absurd scales, NaN / inf anywhere); `neg` = the sign of the dot product.
RV_HD float camera_numerator(const f3 v0, const f3 n, const f3 o, bool &neg)
{
    const float num = dot(v0 - o, n);
    neg = __builtin_signbit(num) != 0;
    const float a = __builtin_fabsf(num);
    const float lo = 0x1p-60f, hi = 0x1p60f;
    const bool safe = (a >= lo) & (a <= hi) & (__builtin_fabsf(n.x) <= hi) & (__builtin_fabsf(n.y) <= hi) & (__builtin_fabsf(n.z) <= hi);
    return safe ? a : __builtin_nanf("");
}

struct RectCamera {
    double r0[3], r1[3], r2[3];  // rows of adj(M3) * sign(det): camera-space coordinates of a world vector, times |det|
    double o[3];
    double aspect, w;            // camera.params.x, 1 / tan(vfov / 2)
    double width, height;
    double cs2;                  // c_s^2
    bool ok;                     // false: every rectangle is the whole screen
};

// cam = FrameParams::cam (columns 0..2 of the camera-to-world matrix, then the origin), compute_pass.comp:44-49 + camera.glsl:29-51
RV_HD RectCamera rect_camera(const float *cam, const float aspect, const float cam_w, const uint32_t width, const uint32_t height)
{
    RectCamera c;
    const double c0[3] = {cam[0], cam[1], cam[2]}, c1[3] = {cam[3], cam[4], cam[5]}, c2[3] = {cam[6], cam[7], cam[8]};
    c.o[0] = cam[9], c.o[1] = cam[10], c.o[2] = cam[11];
    auto cross3 = [](const double *a, const double *b, double *r) {
        r[0] = a[1] * b[2] - a[2] * b[1];
        r[1] = a[2] * b[0] - a[0] * b[2];
        r[2] = a[0] * b[1] - a[1] * b[0];
    };
    cross3(c1, c2, c.r0);
    cross3(c2, c0, c.r1);
    cross3(c0, c1, c.r2);
    const double det = c0[0] * c.r0[0] + c0[1] * c.r0[1] + c0[2] * c.r0[2];
    const double sgn = det < 0.0 ? -1.0 : 1.0;
    for (int i = 0; i < 3; ++i) c.r0[i] *= sgn, c.r1[i] *= sgn, c.r2[i] *= sgn;
    c.aspect = aspect;
    c.w = cam_w;
    c.width = width;
    c.height = height;
    const double n0 = c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2], n1 = c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2], n2 = c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2];
    // the columns must be far from dependent and of sane size: |det| >= 2^-10 |c0| |c1| |c2| (the reference's matrices are rotations: det = 1)
    const double vol = __builtin_sqrt(n0 * n1 * n2);
    c.cs2 = (c.w * c.w) / (c.w * c.w + 1.01 * c.aspect * c.aspect + 1.01);
    c.ok = (det * sgn >= 0x1p-10 * vol) && (vol >= 0x1p-60) && (vol <= 0x1p60) && (c.aspect >= 0x1p-10) && (c.aspect <= 0x1p10) && (c.w >= 0x1p-6) && (c.w <= 0x1p10) &&
           (c.width >= 1.0) && (c.height >= 1.0) && (c.o[0] - c.o[0] == 0.0) && (c.o[1] - c.o[1] == 0.0) && (c.o[2] - c.o[2] == 0.0);
    // (comparisons with NaN are false: a NaN anywhere above leaves ok == false)
    return c;
}

RV_HD uint32_t rect_block(double v)  // outward-rounded block coordinate, clamped to the 16 bits of a field
{
    v = __builtin_floor(v);
    return v < 0.0 ? 0u : (v > 65535.0 ? 65535u : static_cast<uint32_t>(v));
}

// The rectangle of the triangle with prepared record (v0, n, e0, e1) — q0 = (v0, n.x), q1 = (n.yz, e0.xy), q2 = (e0.z, e1) — and camera-record
// numerator `a` (NaN = not safe).  lo = x0 | x1 << 16 in units of 16 pixels, hi = y0 | y1 << 16 in units of 4 rows; x0 > x1: no block at all.
RV_HD void camera_rect(const RectCamera &c, const float *q0, const float *q1, const float *q2, const float a, uint32_t &lo, uint32_t &hi)
{
    lo = kRectAll_lo, hi = kRectAll_hi;
    if (!c.ok || !(a == a)) return;
    const double v0[3] = {q0[0], q0[1], q0[2]}, n[3] = {q0[3], q1[0], q1[1]}, e0[3] = {q1[2], q1[3], q2[0]}, e1[3] = {q2[1], q2[2], q2[3]};
    auto dot3 = [](const double *x, const double *y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; };
    auto abs1 = [](const double *x) { return __builtin_fabs(x[0]) + __builtin_fabs(x[1]) + __builtin_fabs(x[2]); };
    const double vo[3] = {v0[0] - c.o[0], v0[1] - c.o[1], v0[2] - c.o[2]};
    // the error bound of the float test for this triangle (header comment): THETA = 2^-18 (2 + S / h) / kappa
    const double a00 = dot3(e1, e1), a11 = dot3(e0, e0), a01 = dot3(e0, e1);
    const double kappa = (a00 * a11 - a01 * a01) / (a00 * a11);
    const double nn = __builtin_sqrt(dot3(n, n));
    const double h = __builtin_fabs(dot3(vo, n)) / nn;
    const double S = abs1(c.o) + abs1(vo) + abs1(e0) + abs1(e1);
    const double theta = 0x1p-18 * (2.0 + S / h) / kappa;
    if (!(kappa >= 0x1p-12) || !(theta * theta * 16.0 <= c.cs2)) return;  // (NaN / inf: the comparisons fail -> whole screen)
    const double dslope = 2.0 * theta / c.cs2;

    // camera-space coordinates (times |det|: the projection is scale free) of the three vertices of T
    double q[3][3];
    for (int k = 0; k < 3; ++k) {
        const double p[3] = {vo[0] + (k == 1 ? e0[0] : (k == 2 ? e1[0] : 0.0)), vo[1] + (k == 1 ? e0[1] : (k == 2 ? e1[1] : 0.0)),
                             vo[2] + (k == 1 ? e0[2] : (k == 2 ? e1[2] : 0.0))};
        q[k][0] = dot3(p, c.r0), q[k][1] = dot3(p, c.r1), q[k][2] = dot3(p, c.r2);
    }
    const double inf = __builtin_inf();
    double sx0 = inf, sx1 = -inf, sy0 = inf, sy1 = -inf;  // bounding box of the projection in slopes (x / z, y / z)
    bool any_front = false;
    for (int k = 0; k < 3; ++k) {
        const double *A = q[k], *B = q[(k + 1) % 3];
        const bool fa = A[2] > 0.0, fb = B[2] > 0.0;
        if (fa) {
            any_front = true;
            const double sx = A[0] / A[2], sy = A[1] / A[2];
            sx0 = sx < sx0 ? sx : sx0, sx1 = sx > sx1 ? sx : sx1;
            sy0 = sy < sy0 ? sy : sy0, sy1 = sy > sy1 ? sy : sy1;
        }
        if (fa != fb) {  // the edge crosses the camera plane: its front part runs off to infinity, on the side its projection moves towards
            const double *F = fa ? A : B, *K = fa ? B : A;
            const double dx = K[0] - F[0], dy = K[1] - F[1], dz = K[2] - F[2];
            const double gx = dx * F[2] - F[0] * dz, tx = 0x1p-40 * (__builtin_fabs(dx * F[2]) + __builtin_fabs(F[0] * dz));
            const double gy = dy * F[2] - F[1] * dz, ty = 0x1p-40 * (__builtin_fabs(dy * F[2]) + __builtin_fabs(F[1] * dz));
            if (!(gx < -tx)) sx1 = inf;
            if (!(gx > tx)) sx0 = -inf;
            if (!(gy < -ty)) sy1 = inf;
            if (!(gy > ty)) sy0 = -inf;
        }
    }
    if (!any_front) {  // T lies behind the camera plane: no direction on screen comes within THETA of it
        lo = kRectNone_lo, hi = kRectNone_hi;
        return;
    }
    sx0 -= dslope, sx1 += dslope, sy0 -= dslope, sy1 += dslope;
    // slopes -> pixels: direction = (u, v, w) in camera space, u = aspect (2 cx - 1), v = 2 cy - 1, x = W cx, y = H (1 - cy)  (compute_pass.comp:151-156)
    const double px0 = c.width * 0.5 * (sx0 * c.w / c.aspect + 1.0), px1 = c.width * 0.5 * (sx1 * c.w / c.aspect + 1.0);
    const double py0 = c.height * 0.5 * (1.0 - sy1 * c.w), py1 = c.height * 0.5 * (1.0 - sy0 * c.w);
    if (!(px0 <= px1) || !(py0 <= py1)) return;  // NaN (inf - inf cannot occur: dslope is finite)
    // outward to blocks, one more block on every side.  A rectangle wholly off screen keeps x0 > x1 (or y0 > y1) only through the clamps below:
    // that is "no block", exactly what it should be
    const double bx0 = __builtin_floor(px0 / 16.0) - 1.0, bx1 = __builtin_floor(px1 / 16.0) + 1.0;
    const double by0 = __builtin_floor(py0 / 4.0) - 1.0, by1 = __builtin_floor(py1 / 4.0) + 1.0;
    if (bx1 < 0.0 || by1 < 0.0 || bx0 > 65535.0 || by0 > 65535.0) {
        lo = kRectNone_lo, hi = kRectNone_hi;
        return;
    }
    lo = rect_block(bx0) | (rect_block(bx1) << 16);
    hi = rect_block(by0) | (rect_block(by1) << 16);
}

// does the rectangle hold block (bx, by)?
RV_HD bool rect_holds(const uint32_t lo, const uint32_t hi, const uint32_t bx, const uint32_t by)
{
    return (bx >= (lo & 0xFFFFu)) & (bx <= (lo >> 16)) & (by >= (hi & 0xFFFFu)) & (by <= (hi >> 16));
}

}  // namespace rv

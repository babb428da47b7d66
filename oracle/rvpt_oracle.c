/*
 * rvpt_oracle.c — CPU restatement of RVPT's path-trace compute shader (compute_pass.comp: render modes 0-9,
 * all three cameras; mode 9 = Kajiya is the hot path).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (rvpt_amd/, include/, the C-ABI
 * library) may include, link or call this file; only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, as the checker.
 *
 * PARITY PIN: the reference ships no tests, golden images or known-answer vectors (.github/workflows/ci.yml only
 * compiles) and cannot be built here (Vulkan), but it ships the binary it executes, assets/shaders/compute_pass.comp.spv.
 * tools/spv2c.py translates that module instruction by instruction to C (oracle/ref_spv/, built into the never-shipped
 * oracle/_ref/); tests/golden/ref_spv/*.npz are ITS outputs — 111 images over every integrator, camera and material branch
 * plus direct calls of single functions on decision boundaries — and tests/test_ref_spv.py holds this file to them BIT
 * FOR BIT: built with -DORACLE_UNFUSED against the uncontracted execution, as shipped against the execution under the
 * one contraction rule of DESIGN.md §2.  Closed-form identities (tests/test_oracle_kat.py) pin the rest.
 *
 * Each function cites the reference lines it restates (paths relative to the reference tree,
 * shaders under assets/shaders/).  GLSL leaves float contraction and the precision of
 * sin/cos/tan/normalize to the driver; the choices made here are marked [CHOICE] and are the
 * arithmetic specification the HIP kernel implements independently (DESIGN.md "Arithmetic
 * specification").  All arithmetic is IEEE-754 binary32, round-to-nearest-even; an FMA appears
 * exactly where fmaf() is written (compile with -ffp-contract=off).
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* Contraction of the shader's own expressions.  GLSL lets the driver fuse a*b+c or not; the arithmetic specification of
 * this build fuses exactly where O_FMA is written.  -DORACLE_UNFUSED evaluates the same expressions with
 * a separate multiply and add: that build is compared bit for bit with the reference's compiled shader executed without
 * contraction (oracle/ref_spv, tests/test_ref_spv.py).  The polynomial kernels inside sin/cos and the UNORM8 conversion
 * are driver-level builtins and keep fmaf() in both builds (the same code is oracle/ref_spv/spv_shim.h's). */
#ifdef ORACLE_UNFUSED
#define O_FMA(a, b, c) ((a) * (b) + (c))
#else
#define O_FMA(a, b, c) fmaf((a), (b), (c))
#endif

/* compute_pass.comp:5-12 — the double literals rounded to float */
#define O_PI 3.14159274101257324219f
#define O_TWO_PI 6.28318548202514648438f
#define O_INV_PI 0.31830987334251403809f
#define O_EPSILON 0.005f
#define O_INF (__builtin_inff())

typedef struct {
    float x, y, z;
} v3;

/* structs.glsl:1-7 / geometry.h:76-111 */
typedef struct {
    float vert0[4], vert1[4], vert2[4], mat_id[4];
} OTriangle;
/* structs.glsl:9-14 / bvh.h:12-19 */
typedef struct {
    uint32_t first_child_or_primitive, primitive_count;
    float bounds[6];
} OBvhNode;
/* structs.glsl:22-33 / material.h:9-26 */
typedef struct {
    float albedo[4], emission[4], data[4];
} OMaterial;
/* compute_pass.comp:28-40 / rvpt.h:77-89 */
typedef struct {
    int32_t max_bounces, aa;
    uint32_t current_frame;
    int32_t camera_mode, top_left, top_right, bottom_left, bottom_right;
    float split_ratio[2];
} OSettings;

/* ray-independent per-triangle terms of intersect_triangle_fast (intersection.glsl:287-305) */
typedef struct {
    v3 v0, n, e0, e1;
    float a00, a01, a11, inv_det;
} OPrepTri;

/* ------------------------------------------------------------------------------------------ */
/* vector helpers — [CHOICE] dot = fma chain x,y,z; cross = mul,mul,sub; normalize = v*(1/sqrt) */

static inline v3 V(float x, float y, float z)
{
    v3 r = {x, y, z};
    return r;
}
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vscale(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline v3 vneg(v3 a) { return V(-a.x, -a.y, -a.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
/* a*s + b, fused per component */
static inline v3 vfma(v3 a, float s, v3 b)
{
    return V(O_FMA(a.x, s, b.x), O_FMA(a.y, s, b.y), O_FMA(a.z, s, b.z));
}
static inline float vdot(v3 a, v3 b) { return O_FMA(a.z, b.z, O_FMA(a.y, b.y, a.x * b.x)); }
static inline v3 vcross(v3 a, v3 b)
{
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline v3 vnormalize(v3 a)
{
    float inv = 1.0f / sqrtf(vdot(a, a));
    return vscale(a, inv);
}
/* IEEE-754 minNum/maxNum (what v_min_f32/v_max_f32 and fminf/fmaxf compute) */
static inline float o_min(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a < b ? a : b)); }
static inline float o_max(float a, float b) { return (a != a) ? b : ((b != b) ? a : (a > b ? a : b)); }

/* ------------------------------------------------------------------------------------------ */
/* [CHOICE] deterministic sin/cos for |x| <= ~8: 3-term Cody-Waite reduction by pi/2 and the
 * classic single-precision minimax polynomials on [-pi/4, pi/4], all Horner steps fused. */

#define O_TWO_OVER_PI 0.636619746685028076171875f
#define O_PIO2_HI 1.5703125f
#define O_PIO2_MID 4.837512969970703125e-4f
#define O_PIO2_LO 7.54978995489188216e-8f

static inline void o_sincos(float x, float *s_out, float *c_out)
{
    float kf = floorf(fmaf(x, O_TWO_OVER_PI, 0.5f));
    int k = (int)kf;
    float r = fmaf(kf, -O_PIO2_HI, x);
    r = fmaf(kf, -O_PIO2_MID, r);
    r = fmaf(kf, -O_PIO2_LO, r);
    float z = r * r;
    /* sin(r) = r + r*z*(S3 + z*(S2 + z*S1)) */
    float sp = fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f);
    sp = fmaf(z, sp, -1.6666654611e-1f);
    sp = fmaf(sp * z, r, r);
    /* cos(r) = 1 - z/2 + z*z*(C3 + z*(C2 + z*C1)) */
    float cp = fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f);
    cp = fmaf(z, cp, 4.166664568298827e-2f);
    cp = fmaf(cp * z, z, fmaf(z, -0.5f, 1.0f));
    switch (k & 3) {
    case 0: *s_out = sp; *c_out = cp; break;
    case 1: *s_out = cp; *c_out = -sp; break;
    case 2: *s_out = -sp; *c_out = -cp; break;
    default: *s_out = -cp; *c_out = sp; break;
    }
}
static inline float o_tan(float x)
{
    float s, c;
    o_sincos(x, &s, &c);
    return s / c;
}

/* ------------------------------------------------------------------------------------------ */
/* util.glsl:25-33 */
static inline uint32_t o_wang_hash(uint32_t seed)
{
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u;
    seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return seed;
}
/* util.glsl:38-50; uint->float is round-to-nearest-even [CHOICE], /2^32 is exact */
static inline float o_rand(uint32_t *state)
{
    uint32_t s = *state;
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    *state = s;
    return (float)s * 2.3283064365386962890625e-10f;
}

/* samples_mapping.glsl:53-58 */
static inline v3 o_map_uniform_sphere(float u, float v)
{
    float phi = O_TWO_PI * u;
    float ct = (1.0f - v) - v;
    float st = sqrtf(O_FMA(-ct, ct, 1.0f));
    float s, c;
    o_sincos(phi, &s, &c);
    return V(st * c, st * s, ct);
}

/* material.glsl:207-228 */
static inline float o_fresnel(float cos_in, float cos_out, float eta)
{
    float r_perp = O_FMA(eta, cos_in, -cos_out) / O_FMA(eta, cos_in, cos_out);
    float r_par = O_FMA(-eta, cos_out, cos_in) / O_FMA(eta, cos_out, cos_in);
    return 0.5f * O_FMA(r_par, r_par, r_perp * r_perp);
}

/* ------------------------------------------------------------------------------------------ */
/* intersection.glsl:287-305 — ray-independent part, computed once per triangle with the same
 * operations the shader performs per test */
static inline void o_prepare(const OTriangle *t, OPrepTri *p)
{
    v3 v0 = V(t->vert0[0], t->vert0[1], t->vert0[2]);
    v3 v1 = V(t->vert1[0], t->vert1[1], t->vert1[2]);
    v3 v2 = V(t->vert2[0], t->vert2[1], t->vert2[2]);
    p->v0 = v0;
    p->e0 = vsub(v1, v0);
    p->e1 = vsub(v2, v0);
    p->n = vcross(p->e0, p->e1);
    /* mat2(dot(e1,e1), -dot(e0,e1), -dot(e0,e1), dot(e0,e0)), column-major (:302) */
    p->a00 = vdot(p->e1, p->e1);
    p->a01 = -vdot(p->e0, p->e1);
    p->a11 = vdot(p->e0, p->e0);
    /* :305 — A_adj[0][0]*A_adj[1][1] - A_adj[0][1]*A_adj[1][0] */
    p->inv_det = 1.0f / O_FMA(-p->a01, p->a01, p->a00 * p->a11);
}

/* [CHOICE] The one quotient of the intersect loop, t = dot(v0 - o, n) / dot(d, n) (intersection.glsl:292), is formed the way
 * the gfx950 kernel forms it: r0 = v_rcp_f32(b); r = fma(fma(-b, r0, 1), r0, r0); q = a*r; t = fma(fma(-b, q, a), r, q)
 * (Markstein).  v_rcp_f32 gives copysign(inf, b) for zero / subnormal b and copysign(0, b) for |b| > 2^126 or infinite b;
 * for every other b the refined r IS the correctly rounded 1/b (checked for all of them on the GPU,
 * tools/microbench/rcp_probe.hip), hence `1.0f / b` below.  Equal to the IEEE quotient whenever no intermediate leaves
 * the normal range — the domain on which Vulkan specifies division at all; NaN or 0 where the reciprocal flushes (such
 * a t is never accepted).  The fmas belong to the builtin and stay fused in the ORACLE_UNFUSED build (as in sin/cos);
 * oracle/ref_spv/spv_shim.h has the same function for the reference shader's OpFDiv(OpDot, OpDot). */
static inline float o_div_dots(float a, float b)
{
    float r;
    if (b != b || (fabsf(b) >= 0x1p-126f && fabsf(b) <= 0x1p126f)) {
        r = 1.0f / b;
    } else {
        float r0 = fabsf(b) < 0x1p-126f ? copysignf(O_INF, b) : copysignf(0.0f, b);
        r = fmaf(fmaf(-b, r0, 1.0f), r0, r0);
    }
    float q = a * r;
    return fmaf(fmaf(-b, q, a), r, q);
}

/* intersection.glsl:290-312 — ray-dependent part.  Returns accept; *t_out,*u_out,*v_out always
 * written. */
static inline int o_tri_test(v3 o, v3 d, const OPrepTri *p, float mint, float maxt, float *t_out,
                             float *u_out, float *v_out)
{
    float t = o_div_dots(vdot(vsub(p->v0, o), p->n), vdot(d, p->n)); /* :292 */
    v3 pos = vfma(d, t, o);                                 /* :293 */
    v3 p0 = vsub(pos, p->v0);                               /* :296 */
    float b0 = vdot(p0, p->e0), b1 = vdot(p0, p->e1);       /* :299 */
    float u = p->inv_det * O_FMA(p->a01, b1, p->a00 * b0);   /* :308, row 0 of A_adj*b */
    float v = p->inv_det * O_FMA(p->a11, b1, p->a01 * b0);   /* :308, row 1 */
    *t_out = t;
    *u_out = u;
    *v_out = v;
    return (mint < t) && (t < maxt) && (0.0f < u) && (0.0f < v) && (u + v < 1.0f); /* :311 */
}

/* intersection.glsl:341-355 with invdir hoisted (same value every node); the shader's double
 * temporaries hold exactly-widened floats, so float arithmetic is value-identical */
static inline int o_aabb_test(v3 o, v3 invdir, v3 bmin, v3 bmax, float mint, float maxt)
{
    v3 f = vmul(vsub(bmax, o), invdir);
    v3 n = vmul(vsub(bmin, o), invdir);
    v3 tmax = V(o_max(f.x, n.x), o_max(f.y, n.y), o_max(f.z, n.z));
    v3 tmin = V(o_min(f.x, n.x), o_min(f.y, n.y), o_min(f.z, n.z));
    float t1 = o_min(tmax.x, o_min(tmax.y, tmax.z));
    float t0 = o_max(tmin.x, o_max(tmin.y, tmin.z));
    t0 = o_max(t0, mint);
    t1 = o_min(t1, maxt);
    return t1 >= t0;
}

/* the same slab test, also returning the entry distance t0 (ordered traversal below) */
static inline int o_aabb_entry(v3 o, v3 invdir, const float b[6], float mint, float maxt, float *entry)
{
    v3 bmin = V(b[0], b[2], b[4]), bmax = V(b[1], b[3], b[5]);
    v3 f = vmul(vsub(bmax, o), invdir);
    v3 n = vmul(vsub(bmin, o), invdir);
    float t1 = o_min(o_max(f.x, n.x), o_min(o_max(f.y, n.y), o_max(f.z, n.z)));
    float t0 = o_max(o_min(f.x, n.x), o_max(o_min(f.y, n.y), o_min(f.z, n.z)));
    t0 = o_max(t0, mint);
    t1 = o_min(t1, maxt);
    *entry = t0;
    return t1 >= t0;
}

typedef struct {
    const OBvhNode *nodes;
    size_t n_nodes;
    const OTriangle *tris;
    const OPrepTri *prep;
    size_t n_tris;
    const OMaterial *mats;
    size_t n_mats;
    int traversal; /* 0 = bvh (intersect_bvh), 1 = brute force, 2 = bvh with ordered children (build-defined) */
} OScene;

/* Closest hit.  Returns triangle index or -1; *t_hit = closest t.
 * brute: build-defined variant of the dead intersect_triangles loop (intersection.glsl:708-752):
 *        triangles 0..N-1 in buffer order with the shrinking (mint, closest_t) interval.
 * bvh:   intersection.glsl:361-413 (left child first, 64-entry stack, sentinel ~0). */
static long o_closest_hit(const OScene *sc, v3 o, v3 d, float mint, float maxt, float *t_hit)
{
    float closest = maxt;
    long hit = -1;
    float t, u, v;
    if (sc->traversal == 1) {
        for (size_t i = 0; i < sc->n_tris; ++i) {
            if (o_tri_test(o, d, &sc->prep[i], mint, closest, &t, &u, &v)) {
                closest = t;
                hit = (long)i;
            }
        }
    } else if (sc->traversal == 2) {
        /* Ordered traversal — the reference's own "TODO: Order the children on the stack" (intersection.glsl:405),
         * build-defined: the slab tests of both children happen at the parent, the nearer child (smaller entry
         * distance, left on ties) is visited first, the farther one is pushed and re-tested against the then
         * current closest_t when popped.  Same triangle test, same leaf order. */
        uint32_t stack[64];
        int sp = 0;
        v3 invdir = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        float e0, e1;
        if (sc->n_nodes && o_aabb_entry(o, invdir, sc->nodes[0].bounds, mint, closest, &e0)) {
            uint32_t cur = 0;
            for (;;) {
                const OBvhNode *nd = &sc->nodes[cur];
                uint32_t first = nd->first_child_or_primitive;
                int descend = 0;
                if (nd->primitive_count > 0) {
                    for (uint32_t i = first, n = first + nd->primitive_count; i < n; ++i) {
                        if (o_tri_test(o, d, &sc->prep[i], mint, closest, &t, &u, &v)) {
                            closest = t;
                            hit = (long)i;
                        }
                    }
                } else {
                    int h0 = o_aabb_entry(o, invdir, sc->nodes[first].bounds, mint, closest, &e0);
                    int h1 = o_aabb_entry(o, invdir, sc->nodes[first + 1].bounds, mint, closest, &e1);
                    if (h0 && h1) {
                        if (e1 < e0) {
                            stack[sp++] = first;
                            cur = first + 1;
                        } else {
                            stack[sp++] = first + 1;
                            cur = first;
                        }
                        descend = 1;
                    } else if (h0 || h1) {
                        cur = h0 ? first : first + 1;
                        descend = 1;
                    }
                }
                if (descend) continue;
                int found = 0;
                while (sp > 0) { /* pop until a node still passes its slab test against the current closest_t */
                    uint32_t n = stack[--sp];
                    if (o_aabb_entry(o, invdir, sc->nodes[n].bounds, mint, closest, &e0)) {
                        cur = n;
                        found = 1;
                        break;
                    }
                }
                if (!found) break;
            }
        }
    } else {
        uint32_t stack[64];
        int sp = 0;
        v3 invdir = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); /* :341 */
        stack[sp++] = 0xFFFFFFFFu;
        uint32_t top = 0;
        while (top != 0xFFFFFFFFu) {
            const OBvhNode *nd = &sc->nodes[top];
            v3 bmin = V(nd->bounds[0], nd->bounds[2], nd->bounds[4]);
            v3 bmax = V(nd->bounds[1], nd->bounds[3], nd->bounds[5]);
            if (!o_aabb_test(o, invdir, bmin, bmax, mint, closest)) {
                top = stack[--sp];
                continue;
            }
            uint32_t first = nd->first_child_or_primitive;
            if (nd->primitive_count > 0) {
                for (uint32_t i = first, n = first + nd->primitive_count; i < n; ++i) {
                    if (o_tri_test(o, d, &sc->prep[i], mint, closest, &t, &u, &v)) {
                        closest = t;
                        hit = (long)i;
                    }
                }
                top = stack[--sp];
            } else {
                stack[sp++] = first + 1;
                top = first;
            }
        }
    }
    *t_hit = closest;
    return hit;
}

/* camera.glsl:29-51; w = 1/tan(0.5*hfov) is frame-constant and passed in */
/* (M * vec4(x, y, z, w)).xyz — OpMatrixTimesVector, [CHOICE] (((c0*x + c1*y) + c2*z) + c3*w), one fma per column.  The terms
 * that multiply a literal 0 are kept: `s + c3*0` turns a -0 sum into +0 (found by running the compiled shader's spherical
 * camera, oracle/ref_spv) */
static inline v3 o_mat4_vec4(const float cam[20], float x, float y, float z, float w)
{
    float r[3];
    for (int k = 0; k < 3; ++k) {
        float s = cam[k] * x;
        s = O_FMA(cam[4 + k], y, s);
        s = O_FMA(cam[8 + k], z, s);
        s = O_FMA(cam[12 + k], w, s);
        r[k] = s;
    }
    return V(r[0], r[1], r[2]);
}

static inline void o_pinhole_ray(const float cam[20], float w, float x, float y, v3 *org, v3 *dir)
{
    float aspect = cam[16];
    float u = aspect * ((x + x) - 1.0f);
    float v = (y + y) - 1.0f;
    *org = V(cam[12], cam[13], cam[14]);
    *dir = vnormalize(o_mat4_vec4(cam, u, v, w, 0.0f)); /* camera.glsl:46-48 */
}

/* integrators.glsl:547-677 (+ intersect_scene, intersection.glsl:489-517) */
static v3 o_kajiya(const OScene *sc, v3 org, v3 dir, float mint, float maxt, int nbounce,
                   uint32_t *rng, uint64_t *segments)
{
    v3 col = V(0, 0, 0), thr = V(1, 1, 1);
    const v3 blue = V(0.2f, 0.3f, 0.7f);
    for (int i = 0; i < nbounce; ++i) {
        float t;
        ++*segments;
        long hit = o_closest_hit(sc, org, dir, mint, maxt, &t);
        if (hit < 0) {
            /* :578-579 — mix(white, blue, s) = white*(1-s) + blue*s, s unclamped, dir unnormalised */
            float s = O_FMA(dir.y, 0.5f, 0.5f);
            float oms = 1.0f - s;
            v3 bg = V(O_FMA(blue.x, s, oms), O_FMA(blue.y, s, oms), O_FMA(blue.z, s, oms));
            return V(O_FMA(thr.x, bg.x, col.x), O_FMA(thr.y, bg.y, col.y), O_FMA(thr.z, bg.z, col.z));
        }
        const OPrepTri *pt = &sc->prep[hit];
        const OMaterial *m = &sc->mats[(int)sc->tris[hit].mat_id[0]]; /* intersection.glsl:398-399 */
        int type = (int)m->data[0];                                  /* :52 */
        v3 base = V(m->albedo[0], m->albedo[1], m->albedo[2]);
        v3 emis = V(m->emission[0], m->emission[1], m->emission[2]);
        float ior = m->albedo[3]; /* :54 */

        v3 normal = vnormalize(pt->n); /* intersection.glsl:511 */
        v3 pos = vfma(dir, t, org);    /* intersection.glsl:513 */

        col = V(O_FMA(thr.x, emis.x, col.x), O_FMA(thr.y, emis.y, col.y), O_FMA(thr.z, emis.z, col.z)); /* :582 */

        v3 dir_in = vnormalize(dir); /* :587 */
        float cos_view = vdot(dir_in, normal);
        float cos_in, eta = ior;
        if (cos_view > 0.0f) { /* :598-609 */
            cos_in = cos_view;
            normal = vneg(normal);
        } else {
            cos_in = -cos_view;
            eta = 1.0f / eta;
        }
        v3 pos_out, dir_out;
        switch (type) {
        case 0: { /* Lambert :617-623 */
            pos_out = vfma(normal, O_EPSILON, pos);
            float u = o_rand(rng);
            float v = o_rand(rng); /* material.glsl:106, left-to-right */
            dir_out = vadd(normal, o_map_uniform_sphere(u, v));
            thr = vmul(thr, vscale(vscale(base, O_INV_PI), O_PI)); /* material.glsl:90 */
            break;
        }
        case 1: /* mirror :625-631 */
            pos_out = vfma(normal, O_EPSILON, pos);
            dir_out = vfma(normal, cos_in + cos_in, dir_in);
            thr = vmul(thr, base);
            break;
        case 2: { /* dielectric :633-665 */
            float k = O_FMA(-cos_in, cos_in, 1.0f);
            float c2 = O_FMA(-(eta * eta), k, 1.0f);
            float cos_out = 0.0f;
            int refl = (c2 <= 0.0f);
            if (!refl) {
                cos_out = sqrtf(o_max(0.0f, c2));
                float f = o_fresnel(cos_in, cos_out, eta);
                refl = (o_rand(rng) < f);
            }
            if (refl) {
                pos_out = vfma(normal, O_EPSILON, pos);
                dir_out = vfma(normal, cos_in + cos_in, dir_in);
            } else {
                pos_out = vfma(normal, -O_EPSILON, pos);
                dir_out = vfma(normal, O_FMA(eta, cos_in, -cos_out), vscale(dir_in, eta));
            }
            thr = vmul(thr, base);
            break;
        }
        default: /* :666-667 */
            return V(0, 0, 0);
        }
        org = pos_out;
        dir = dir_out;
    }
    return V(0, 0, 0); /* :674-675 */
}

/* ------------------------------------------------------------------------------------------ */
/* util.glsl:77-96 — (sin(theta)*cos(phi), sin(theta)*sin(phi), cos(theta)) */
static inline v3 o_unit_spherical(float phi, float theta)
{
    float sp, cp, st, ct;
    o_sincos(phi, &sp, &cp);
    o_sincos(theta, &st, &ct);
    return V(st * cp, st * sp, ct);
}
/* camera.glsl:55-76 — origin = M*(scale*u, scale*v, 0, 1), direction = M[2].xyz (not normalised) */
static inline void o_ortho_ray(const float cam[20], float x, float y, v3 *org, v3 *dir)
{
    float scale = cam[18], aspect = cam[16];
    float u = aspect * ((x + x) - 1.0f);
    float v = (y + y) - 1.0f;
    float su = scale * u, sv = scale * v;
    *org = o_mat4_vec4(cam, su, sv, 0.0f, 1.0f); /* camera.glsl:71 */
    *dir = V(cam[8], cam[9], cam[10]);
}
/* camera.glsl:80-99 — direction = M * (unit_spherical(phi,theta).xzy, 0), not normalised */
static inline void o_spherical_ray(const float cam[20], float x, float y, v3 *org, v3 *dir)
{
    float phi = x * O_TWO_PI;
    float theta = y * O_PI;
    v3 s = o_unit_spherical(phi, theta);
    v3 l = V(s.x, s.z, s.y); /* .xzy */
    *org = V(cam[12], cam[13], cam[14]);
    *dir = o_mat4_vec4(cam, l.x, l.y, l.z, 0.0f);
}
/* compute_pass.comp:102-118 */
static inline void o_camera_ray(int mode, const float cam[20], float w, float x, float y, v3 *org, v3 *dir)
{
    if (mode == 0)
        o_pinhole_ray(cam, w, x, y, org, dir);
    else if (mode == 1)
        o_ortho_ray(cam, x, y, org, dir);
    else
        o_spherical_ray(cam, x, y, org, dir);
}

/* intersect_scene (intersection.glsl:489-517): closest hit with normalised normal, position and material */
typedef struct {
    int hit;
    float t;          /* INF on a miss */
    v3 pos, normal;   /* zero on a miss */
    v3 base, emissive;
    float ior;
    int type;
} OHit;
static OHit o_scene_hit(const OScene *sc, v3 org, v3 dir, float mint, float maxt, uint64_t *segments)
{
    OHit h;
    memset(&h, 0, sizeof h);
    ++*segments;
    long i = o_closest_hit(sc, org, dir, mint, maxt, &h.t);
    h.hit = i >= 0;
    if (!h.hit) {
        h.t = O_INF;
        return h;
    }
    const OMaterial *m = &sc->mats[(int)sc->tris[i].mat_id[0]];
    h.type = (int)m->data[0];
    h.base = V(m->albedo[0], m->albedo[1], m->albedo[2]);
    h.emissive = V(m->emission[0], m->emission[1], m->emission[2]);
    h.ior = m->albedo[3];
    h.normal = vnormalize(sc->prep[i].n);
    h.pos = vfma(dir, h.t, org);
    return h;
}
/* intersect_scene_any -> intersect_bvh_any (intersection.glsl:417-485): true at the first accepted triangle in
 * traversal order, no interval shrinking; brute force: first accepted triangle in buffer order */
static int o_scene_any(const OScene *sc, v3 o, v3 d, float mint, float maxt, uint64_t *segments)
{
    float t, u, v;
    ++*segments;
    if (sc->traversal == 1) {
        for (size_t i = 0; i < sc->n_tris; ++i)
            if (o_tri_test(o, d, &sc->prep[i], mint, maxt, &t, &u, &v)) return 1;
        return 0;
    }
    if (sc->traversal == 2) /* ordered mode has no separate any-hit walk: "did the closest-hit query find anything" */
        return o_closest_hit(sc, o, d, mint, maxt, &t) >= 0;
    uint32_t stack[64];
    int sp = 0;
    v3 invdir = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    stack[sp++] = 0xFFFFFFFFu;
    uint32_t top = 0;
    while (top != 0xFFFFFFFFu) {
        const OBvhNode *nd = &sc->nodes[top];
        v3 bmin = V(nd->bounds[0], nd->bounds[2], nd->bounds[4]);
        v3 bmax = V(nd->bounds[1], nd->bounds[3], nd->bounds[5]);
        if (!o_aabb_test(o, invdir, bmin, bmax, mint, maxt)) {
            top = stack[--sp];
            continue;
        }
        uint32_t first = nd->first_child_or_primitive;
        if (nd->primitive_count > 0) {
            for (uint32_t i = first, n = first + nd->primitive_count; i < n; ++i)
                if (o_tri_test(o, d, &sc->prep[i], mint, maxt, &t, &u, &v)) return 1;
            top = stack[--sp];
        } else {
            stack[sp++] = first + 1;
            top = first;
        }
    }
    return 0;
}

static inline v3 o_splat(float x) { return V(x, x, x); }
/* mix(white, blue, s) = white*(1-s) + blue*s, s unclamped; [CHOICE] fma(blue, s, 1-s) */
static inline v3 o_sky(float s)
{
    float oms = 1.0f - s;
    return V(O_FMA(0.2f, s, oms), O_FMA(0.3f, s, oms), O_FMA(0.7f, s, oms));
}
static inline v3 o_madd(v3 a, v3 b, v3 c) { return V(O_FMA(a.x, b.x, c.x), O_FMA(a.y, b.y, c.y), O_FMA(a.z, b.z, c.z)); }
/* integrators.glsl:124,243,294 — `normalize(vec3(0.5, 1.0, 0.3))` is a constant expression: glslang folded it (in double
 * precision) and the compiled shader holds these three floats (0x3edd267b, 0x3f5d267b, 0x3e84b0b0); a float normalize at
 * run time gives 0x3e84b0b1 for z.  Found by executing the compiled shader (oracle/ref_spv). */
static inline v3 o_light_dir(void) { return V(0.4319342076778412f, 0.8638684153556824f, 0.25916051864624023f); }

/* the normal flip + relative ior shared by Whitted / Cook / Kajiya (e.g. integrators.glsl:305-327) */
typedef struct {
    v3 dir_in, normal;
    float cos_in, eta;
} OFrame;
static inline OFrame o_frame(v3 dir, const OHit *h)
{
    OFrame f;
    f.dir_in = vnormalize(dir);
    f.normal = h->normal;
    float cos_view = vdot(f.dir_in, f.normal);
    f.eta = h->ior;
    if (cos_view > 0.0f) {
        f.cos_in = cos_view;
        f.normal = vneg(f.normal);
    } else {
        f.cos_in = -cos_view;
        f.eta = 1.0f / f.eta;
    }
    return f;
}
/* mirror / dielectric branches shared by Whitted, Cook and Kajiya (integrators.glsl:351-393, 482-524, 625-665);
 * returns 0 for an unknown type */
static inline int o_specular_bounce(const OHit *h, const OFrame *f, uint32_t *rng, v3 *thr, v3 *pos_out, v3 *dir_out)
{
    if (h->type == 1) {
        *pos_out = vfma(f->normal, O_EPSILON, h->pos);
        *dir_out = vfma(f->normal, f->cos_in + f->cos_in, f->dir_in);
        *thr = vmul(*thr, h->base);
        return 1;
    }
    if (h->type == 2) {
        float k = O_FMA(-f->cos_in, f->cos_in, 1.0f);
        float c2 = O_FMA(-(f->eta * f->eta), k, 1.0f);
        float cos_out = 0.0f;
        int refl = (c2 <= 0.0f);
        if (!refl) {
            cos_out = sqrtf(o_max(0.0f, c2));
            refl = (o_rand(rng) < o_fresnel(f->cos_in, cos_out, f->eta));
        }
        if (refl) {
            *pos_out = vfma(f->normal, O_EPSILON, h->pos);
            *dir_out = vfma(f->normal, f->cos_in + f->cos_in, f->dir_in);
        } else {
            *pos_out = vfma(f->normal, -O_EPSILON, h->pos);
            *dir_out = vfma(f->normal, O_FMA(f->eta, f->cos_in, -cos_out), vscale(f->dir_in, f->eta));
        }
        *thr = vmul(*thr, h->base);
        return 1;
    }
    return 0;
}

/* distance_functions.glsl:27-60 (distance_triangle, after iquilezles.org "distfunctions").
 * [CHOICE] sign(x) = 1 / -1 / 0 (0 for NaN), clamp = minNum(maxNum(x,0),1), `v*k - w` fused per component */
static inline float o_sign(float x) { return (x > 0.0f) ? 1.0f : ((x < 0.0f) ? -1.0f : 0.0f); }
static inline float o_clamp01(float x) { return o_min(o_max(x, 0.0f), 1.0f); }
static inline float o_edge_dist2(v3 e, v3 q) /* dot2(e*clamp(dot(e,q)/dot2(e),0,1) - q) */
{
    float k = o_clamp01(vdot(e, q) / vdot(e, e));
    v3 w = V(O_FMA(e.x, k, -q.x), O_FMA(e.y, k, -q.y), O_FMA(e.z, k, -q.z));
    return vdot(w, w);
}
static float o_distance_triangle(v3 p, v3 a, v3 b, v3 c)
{
    v3 ba = vsub(b, a), pa = vsub(p, a);
    v3 cb = vsub(c, b), pb = vsub(p, b);
    v3 ac = vsub(a, c), pc = vsub(p, c);
    v3 nor = vcross(ba, ac);
    float s = (o_sign(vdot(vcross(ba, nor), pa)) + o_sign(vdot(vcross(cb, nor), pb))) + o_sign(vdot(vcross(ac, nor), pc));
    float m;
    if (s < 2.0f) {
        m = o_min(o_min(o_edge_dist2(ba, pa), o_edge_dist2(cb, pb)), o_edge_dist2(ac, pc));
    } else {
        float dn = vdot(nor, pa);
        m = (dn * dn) / vdot(nor, nor);
    }
    return sqrtf(m);
}
/* integrator_Hart (integrators.glsl:681-693) over intersect_scene_st (distance_functions.glsl:70-116):
 * sphere tracing with MARCH_ITER = 32, MARCH_EPS = 0.1 (compute_pass.comp:10-11); returns iter / (MARCH_ITER-1) */
static v3 o_hart(const OScene *sc, v3 org, v3 dir, float mint, float maxt)
{
    float t = mint;
    v3 p = vfma(dir, t, org);
    int i;
    for (i = 0; i < 32; ++i) {
        float best = O_INF; /* t_radius_idx.x; min_idx keeps lhs only if lhs.x < rhs.x (:64-67) */
        for (size_t j = 0; j < sc->n_tris; ++j) {
            const OTriangle *tr = &sc->tris[j];
            float dist = o_distance_triangle(p, V(tr->vert0[0], tr->vert0[1], tr->vert0[2]), V(tr->vert1[0], tr->vert1[1], tr->vert1[2]),
                                             V(tr->vert2[0], tr->vert2[1], tr->vert2[2]));
            best = (best < dist) ? best : dist;
        }
        float min_radius = o_min(O_INF, best); /* min(s_radius_idx.x, t_radius_idx.x), s_radius stays INF */
        if (min_radius < 0.1f || min_radius > maxt) break;
        t += min_radius;
        p = vfma(dir, min_radius, p);
    }
    float g = (float)i / 31.0f;
    return V(g, g, g);
}

/* integrators.glsl:24-543: every mode except Kajiya (9, above); any other index is integrator_Hart (:93-97 default) */
static v3 o_integrator(int mode, const OScene *sc, v3 org, v3 dir, int nbounce, uint32_t *rng, uint64_t *seg)
{
    const float mint = 0.0f, maxt = O_INF; /* compute_pass.comp:77-97 */
    switch (mode) {
    case 0: /* binary :24-38 */
        return o_splat(o_scene_any(sc, org, dir, mint, maxt, seg) ? 1.0f : 0.0f);
    case 1: { /* color :42-60 */
        OHit h = o_scene_hit(sc, org, dir, mint, maxt, seg);
        return h.hit ? h.base : V(0, 0, 0);
    }
    case 2: { /* depth :64-84 */
        OHit h = o_scene_hit(sc, org, dir, mint, maxt, seg);
        return o_splat(1.0f / (sqrtf(vdot(dir, dir)) * h.t));
    }
    case 3: { /* normal :88-105 — 0.5*normal + 0.5*isect */
        OHit h = o_scene_hit(sc, org, dir, mint, maxt, seg);
        float half_isect = 0.5f * (h.hit ? 1.0f : 0.0f);
        return V(O_FMA(0.5f, h.normal.x, half_isect), O_FMA(0.5f, h.normal.y, half_isect), O_FMA(0.5f, h.normal.z, half_isect));
    }
    case 4: { /* Utah :109-155 */
        OHit h = o_scene_hit(sc, org, dir, mint, maxt, seg);
        if (!h.hit) return o_sky(dir.y);
        v3 col = vadd(o_splat(0.1f), h.emissive);
        v3 n = (vdot(dir, h.normal) < 0.0f) ? h.normal : vneg(h.normal);
        float cos_light = o_max(0.0f, vdot(o_light_dir(), n));
        return vfma(h.base, cos_light, col);
    }
    case 5: { /* ambient occlusion :159-208 */
        OHit h = o_scene_hit(sc, org, dir, mint, maxt, seg);
        if (!h.hit) return V(0, 0, 0);
        v3 n = (vdot(dir, h.normal) < 0.0f) ? h.normal : vneg(h.normal);
        float acc = 0.0f;
        for (int i = 0; i < nbounce; ++i) {
            v3 o2 = vfma(n, O_EPSILON, h.pos);
            float u = o_rand(rng);
            float v = o_rand(rng);
            v3 d2 = vadd(n, o_map_uniform_sphere(u, v));
            acc += o_scene_any(sc, o2, d2, mint, maxt, seg) ? 1.0f : 0.0f;
        }
        return o_splat(1.0f - acc / (float)nbounce);
    }
    case 6: { /* Appel :212-263 */
        OHit h = o_scene_hit(sc, org, dir, mint, maxt, seg);
        if (!h.hit) return V(1, 1, 1);
        v3 dir_in = vnormalize(dir);
        v3 n = (vdot(dir_in, h.normal) > 0.0f) ? vneg(h.normal) : h.normal;
        v3 l = o_light_dir();
        if (o_scene_any(sc, vfma(n, O_EPSILON, h.pos), l, 0.0f, O_INF, seg)) return V(0, 0, 0);
        return o_splat(o_max(0.0f, vdot(l, n)));
    }
    case 7: { /* Whitted :267-403 */
        v3 col = o_splat(0.1f), thr = V(1, 1, 1);
        for (int i = 0; i < nbounce; ++i) {
            OHit h = o_scene_hit(sc, org, dir, mint, maxt, seg);
            if (!h.hit) return o_madd(thr, o_sky(dir.y), col);
            col = o_madd(thr, h.emissive, col);
            OFrame f = o_frame(dir, &h);
            if (h.type == 0) {
                v3 l = o_light_dir();
                if (o_scene_any(sc, vfma(f.normal, O_EPSILON, h.pos), l, 0.0f, O_INF, seg)) return col;
                float cos_light = o_max(0.0f, vdot(l, f.normal));
                return vfma(vmul(thr, h.base), cos_light, col);
            }
            v3 po, dn;
            if (!o_specular_bounce(&h, &f, rng, &thr, &po, &dn)) return V(0, 0, 0);
            org = po;
            dir = dn;
        }
        return V(0, 0, 0);
    }
    case 8: { /* Cook :407-543 */
        v3 col = V(0, 0, 0), thr = V(1, 1, 1);
        for (int i = 0; i < nbounce; ++i) {
            OHit h = o_scene_hit(sc, org, dir, mint, maxt, seg);
            if (!h.hit) return o_madd(thr, o_sky(dir.y), col);
            col = o_madd(thr, h.emissive, col);
            OFrame f = o_frame(dir, &h);
            if (h.type == 0) {
                v3 po = vfma(f.normal, O_EPSILON, h.pos);
                float u = o_rand(rng);
                float v = o_rand(rng);
                v3 dn = vadd(f.normal, o_map_uniform_sphere(u, v));
                thr = vmul(thr, vscale(vscale(h.base, O_INV_PI), O_PI));
                OHit h2 = o_scene_hit(sc, po, dn, mint, maxt, seg);
                if (!h2.hit) return o_madd(thr, o_sky(dn.y), col);
                return o_madd(thr, h2.emissive, col);
            }
            v3 po, dn;
            if (!o_specular_bounce(&h, &f, rng, &thr, &po, &dn)) return V(0, 0, 0);
            org = po;
            dir = dn;
        }
        return V(0, 0, 0);
    }
    default: /* eval_integrator's default branch (compute_pass.comp:96-97) */
        ++*seg; /* statistics: the march counts as one query */
        return o_hart(sc, org, dir, mint, maxt);
    }
}

/* compute_pass.comp:134-144 */
static inline int o_select_mode(const OSettings *s, float psx, float psy)
{
    int idx = s->top_left;
    if (psy > s->split_ratio[1]) {
        if (psx < s->split_ratio[0])
            idx = s->bottom_left;
        else
            idx = s->bottom_right;
    } else if (psx > s->split_ratio[0])
        idx = s->top_right;
    return idx;
}

/* ------------------------------------------------------------------------------------------ */
/* exported API                                                                                */

ORACLE_API int oracle_abi_version(void) { return 1; }

ORACLE_API void oracle_prepare(const OTriangle *tris, size_t n, OPrepTri *out)
{
    for (size_t i = 0; i < n; ++i) o_prepare(&tris[i], &out[i]);
}

/*
 * compute_pass.comp:121-167 over rows [y0,y1) of a W x H image.
 *   prev : row-major RGBA32F full image (W*H*4) or NULL; ignored when current_frame == 0
 *   out  : row-major RGBA32F full image; only rows [y0,y1) are written; alpha = 0 (:165-166)
 *   stats: optional, stats[0] += segments, stats[1] += samples
 * Build-defined: FP32 storage instead of rgba8 (see oracle_quantize_rgba8 for the compat path),
 * all rows rendered (the reference drops H % 16 rows, rvpt.cpp:1035-1036).
 * Every render mode (eval_integrator, compute_pass.comp:68-99: 0..9, anything else = integrator_Hart) and camera
 * mode (0 pinhole / 1 ortho / else spherical, :102-118).  Returns 0.
 */
/* The frame loop with the per-triangle terms already prepared (oracle_prepare; n_tris records of sizeof(OPrepTri) = 64
 * bytes): what a timing loop calls so that the one-off preparation is not part of every frame. */
ORACLE_API int oracle_render_prepared(const OSettings *st, const float cam[20], const OBvhNode *nodes,
                                      size_t n_nodes, const OTriangle *tris, const OPrepTri *prep, size_t n_tris,
                                      const OMaterial *mats, size_t n_mats, uint32_t W, uint32_t H,
                                      int traversal, const float *prev, float *out, uint32_t y0, uint32_t y1,
                                      uint64_t *stats)
{
    OScene sc = {nodes, n_nodes, tris, prep, n_tris, mats, n_mats, traversal};

    const float inv_w = 1.0f / (float)W, inv_h = 1.0f / (float)H; /* compute_pass.comp:51 */
    const uint32_t frame = st->current_frame;
    const float cf = (float)frame;
    const float inv_cf = 1.0f / (float)(frame + 1u); /* :54 */
    const float w = 1.0f / o_tan(0.5f * cam[17]);     /* camera.glsl:42 */
    const int aa = st->aa;
    int bad_mode = 0;
    uint64_t seg_total = 0, smp_total = 0;

    if (y1 > H) y1 = H;
    /* rows in chunks of 4: few enough scheduler round trips for 256 threads, fine enough for the sky/mesh imbalance */
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : seg_total, smp_total) reduction(| : bad_mode)
    for (uint32_t y = y0; y < y1; ++y) {
        for (uint32_t x = 0; x < W; ++x) {
            float *px = out + ((size_t)y * W + x) * 4;
            int mode = o_select_mode(st, (float)x * inv_w, (float)y * inv_h);
            uint32_t p_idx = x + y * W;                 /* util.glsl:35 */
            uint32_t rng = o_wang_hash(p_idx) + frame;  /* util.glsl:36 */
            v3 sampled = V(0, 0, 0);
            uint64_t seg = 0;
            for (int i = 0; i < aa; ++i) {
                float r0 = o_rand(&rng);
                float r1 = o_rand(&rng);
                float cx = ((float)x + r0) * inv_w; /* :153 */
                float cy = ((float)y + r1) * inv_h;
                cy = 1.0f - cy; /* :154 */
                v3 org, dir;
                o_camera_ray(st->camera_mode, cam, w, cx, cy, &org, &dir);
                v3 L = (mode == 9) ? o_kajiya(&sc, org, dir, 0.0f, O_INF, st->max_bounces, &rng, &seg)
                                   : o_integrator(mode, &sc, org, dir, st->max_bounces, &rng, &seg);
                sampled = vadd(sampled, L);
            }
            float faa = (float)aa;
            sampled = V(sampled.x / faa, sampled.y / faa, sampled.z / faa); /* :161 */
            v3 pv = V(0, 0, 0);
            if (frame != 0 && prev) { /* :146-148 — min(frame,1) gates the load */
                const float *pp = prev + ((size_t)y * W + x) * 4;
                pv = V(pp[0], pp[1], pp[2]);
            }
            /* :162-163 — [CHOICE] (prev*cf + sampled) fused, then * 1/(cf+1) */
            px[0] = O_FMA(pv.x, cf, sampled.x) * inv_cf;
            px[1] = O_FMA(pv.y, cf, sampled.y) * inv_cf;
            px[2] = O_FMA(pv.z, cf, sampled.z) * inv_cf;
            px[3] = 0.0f;
            seg_total += seg;
            smp_total += (uint64_t)aa;
        }
    }
    if (stats) {
        stats[0] += seg_total;
        stats[1] += smp_total;
    }
    return bad_mode ? -3 : 0;
}

ORACLE_API int oracle_render(const OSettings *st, const float cam[20], const OBvhNode *nodes,
                             size_t n_nodes, const OTriangle *tris, size_t n_tris,
                             const OMaterial *mats, size_t n_mats, uint32_t W, uint32_t H,
                             int traversal, const float *prev, float *out, uint32_t y0, uint32_t y1,
                             uint64_t *stats)
{
    OPrepTri *prep = (OPrepTri *)malloc(sizeof(OPrepTri) * (n_tris ? n_tris : 1));
    if (!prep) return -2;
    oracle_prepare(tris, n_tris, prep);
    int rc = oracle_render_prepared(st, cam, nodes, n_nodes, tris, prep, n_tris, mats, n_mats, W, H, traversal, prev, out, y0, y1, stats);
    free(prep);
    return rc;
}

/* rgba8 UNORM store/load (compute_pass.comp:41-42; Vulkan float->UNORM: clamp, scale, round to
 * nearest; NaN -> 0).  [CHOICE] ties round half up: floor(x*255 + 0.5). */
ORACLE_API void oracle_quantize_rgba8(const float *src, uint8_t *dst, size_t n_components)
{
    for (size_t i = 0; i < n_components; ++i) {
        float x = src[i];
        if (!(x > 0.0f)) x = 0.0f; /* also catches NaN */
        if (x > 1.0f) x = 1.0f;
        dst[i] = (uint8_t)floorf(fmaf(x, 255.0f, 0.5f));
    }
}
ORACLE_API void oracle_dequantize_rgba8(const uint8_t *src, float *dst, size_t n_components)
{
    for (size_t i = 0; i < n_components; ++i) dst[i] = (float)src[i] / 255.0f;
}

/* ---- known-answer-test entry points ---------------------------------------------------------- */
ORACLE_API uint32_t oracle_wang_hash(uint32_t seed) { return o_wang_hash(seed); }
ORACLE_API void oracle_rand_stream(uint32_t p_idx, uint32_t frame, size_t n, float *out,
                                   uint32_t *states)
{
    uint32_t s = o_wang_hash(p_idx) + frame;
    for (size_t i = 0; i < n; ++i) {
        out[i] = o_rand(&s);
        if (states) states[i] = s;
    }
}
/* compute_pass.comp:102-118: camera mode 0 pinhole, 1 orthographic, anything else spherical */
ORACLE_API void oracle_camera_ray(int mode, const float cam[20], float x, float y, float org[3], float dir[3])
{
    v3 o, d;
    o_camera_ray(mode, cam, 1.0f / o_tan(0.5f * cam[17]), x, y, &o, &d);
    org[0] = o.x; org[1] = o.y; org[2] = o.z;
    dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
}
ORACLE_API float oracle_distance_triangle(const float p[3], const float a[3], const float b[3], const float c[3])
{
    return o_distance_triangle(V(p[0], p[1], p[2]), V(a[0], a[1], a[2]), V(b[0], b[1], b[2]), V(c[0], c[1], c[2]));
}
ORACLE_API void oracle_sincos(float x, float *s, float *c) { o_sincos(x, s, c); }
ORACLE_API float oracle_tan(float x) { return o_tan(x); }
ORACLE_API void oracle_sphere_point(float u, float v, float out[3])
{
    v3 p = o_map_uniform_sphere(u, v);
    out[0] = p.x;
    out[1] = p.y;
    out[2] = p.z;
}
/* OpenMP threads used by the render calls (bench.py's cpu_baseline: the cores this process may actually run on).  Returns
 * the previous setting; 0 if built without OpenMP. */
ORACLE_API int oracle_set_threads(int n)
{
#ifdef _OPENMP
    int prev = omp_get_max_threads();
    if (n > 0) omp_set_num_threads(n);
    return prev;
#else
    (void)n;
    return 0;
#endif
}
ORACLE_API float oracle_div_dots(float a, float b) { return o_div_dots(a, b); }
ORACLE_API float oracle_fresnel(float cos_in, float cos_out, float eta) { return o_fresnel(cos_in, cos_out, eta); }
ORACLE_API int oracle_tri_test(const float org[3], const float dir[3], const OTriangle *tri, float mint,
                               float maxt, float tuv[3])
{
    OPrepTri p;
    o_prepare(tri, &p);
    return o_tri_test(V(org[0], org[1], org[2]), V(dir[0], dir[1], dir[2]), &p, mint, maxt, &tuv[0],
                      &tuv[1], &tuv[2]);
}
ORACLE_API int oracle_aabb_test(const float org[3], const float dir[3], const float bmin[3],
                                const float bmax[3], float mint, float maxt)
{
    v3 d = V(dir[0], dir[1], dir[2]);
    v3 inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    return o_aabb_test(V(org[0], org[1], org[2]), inv, V(bmin[0], bmin[1], bmin[2]),
                       V(bmax[0], bmax[1], bmax[2]), mint, maxt);
}
ORACLE_API void oracle_pinhole_ray(const float cam[20], float x, float y, float org[3], float dir[3])
{
    v3 o, d;
    o_pinhole_ray(cam, 1.0f / o_tan(0.5f * cam[17]), x, y, &o, &d);
    org[0] = o.x; org[1] = o.y; org[2] = o.z;
    dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
}
/* closest hit of one ray: returns triangle index or -1 */
ORACLE_API long oracle_closest_hit(const OBvhNode *nodes, size_t n_nodes, const OTriangle *tris,
                                   size_t n_tris, int traversal, const float org[3], const float dir[3],
                                   float *t_hit)
{
    OPrepTri *prep = (OPrepTri *)malloc(sizeof(OPrepTri) * (n_tris ? n_tris : 1));
    oracle_prepare(tris, n_tris, prep);
    OScene sc = {nodes, n_nodes, tris, prep, n_tris, NULL, 0, traversal};
    long h = o_closest_hit(&sc, V(org[0], org[1], org[2]), V(dir[0], dir[1], dir[2]), 0.0f, O_INF, t_hit);
    free(prep);
    return h;
}

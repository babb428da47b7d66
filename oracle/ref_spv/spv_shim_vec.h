/*
 * spv_shim_vec.h — vector-typed part of the shim (included by the generated file after its typedefs of vec2f, vec3f,
 * vec4f, vec2i, mat2_vec2f, mat4_vec4f).  See spv_shim.h for the complete list and the rules.
 *
 *   OpDot:               ((a.x*b.x + a.y*b.y) + a.z*b.z), components in order
 *   OpMatrixTimesVector: (((c0*v.x + c1*v.y) + c2*v.z) + c3*v.w), columns in order
 *   Cross:               (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y)
 *   Length:              sqrt(dot(v,v));   Normalize: v * (1 / sqrt(dot(v,v)))
 * With -DREF_SPV_FUSED each "+ p*q" above becomes one fused multiply-add in the same order (the contraction a Vulkan
 * compiler is allowed to apply; the product's arithmetic).
 */
#ifndef SPV_SHIM_VEC_H
#define SPV_SHIM_VEC_H

#ifdef REF_SPV_FUSED
#define SHIM_MAD(a, b, c) fmaf((a), (b), (c))
#else
#define SHIM_MAD(a, b, c) ((c) + (a) * (b))
#endif

/* every group below exists only if the module declares the type it works on (tools/spv2c.py defines SHIM_HAS_<type>) */
#ifdef SHIM_HAS_vec3f
static inline float shim_dot3(vec3f a, vec3f b)
{
#ifdef REF_SPV_LIBM /* another admissible driver: the products summed from the last component down, no fma */
    return (a.v[2] * b.v[2] + a.v[1] * b.v[1]) + a.v[0] * b.v[0];
#endif
    float r = a.v[0] * b.v[0];
    r = SHIM_MAD(a.v[1], b.v[1], r);
    r = SHIM_MAD(a.v[2], b.v[2], r);
    return r;
}
static inline vec3f shim_cross3f(vec3f a, vec3f b)
{
    vec3f r;
    r.v[0] = a.v[1] * b.v[2] - b.v[1] * a.v[2];
    r.v[1] = a.v[2] * b.v[0] - b.v[2] * a.v[0];
    r.v[2] = a.v[0] * b.v[1] - b.v[0] * a.v[1];
    return r;
}
static inline float shim_length3f(vec3f a) { return sqrtf(shim_dot3(a, a)); }
static inline vec3f shim_normalize3f(vec3f a)
{
#ifdef REF_SPV_LIBM /* another admissible driver: v / length(v), a division per component */
    const float len = sqrtf(shim_dot3(a, a));
    vec3f q;
    q.v[0] = a.v[0] / len;
    q.v[1] = a.v[1] / len;
    q.v[2] = a.v[2] / len;
    return q;
#endif
    const float inv = 1.0f / sqrtf(shim_dot3(a, a));
    vec3f r;
    r.v[0] = a.v[0] * inv;
    r.v[1] = a.v[1] * inv;
    r.v[2] = a.v[2] * inv;
    return r;
}
#endif
#ifdef SHIM_HAS_vec2f
static inline float shim_dot2(vec2f a, vec2f b)
{
    float r = a.v[0] * b.v[0];
    r = SHIM_MAD(a.v[1], b.v[1], r);
    return r;
}
#endif
#ifdef SHIM_HAS_mat2_vec2f
static inline vec2f shim_mat2x2_times_vec(mat2_vec2f m, vec2f x)
{
    vec2f r;
    for (int k = 0; k < 2; ++k) {
        float s = m.c[0].v[k] * x.v[0];
        s = SHIM_MAD(m.c[1].v[k], x.v[1], s);
        r.v[k] = s;
    }
    return r;
}
#endif
#ifdef SHIM_HAS_mat4_vec4f
static inline vec4f shim_mat4x4_times_vec(mat4_vec4f m, vec4f x)
{
    vec4f r;
    for (int k = 0; k < 4; ++k) {
        float s = m.c[0].v[k] * x.v[0];
        s = SHIM_MAD(m.c[1].v[k], x.v[1], s);
        s = SHIM_MAD(m.c[2].v[k], x.v[2], s);
        s = SHIM_MAD(m.c[3].v[k], x.v[3], s);
        r.v[k] = s;
    }
    return r;
}
#endif

#if defined(SHIM_HAS_image) && defined(SHIM_HAS_vec2i) && defined(SHIM_HAS_vec4f)
static inline vec2i shim_image_size(shim_image* im)
{
    vec2i r;
    r.v[0] = im->width;
    r.v[1] = im->height;
    return r;
}
static inline vec4f shim_image_read(shim_image* im, vec2i p)
{
    vec4f r;
    const float* t = im->texels + 4 * ((size_t)p.v[1] * (size_t)im->width + (size_t)p.v[0]);
    for (int k = 0; k < 4; ++k) r.v[k] = t[k];
    return r;
}
static inline void shim_image_write(shim_image* im, vec2i p, vec4f x)
{
    float* t = im->texels + 4 * ((size_t)p.v[1] * (size_t)im->width + (size_t)p.v[0]);
    for (int k = 0; k < 4; ++k) t[k] = im->unorm8 ? shim_unorm8(x.v[k]) : x.v[k];
}
#endif
#endif

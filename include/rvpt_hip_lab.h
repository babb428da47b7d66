/*
 * rvpt_hip_lab.h — what the LABORATORY build of the library (rvpt_amd/librvpt_hip_debug.so: the sources of librvpt_hip.so compiled with -DRVPT_HIP_LAB=1
 * -DRV_REPORT_STACK_OVERFLOW=1, rvpt_amd/build.py) exports beside the C ABI of rvpt_hip.h.  None of it has a counterpart in the reference; a caller of
 * `class RVPT` needs none of it; the parity tests, tools/fuzz_culls.py and the experiments do:
 *   - diagnostics of the arithmetic specification and of the packet kernel's exact culls (rvpt_hip_selftest_*),
 *   - the host-side forms of data the kernels consume, GPU-free (rvpt_camera_rects, rvpt_bounce_rows, rvpt_claim_order, rvpt_bvh_wide_form, rvpt_bvh_quant_form),
 *   - the kernels that were built, are bit-exact and measured SLOWER (profiles/EXPERIMENTS.md): the 8-wide walk (RVPT_HIP_BVH_WIDE8=1) and the walk over
 *     64-byte quantised nodes (RVPT_HIP_BVH_QUANT=1),
 *   - the tuning knobs the sweeps of rounds 1-5 found flat (RVPT_HIP_BVH_WIDE, _WIDE_RESIDENT, _NO_RESIDENT, _NO_PACKED_HEADS, _CALLER_LAYOUT, _TOP_NODES,
 *     _STACK_LDS, _REFILL, _LEAF_BATCH, _CAM_MIN, _DETACH, _FORCE_STACK_LEVELS, RVPT_HIP_BRUTE_PACKETS, RVPT_HIP_PACKETS_LEAN_INSTANCE, _BLOCKS_PER_CU, _FIRST_UNITS, _CLAIM_UNITS,
 *     RVPT_HIP_TIMELINE): the release library reads none of them,
 *   - the kernels' internal checks (RVPT_HIP_DEBUG=1: a traversal-stack overflow is reported by rvpt_hip_wait).
 * rvpt_hip_build_flags() tells the two builds apart.  The release library reads: RVPT_HIP_QUIET, RVPT_HIP_DEBUG (refused without the checks),
 * RVPT_HIP_FRAMES_IN_FLIGHT, RVPT_HIP_NO_OVERLAP, RVPT_HIP_PACKETS_CULL, RVPT_HIP_PACKETS_BOUNCE_CULL, RVPT_HIP_PACKETS_BOX_CULL (A/B of the exact culls on the shipped kernels), RVPT_HIP_PACKETS_INTERLEAVE (A/B of the packet kernel's claim order),
 * RVPT_HIP_COMM_TIMEOUT_S, GPU_MAX_HW_QUEUES (to print its note), RVPT_BVH_THREADS / RVPT_BVH_TRAVERSAL_COST (the builder).
 */
#ifndef RVPT_HIP_LAB_H
#define RVPT_HIP_LAB_H

#include "rvpt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostics of the arithmetic specification (no reference counterpart; used by the parity tests).
 * selftest_div: out[i] = the kernels' ray/plane quotient (Markstein's sequence on v_rcp_f32) of host arrays
 * a[i], b[i], evaluated on `device_id`.  selftest_rcp: for every binary32 b of every exponent 1..254 compares the refined
 * hardware reciprocal with the correctly rounded 1/b; mismatches_per_exponent[e] (256 entries) = number of mantissas
 * that differ (expected: 0 for e <= 252 and for 2^126 itself, i.e. [253] == 2^23 - 1, [254] == 2^23: 1/b subnormal). */
int rvpt_hip_selftest_div(int device_id, const float *a, const float *b, float *out, size_t n);
int rvpt_hip_selftest_rcp(int device_id, uint64_t mismatches_per_exponent[256]);
/* selftest_pretest (ABI 5): the division-free pre-test of a camera round (rvpt_early_out.h: `!(a > closest * den)` on the camera record of a plane
 * at distance a[i] >= 0) against the quotient it stands in for: out[i] bit 0 = the pre-test lets the pair through, bit 1 = the quotient satisfies
 * 0 < t < closest.  The pre-test must never stop a pair the quotient accepts: bit 1 implies bit 0 (tests/test_gpu_parity.py).  selftest_rcp also
 * counts, per exponent, the b for which v_rcp_f32(-b) != -v_rcp_f32(b) (expected: none). */
int rvpt_hip_selftest_pretest(int device_id, const float *a, const float *den, const float *closest, unsigned char *out, size_t n);

/* The screen rectangles of the packet kernel's camera rounds (ABI 6; rvpt_amd/csrc/rvpt_rect.h has the construction and the error bound).  For the
 * camera of a launch, every triangle gets the conservative rectangle of 16 x 4 pixel blocks outside which no camera ray (camera.glsl:29-51 through
 * compute_pass.comp:151-156) can be accepted by the triangle test (intersection.glsl:267-323); a camera round — 64 rays of ONE block — skips the
 * triangles whose rectangle does not hold its block.  A superset test: images do not change (RVPT_HIP_PACKETS_CULL=0 switches it off).
 * rvpt_camera_rects: the same function on the host, no GPU needed — `prepared` = n_tris x 16 floats (v0, n, e0, e1, Gram terms: what
 * rvpt_hip_selftest_camera_rects returns in prepared_out), rects_out = n_tris x 2 words: x0 | x1 << 16 (units of 16 pixels), y0 | y1 << 16 (units of
 * 4 rows); x0 > x1 = no block.
 * rvpt_hip_selftest_camera_rects: on the context's scene, camera and image size (after upload_scene + set_frame), every pixel x n_samples jittered camera
 * rays (the samples of frames current_frame .. + n_samples - 1) x every triangle through the kernels' float test with an open interval:
 * out[0] = accepted pairs, out[1] = accepted pairs whose block lies outside the triangle's rectangle (the claim: 0), out[2] = (block, triangle) pairs whose
 * rectangle holds the block, out[3] = all (block, triangle) pairs.  prepared_out (n_tris x 16 floats) / rects_out (n_tris x 2 words): optional copies of the
 * device's prepared records and rectangles. */
/* The bounce cull of the packet kernel (ABI 6; rvpt_amd/csrc/rvpt_packets.hip: bounce_visibility): a segment that leaves triangle A on side s — Lambert, mirror and
 * reflecting dielectric leave on the side the ray came from, a refracting one on the other (integrators.glsl:600-668) — cannot hit a triangle that lies wholly
 * behind A's plane as seen from s; a table made once per scene says which can, and a bounce round walks the union of its 64 rays' rows.  A superset test
 * (RVPT_HIP_PACKETS_BOUNCE_CULL=0 switches it off).  selftest_bounce_cull: on the context's scene and camera, every pixel x n_samples full paths traced against
 * EVERY triangle: out[0] = (segment, triangle) pairs the float test accepts with its interval wide open on segments that leave a triangle, out[1] = those whose
 * triangle the table excludes (the claim: 0), out[2] = bits set in the table, out[3] = its size in bits (2 n^2); (ABI 8) the leaf boxes of the same rounds
 * (rvpt_amd/csrc/rvpt_vis.h; RVPT_HIP_PACKETS_BOX_CULL=0 switches them off): out[4] = accepted pairs whose ray fails the slab test of the triangle's leaf box (the
 * claim: 0), out[5] / out[6] = (segment, leaf box) pairs tested / passed, out[7] = 0. */
int rvpt_hip_selftest_bounce_cull(rvpt_hip_ctx *ctx, uint32_t n_samples, uint64_t out[8]);
/* selftest_fast_div (ABI 6, host only, no GPU): q[i] = x[i] / divisor through the multiply-high form the frame kernels use to turn a claimed work index into
 * (frame, tile, pixel) (rvpt_kernels.h: FastDiv) — must equal the integer quotient for every x and every divisor >= 1. */
int rvpt_hip_selftest_fast_div(uint32_t divisor, const uint32_t *x, uint32_t *q, size_t n);
int rvpt_camera_rects(const float *prepared, size_t n_tris, const rvpt_camera_data *cam, uint32_t width, uint32_t height, uint32_t *rects_out);
int rvpt_hip_selftest_camera_rects(rvpt_hip_ctx *ctx, uint32_t n_samples, uint64_t out[4], float *prepared_out, uint32_t *rects_out);

/* rvpt_bounce_rows (host only, no GPU): the bounce cull's table as upload_scene builds it (rvpt_amd/csrc/rvpt_vis.h) for n_tris <= 1024 triangles — `tris` the
 * reference Triangle records (for the scene scale: largest |coordinate| + largest extent), `prepared` their prepared records (n_tris x 16 floats); rows_out =
 * 2 n_tris rows of ceil(n_tris / 32) words, row 2 A + s bit B = 0 only when B lies wholly behind A's plane seen from side s by more than 2^-10 scene scales and
 * both are well shaped; *scale_out = the scene scale, 0 = no table for this scene (nothing written). */
int rvpt_bounce_rows(const float *tris, const float *prepared, size_t n_tris, uint32_t *rows_out, double *scale_out);
/* rvpt_bounce_leaf_boxes (host only, no GPU): the leaf boxes of the bounce rounds as upload_scene builds them (rvpt_amd/csrc/rvpt_vis.h) — boxes_out = 8 floats per
 * group of *leaf_tris_out (8) consecutive triangles: lo.xyz, hi.xyz, 0, 0, widened by 2^-9 (scene scale + 2 EPSILON); a group with a badly shaped or non-finite
 * triangle gets (-inf, +inf); tri_boxes_out (may be NULL): 8 floats per TRIANGLE, its own box (the second level of the same cull).  Nothing is written for a scene
 * without a scale. */
int rvpt_bounce_leaf_boxes(const float *tris, size_t n_tris, float *boxes_out, uint32_t *leaf_tris_out, float *tri_boxes_out);
/* rvpt_claim_order (host only, no GPU): the order in which the packet kernel's launches of fewer than four frames deal a frame's 16 x 4 pixel blocks (round 6;
 * rvpt_kernels.h: claim_order_block, rvpt_abi.hip: plan_claim_order) for a rank that owns n_work_frame work items (owned tiles x 256) and groups of group_blocks
 * (1, 2, 4, 8) consecutive blocks: order_out[b] (n_work_frame / 64 entries, may be NULL) = the tile-linear block the b-th block of the claim order is; params_out
 * (may be NULL) = {groups, stride, log2 group_blocks}, groups == 0 when this size keeps the tile-linear order (the identity is written).  Always a bijection. */
int rvpt_claim_order(uint32_t n_work_frame, uint32_t group_blocks, uint32_t *order_out, uint32_t params_out[3]);

/* The 4-wide regrouping of a binary tree in the reference node layout that BVH contexts walk by default (rvpt_bvh4.hip) — what
 * rvpt_hip_upload_scene builds internally, exported so that a host (or a test) can look at it.  An inner node's child list [left, right] has inner
 * children replaced, in place, by their two children (largest box first) until it holds four — ONLY across boxes that contain their children's boxes,
 * which keeps the reference's traversal (intersection.glsl:361-413: a node is visited iff its own box passes when the depth-first, left-first order
 * reaches it) bit for bit.  wide_out: 32 floats per wide node — minx[4] maxx[4] miny[4] maxy[4] minz[4] maxz[4] head[4] pad[4], breadth first;
 * head = first | count << head_shift for a leaf child (count > 0), the wide index of an inner child (count 0), 0xFFFFFFFF for an unused slot.
 * *n_wide_out = 0 when the tree has no wide form (the root is a leaf, head_shift == 0: leaf sizes do not pack beside the indices).  stack_need_out:
 * the most slots a depth-first walk of the wide tree holds at once.  RVPT_HIP_ERR_SIZE if wide_capacity (in nodes) is too small.  No GPU needed. */
int rvpt_bvh_wide_form(const rvpt_bvh_node *nodes, size_t n_nodes, uint32_t head_shift, float *wide_out, size_t wide_capacity, size_t *n_wide_out,
                       uint32_t *stack_need_out);

/* The 64-byte QUANTISED form of those wide nodes (rvpt_bvh4.hip: trace_bvh4q, opt-in with RVPT_HIP_BVH_QUANT=1; profiles/EXPERIMENTS.md 5.16) and the exact
 * leaf boxes that go with it.  Under containment inner boxes only cull (intersection.glsl:361-413 visits a node iff its OWN box passes), so a child box may
 * be any superset as long as a leaf's exact box is tested at its visit.  quant_out: 16 words per wide node — origin x y z (float), scale x y z (float, a
 * power of two), qminx qmaxx qminy qmaxy qminz qmaxz (byte k = child k; [origin + qmin scale, origin + qmax scale] contains the child's exact box), the
 * four heads of the 128-byte form.  leaf_boxes_out (may be NULL): 8 floats per TRIANGLE index, at [8 first] the box of the leaf that starts at `first`
 * (minx maxx miny maxy minz maxz 0 0).  *extent_out: the largest |coordinate| of the tree (the margin of the kernel's conservative test).
 * *n_quant_out = 0 when the tree has no quantised form (no wide form; an inner node that does not contain a child; two leaves starting at one triangle;
 * a non-finite bound): the exact nodes serve it.  No GPU needed. */
int rvpt_bvh_quant_form(const rvpt_bvh_node *nodes, size_t n_nodes, uint32_t head_shift, size_t n_tris, uint32_t *quant_out, size_t quant_capacity,
                        size_t *n_quant_out, float *leaf_boxes_out, float *extent_out);

#ifdef __cplusplus
}
#endif
#endif /* RVPT_HIP_LAB_H */

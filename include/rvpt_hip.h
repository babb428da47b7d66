/*
 * rvpt_hip.h — C ABI of the MI355X (gfx950) path-trace backend that replaces RVPT's
 * VkComputePipeline dispatch of assets/shaders/compute_pass.comp.
 *
 * The reference has no plugin/FFI layer; the seam sits inside `class RVPT`
 * (src/rvpt/rvpt.{h,cpp}).  Every entry point below names the reference code it
 * replaces (paths relative to the reference tree).  Plain pointers and sizes only;
 * no exceptions, asserts or C++/torch types cross this boundary.
 *
 * All functions return 0 on success and a negative RVPT_HIP_ERR_* code on failure;
 * rvpt_hip_last_error() gives the text (mirrors VK_CHECK_RESULT + fmt::print,
 * src/rvpt/vk_util.h:18-27).
 *
 * One context == one GPU == one process rank.  A context is not thread-safe (the
 * reference is single-threaded; Queue::submit_mutex, src/rvpt/vk_util.h:160, is
 * never contended).  The caller owns every host array; the library owns all device
 * memory and keeps no host pointer after a call returns.
 *
 * PROCESS ENVIRONMENT: the library never modifies it.  A context creates seven HIP streams of its own (six
 * for launches in flight — three rotate for most launches, six for short BVH launches — plus one for
 * uploads, the temporal blend and read-back); ROCm maps streams onto GPU_MAX_HW_QUEUES hardware queues
 * (default 4), so a host that wants the published throughput sets GPU_MAX_HW_QUEUES=8 (or more) BEFORE the
 * first HIP call of the process (measured: -10 % when two frame streams share a queue).  rvpt_hip_create
 * writes one line to stderr, once per process, when it finds the variable unset or below 8
 * (RVPT_HIP_QUIET=1 silences it).  INTEGRATION.md.
 */
#ifndef RVPT_HIP_H
#define RVPT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RVPT_HIP_ABI_VERSION 8 /* 2: + rvpt_hip_dispatch_frames, RVPT_HIP_TRAVERSAL_BVH_ORDERED; 3: + the RCCL communicator (comm_*, gather, collective read), selftest_*;
                                  4: + rvpt_hip_comm_barrier, bounded collectives (RVPT_HIP_COMM_TIMEOUT_S);
                                  5: the wavefront pipelines of ABI 3-4 are retired (flags 0x40 / 0x80 / 0x100 are rejected), + RVPT_HIP_BVH_PER_LANE;
                                     unknown flag bits are an error;
                                  6: + rvpt_camera_rects, rvpt_hip_selftest_camera_rects (the screen rectangles of the packet kernel's camera rounds), rvpt_hip_selftest_bounce_cull;
                                  7: + rvpt_bvh_quant_form (the 64-byte quantised wide nodes, RVPT_HIP_BVH_QUANT=1);
                                  8: the release library exports the 31 entry points of THIS header only — what a caller of `class RVPT` needs; the
                                     selftests, the host-side forms of the device data (rvpt_camera_rects, rvpt_bounce_rows, rvpt_bvh_wide_form,
                                     rvpt_bvh_quant_form), the opt-in walks that measured slower (8-wide, quantised) and the tuning knobs live in the
                                     laboratory build librvpt_hip_debug.so (include/rvpt_hip_lab.h); + rvpt_hip_build_flags, rvpt_hip_get_cull_info,
                                     rvpt_hip_comm_info */

/* ---- POD layouts: byte-identical to the reference's GPU buffers ------------------ */

/* src/rvpt/geometry.h:76-111 == assets/shaders/structs.glsl:1-7 (64 B).
 * vert{0,1,2}[3] hold the host-side face normal (unused by the live shader code);
 * mat_id[0] is the material index stored as a float. */
typedef struct rvpt_triangle {
    float vert0[4];
    float vert1[4];
    float vert2[4];
    float mat_id[4];
} rvpt_triangle;

/* src/rvpt/bvh.h:12-19 == structs.glsl:9-14 (32 B).  Root is node 0, the two children of
 * an inner node are `first` and `first+1`, a node is a leaf iff primitive_count > 0.
 * bounds = {minx,maxx,miny,maxy,minz,maxz} (assets/shaders/intersection.glsl:376-377). */
typedef struct rvpt_bvh_node {
    uint32_t first_child_or_primitive;
    uint32_t primitive_count;
    float bounds[6];
} rvpt_bvh_node;

/* src/rvpt/material.h:9-26 == structs.glsl:22-33 (48 B).  type = (int)data[0]
 * (0 Lambert, 1 mirror, 2 dielectric); ior is read from albedo[3]
 * (assets/shaders/intersection.glsl:45-57). */
typedef struct rvpt_material {
    float albedo[4];
    float emission[4];
    float data[4];
} rvpt_material;

/* src/rvpt/rvpt.h:77-89 == compute_pass.comp:28-40 (std140, 40 B). */
typedef struct rvpt_render_settings {
    int32_t max_bounces;
    int32_t aa;
    uint32_t current_frame;
    int32_t camera_mode;
    int32_t top_left_render_mode;
    int32_t top_right_render_mode;
    int32_t bottom_left_render_mode;
    int32_t bottom_right_render_mode;
    float split_ratio[2];
} rvpt_render_settings;

/* Camera::get_data(), src/rvpt/camera.cpp:55-66 == compute_pass.comp:44-49 (80 B):
 * column-major camera-to-world mat4, then params = (aspect, vfov_rad, ortho_scale, 0). */
typedef struct rvpt_camera_data {
    float matrix[16];
    float params[4];
} rvpt_camera_data;

/* ---- error codes ------------------------------------------------------------------ */
#define RVPT_HIP_OK 0
#define RVPT_HIP_ERR_INVALID (-1)     /* bad argument / call order                       */
#define RVPT_HIP_ERR_HIP (-2)         /* a HIP runtime call failed                       */
#define RVPT_HIP_ERR_UNSUPPORTED (-3) /* reserved: every render / camera mode of compute_pass.comp is implemented */
#define RVPT_HIP_ERR_NO_DEVICE (-4)   /* no gfx950 device visible                        */
#define RVPT_HIP_ERR_SIZE (-5)        /* destination buffer too small                    */
#define RVPT_HIP_ERR_COMM (-6)        /* RCCL missing or a collective call failed        */

/* ---- create flags ------------------------------------------------------------------- */
#define RVPT_HIP_TRAVERSAL_BRUTE 0x0u /* LDS-staged brute-force closest hit (north star)  */
#define RVPT_HIP_TRAVERSAL_BVH 0x1u   /* intersect_bvh semantics (intersection.glsl:361)  */
#define RVPT_HIP_TRAVERSAL_BVH_ORDERED 0x2u /* same BVH, nearer child first — the reference's own
                                         "TODO: Order the children on the stack" (intersection.glsl:405); same image
                                         except where exact ties / slab rounding decide, far fewer nodes visited */
#define RVPT_HIP_TRAVERSAL_MASK 0x3u
#define RVPT_HIP_COUNT_SEGMENTS 0x4u  /* count path segments (for the roofline model)     */
#define RVPT_HIP_KERNEL_SIMPLE 0x8u   /* one-pixel-per-lane kernel, no ray regeneration   */
#define RVPT_HIP_TIMING 0x10u         /* bracket every frame kernel with hipEvents        */
#define RVPT_HIP_ACCUM_UNORM8 0x20u   /* reference-format accumulation: the running mean is clamped to [0,1]
                                         and rounded to 8 bits after every frame, as storing to the rgba8
                                         temporal image does (compute_pass.comp:41-42,165); default is FP32 */
#define RVPT_HIP_BRUTE_MIXED_PACKETS 0x200u /* brute-force contexts: round 2's frame kernel (a lane takes its next pixel the moment its pixel
                                          is finished: packets mix camera and bounce rays) instead of the packet kernel that is the default
                                          for LDS-resident scenes in the lean configuration (rvpt_packets.hip: full packets of one kind per
                                          round, camera rays with the packet-uniform early-out).  Same image either way */
#define RVPT_HIP_BVH_PER_LANE 0x400u  /* BVH contexts: rounds 1-3's kernels — binary nodes, every segment walks the tree per lane — instead of the walk over the
                                         4-wide regrouping of the tree (rvpt_bvh4.hip; with camera packets where the scene is LDS-resident) that is the default
                                         for the reference's child order.  Same image */
#define RVPT_HIP_FLAGS_KNOWN 0x63Fu   /* every bit above; rvpt_hip_create rejects anything else (0x40, 0x80, 0x100: the wavefront
                                         pipelines of ABI 3-4, measured at 0.55x / 0.7x of the persistent kernels and retired) */

/* ---- read formats --------------------------------------------------------------------- */
#define RVPT_HIP_FORMAT_RGBA32F 0     /* float radiance running mean, alpha 0            */
#define RVPT_HIP_FORMAT_RGBA8_UNORM 1 /* clamp+quantise of the above (reference image format,
                                         compute_pass.comp:41-42)                        */

/* Image tiles are RVPT_HIP_TILE x RVPT_HIP_TILE pixels — the footprint of one reference
 * work-group (compute_pass.comp:27).  The tile at (tx, ty) of the tiles_x-wide tile grid has slot
 * s = ty * tiles_x + (tx + RVPT_HIP_TILE_SHIFT * ty) % tiles_x (row-major, every row rotated by
 * RVPT_HIP_TILE_SHIFT more tiles than the one above: a diagonal pattern even where tiles_x is a
 * multiple of tile_world) and is owned by rank s % tile_world as that rank's local tile s / tile_world
 * (ABI 5; ABI <= 4: s = ty * tiles_x + tx, which gave each of 8 ranks whole tile columns of a 1920-wide image). */
#define RVPT_HIP_TILE 16
#define RVPT_HIP_TILE_SHIFT 3

typedef struct rvpt_hip_ctx rvpt_hip_ctx;

int rvpt_hip_abi_version(void);
/* (ABI 8) What this build of the library carries: RVPT_HIP_BUILD_LAB = the laboratory build (include/rvpt_hip_lab.h), RVPT_HIP_BUILD_DEBUG_CHECKS = the
 * kernels' internal checks (a BVH traversal that pushes past the stack the host sized is reported by rvpt_hip_wait under RVPT_HIP_DEBUG=1; without them
 * rvpt_hip_create refuses RVPT_HIP_DEBUG=1 instead of ignoring it). */
#define RVPT_HIP_BUILD_LAB 0x1u
#define RVPT_HIP_BUILD_DEBUG_CHECKS 0x2u
uint32_t rvpt_hip_build_flags(void);

/* Number of usable devices (replaces vk-bootstrap device selection, rvpt.cpp:477-570). */
int rvpt_hip_device_count(int *count);

/* Replaces create_rendering_resources()/add_per_frame_data(): pipeline (rvpt.cpp:676-681),
 * temporal image (rvpt.cpp:759-766), per-frame buffers + output image (rvpt.cpp:798-866).
 * Allocates the scene buffers and this rank's RGBA32F accumulator tiles on `device_id`.
 * tile_rank/tile_world select the image partition (1 GPU: 0/1). */
int rvpt_hip_create(rvpt_hip_ctx **out, int device_id, uint32_t width, uint32_t height,
                    uint32_t tile_rank, uint32_t tile_world, uint32_t flags);
void rvpt_hip_destroy(rvpt_hip_ctx *ctx);

/* Replaces the three scene memcpys the reference repeats every frame (rvpt.cpp:124-126).
 * Call when the scene changes.  `nodes` may be NULL for brute-force contexts.  Triangles must
 * already be in BVH-leaf order (Bvh::permute_primitives, bvh.h:72-79) for BVH contexts. */
int rvpt_hip_upload_scene(rvpt_hip_ctx *ctx, const rvpt_bvh_node *nodes, size_t n_nodes,
                          const rvpt_triangle *tris, size_t n_tris, const rvpt_material *mats,
                          size_t n_mats);

/* Replaces the settings + camera uniform copies (rvpt.cpp:118,120).  The accumulate/reset
 * rule (rvpt.cpp:21-29,102-111) stays with the caller: settings->current_frame is
 * authoritative, 0 means "ignore the accumulator". */
int rvpt_hip_set_frame(rvpt_hip_ctx *ctx, const rvpt_render_settings *settings,
                       const rvpt_camera_data *camera);

/* Replaces record_compute_command_buffer() + queue submit (rvpt.cpp:1005-1039,352-354):
 * asynchronous enqueue of one frame.  Up to `frames_in_flight` frame kernels overlap on the device (they
 * write per-frame sample buffers); the temporal blend into the accumulator runs in dispatch order. */
int rvpt_hip_dispatch(rvpt_hip_ctx *ctx);

/* The reference's steady state — camera and settings unchanged, update() only increments current_frame
 * (rvpt.cpp:102-111) — as one call: enqueues the n_frames consecutive frames settings->current_frame ...
 * current_frame + n_frames - 1 of the last set_frame().  The accumulator ends up bit-identical to n_frames calls of
 * set_frame(current_frame + k) / dispatch(); the frames share one kernel launch (work items = frames x pixels, one
 * temporal-blend pass applying the frames in order), which removes the per-frame ramp-up and drain from the
 * device time.  The caller advances its frame counter by n_frames.  1 <= n_frames <= RVPT_HIP_MAX_FRAMES_PER_DISPATCH. */
#define RVPT_HIP_MAX_FRAMES_PER_DISPATCH 64u
int rvpt_hip_dispatch_frames(rvpt_hip_ctx *ctx, uint32_t n_frames);

/* Replaces raytrace_work_fence.wait()/reset() (rvpt.cpp:115-116).  query: 0 done, 1 pending. */
int rvpt_hip_wait(rvpt_hip_ctx *ctx);
int rvpt_hip_query(rvpt_hip_ctx *ctx);
/* Fence::wait with its timeout (vk_util.cpp:65,94-97: DEFAULT_FENCE_TIMEOUT = 1 s, result ignored upstream): waits at
 * most timeout_ns for everything dispatched so far.  0 done, 1 still pending when the time was up, negative error. */
int rvpt_hip_wait_for(rvpt_hip_ctx *ctx, uint64_t timeout_ns);

/* Host read-back of the frame (the reference only samples output_image in its blit,
 * rvpt.cpp:851-852,960-964).  Row-major, top row first, width*height*4 components.  Implies rvpt_hip_wait.
 * Partitioned image (tile_world > 1):
 *   - with a communicator (rvpt_hip_comm_init / rvpt_hip_comm_init_all) the call is COLLECTIVE: every rank calls it, the
 *     per-tile radiance is gathered to rank 0 over RCCL and un-tiled there; rank 0 receives the whole frame, the other ranks
 *     only send (their dst may be NULL and is not written).  In a single-process group only rank 0's context is called;
 *   - without one, pixels of tiles this rank does not own read as 0 (a host doing its own exchange uses
 *     rvpt_hip_tile_buffer / rvpt_hip_untile). */
int rvpt_hip_read(rvpt_hip_ctx *ctx, int format, void *dst, size_t dst_bytes);

/* ---- multi-GPU: one RCCL communicator over the tile_world ranks of a partitioned image (no reference counterpart: the
 * reference is single-device; SURVEY §8(b) "creates streams (+ RCCL comm if n_devices>1)", §8(e)) -------------------
 * The only exchange of the path is the gather above — grouped ncclSend/ncclRecv of each rank's tile-linear accumulator
 * (xGMI: every peer on its own link to the root), nothing per frame.  librccl is loaded on first use.
 *
 * One process per GPU: rank 0 makes an id (comm_unique_id), the host hands the 128 bytes to every rank by whatever
 * means it has (MPI, a file, torch.distributed's store), every rank calls comm_init on its context; rank and world are
 * the context's tile_rank / tile_world.
 * One process, several GPUs: create one context per device (tile_rank i of n, any device ids) and pass them, in rank
 * order, to comm_init_all (ncclCommInitAll); collectives are then driven through rank 0's context alone. */
#define RVPT_HIP_COMM_ID_BYTES 128
int rvpt_hip_comm_unique_id(void *id_out, size_t id_bytes);
int rvpt_hip_comm_init(rvpt_hip_ctx *ctx, const void *unique_id, size_t id_bytes);
int rvpt_hip_comm_init_all(rvpt_hip_ctx *const *ctxs, int n);
/* Collective failure behaviour (ABI 4).  Arguments, allocations and this rank's own frames in flight are dealt with BEFORE a rank
 * enters a group call; a rank that finds its own arguments invalid still takes part in the exchange and reports the error afterwards
 * (it never leaves its peers waiting).  Every collective — and ncclCommInitRank's bootstrap — is given RVPT_HIP_COMM_TIMEOUT_S seconds
 * (default 120): when a peer never arrives the call returns RVPT_HIP_ERR_COMM with the reason in rvpt_hip_last_error, the communicator
 * is aborted and later collectives on the context report "no communicator".  rendering is unaffected.
 *
 * comm_barrier: every rank's work dispatched so far has finished when it returns (rvpt_hip_wait on this rank, then a one-float
 * all-reduce on the communicator: ~30 us warm).  What a host uses to bracket a timed region without a second RCCL communicator of
 * its own (bench.py).  Single-process groups: through rank 0's context. */
int rvpt_hip_comm_barrier(rvpt_hip_ctx *ctx);
/* Leave the communicator (ncclCommDestroy; rvpt_hip_destroy does it too): the context is a plain partition member again, its reads
 * are local.  For hosts whose ranks did not all manage to join.  A single-process group dissolves as a whole. */
int rvpt_hip_comm_destroy(rvpt_hip_ctx *ctx);
/* (ABI 8) What RCCL itself says about the context's communicator — ncclCommCount, ncclCommUserRank, ncclGetVersion (e.g. 22105 = 2.21.5) — so that a
 * scaling record can state that the gather ran over RCCL with N ranks (bench.py: "collective").  RVPT_HIP_ERR_COMM without a communicator; an entry point
 * this RCCL lacks leaves its output 0 / -1.  Any out pointer may be NULL. */
int rvpt_hip_comm_info(rvpt_hip_ctx *ctx, int *n_ranks, int *rank, int *rccl_version);
/* The same gather, leaving the frame on the device: rank 0 passes width*height*16 bytes of its own device memory
 * (row-major RGBA32F); the other ranks pass NULL.  Collective like rvpt_hip_read. */
int rvpt_hip_gather(rvpt_hip_ctx *ctx, void *dst_dev_rgba32f);

/* Multi-GPU plumbing (no reference counterpart; the reference is single-device).
 * tile_buffer: device pointer + byte size of this rank's tile-linear RGBA32F accumulator
 * (owned tiles in ascending slot order, 16x16x4 floats each) — the RCCL gather payload.
 * max_tile_bytes: the same size for the rank that owns most tiles (gather slot size).
 * untile: scatter `n_ranks` gathered slots (device memory, slot r at r*slot_bytes) into a
 * row-major RGBA32F image on this context's device. */
int rvpt_hip_tile_buffer(rvpt_hip_ctx *ctx, void **device_ptr, size_t *bytes,
                         size_t *max_tile_bytes);
int rvpt_hip_untile(rvpt_hip_ctx *ctx, const void *gathered_dev, size_t slot_bytes,
                    uint32_t n_ranks, void *dst_dev_rgba32f);

/* Restore / snapshot the accumulator from host memory (row-major RGBA32F); enables resume of a
 * long accumulation.  (No reference counterpart: its temporal image dies with the process.) */
int rvpt_hip_write_accum(rvpt_hip_ctx *ctx, const void *src_rgba32f, size_t src_bytes);

/* Timing + counters (replaces Timer, src/rvpt/timer.cpp:15-46).  kernel_ms_last: hipEvent time
 * of the last frame-kernel launch; kernel_ms_sum / n_dispatches (launches: a dispatch_frames call is
 * one launch over its n frames) since create or reset. */
int rvpt_hip_get_timing(rvpt_hip_ctx *ctx, float *kernel_ms_last, double *kernel_ms_sum,
                        uint64_t *n_dispatches);
int rvpt_hip_reset_timing(rvpt_hip_ctx *ctx);
/* stats[0] = path segments traced, stats[1] = samples traced (needs RVPT_HIP_COUNT_SEGMENTS). */
int rvpt_hip_get_stats(rvpt_hip_ctx *ctx, uint64_t stats[2]);

/* Launch shape of the last dispatched frame kernel: work-groups, dynamic LDS bytes per work-group,
 * kernel variant (0 brute/LDS-resident with mixed packets, 1 brute/LDS-streamed, 2 bvh: binary per-lane walk, 3 the same with the scene in LDS,
 * 6 brute/LDS-resident packet kernel, 10 bvh over the 4-wide regrouping of the tree, 11 the same with the scene in LDS (and camera packets in the lean
 * configuration), 12 / 13 the opt-in 8-wide walk / 4-wide walk over 64-byte quantised nodes (RVPT_HIP_BVH_WIDE8=1 / RVPT_HIP_BVH_QUANT=1: bit-exact, measured
 * slower); 6, 10 and 11 are the defaults; 4, 5, 7, 8 and 9 were experiments of rounds 3-4 and are retired), and how many frames the context
 * keeps in flight (the reference: MAX_FRAMES_IN_FLIGHT = 2, rvpt.h:25).  Any out pointer may be NULL. */
int rvpt_hip_get_launch_info(rvpt_hip_ctx *ctx, uint32_t *grid_blocks, uint32_t *lds_bytes,
                             uint32_t *kernel_variant, uint32_t *frames_in_flight);

/* (ABI 8) Which exact culls of the packet kernel the last dispatched launch rode with: bit 0 = the screen rectangles of the camera rounds,
 * bit 1 = the bounce cull's table (absent when the scene has none or the launch camera is further than 64 scene scales from the origin: the table's premise),
 * bit 2 = the launch's work plan starts every camera round on a 16 x 4 block (where it does not, the kernel skips the rectangles for that round), bit 4 = the leaf
 * boxes of the bounce rounds (with the table; RVPT_HIP_PACKETS_BOX_CULL=0 switches them off), bit 5 = the interleaved claim order (a frame's blocks dealt from all
 * over the frame; RVPT_HIP_PACKETS_INTERLEAVE=0 gives the tile-linear order), bit 6 = the kernel instance for launches with all three culls (the walks without
 * a cull compiled out: fewer registers to keep alive).  0 for
 * every other kernel.  The image never depends on these; tools/fuzz_culls.py records them. */
int rvpt_hip_get_cull_info(rvpt_hip_ctx *ctx, uint32_t *flags);

const char *rvpt_hip_last_error(rvpt_hip_ctx *ctx);

/* Host-side binned-SAH BVH build with the reference node layout (replaces
 * BinnedBvhBuilder::build_bvh, src/rvpt/bvh_builder.cpp:11-199; called once at init,
 * rvpt.cpp:83-86).  nodes_out must hold 2*n_tris-1 nodes; prim_indices_out n_tris entries
 * (leaf order -> original triangle index, i.e. Bvh::primitive_indices).  No GPU needed. */
int rvpt_bvh_build(const rvpt_triangle *tris, size_t n_tris, rvpt_bvh_node *nodes_out,
                   size_t *n_nodes_out, uint32_t *prim_indices_out);

#ifdef __cplusplus
}
#endif
#endif /* RVPT_HIP_H */

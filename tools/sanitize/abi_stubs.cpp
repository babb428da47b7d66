#include "../../include/rvpt_hip.h"
extern "C" {
int rvpt_hip_create(rvpt_hip_ctx **, int, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t) { return -4; }
void rvpt_hip_destroy(rvpt_hip_ctx *) {}
int rvpt_hip_upload_scene(rvpt_hip_ctx *, const rvpt_bvh_node *, size_t, const rvpt_triangle *, size_t, const rvpt_material *, size_t) { return -4; }
int rvpt_hip_set_frame(rvpt_hip_ctx *, const rvpt_render_settings *, const rvpt_camera_data *) { return -4; }
int rvpt_hip_dispatch(rvpt_hip_ctx *) { return -4; }
int rvpt_hip_dispatch_frames(rvpt_hip_ctx *, uint32_t) { return -4; }
int rvpt_hip_wait(rvpt_hip_ctx *) { return -4; }
int rvpt_hip_read(rvpt_hip_ctx *, int, void *, size_t) { return -4; }
const char *rvpt_hip_last_error(rvpt_hip_ctx *) { return "stub"; }
}

// rt_lbvh.hpp — device-side triangle-BVH builder (rt_lbvh.hip), called by rt_scene_commit when the scene's
// builder is RT_BUILDER_GPU_LBVH.  H must have been built with gpu_bvh set (slot / node layout only).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "rt_amd.h"
#include "rt_core.hpp"
#include "rt_scene.hpp"

namespace rt {
// On RT_OK: *d_tris (3 q4 per slot) and *d_nodes (2 q4 per node) are device allocations owned by the caller,
// *max_depth = inner nodes on the longest root-to-leaf path; the builder shortens its Morton codes until that
// is <= max_ok or it runs out of options (the caller compares the result with its stack bound),
// *seconds = wall time of the build including the triangle upload.
int lbvh_build(const HostScene& H, hipStream_t stream, uint32_t max_ok, q4** d_tris, q4** d_nodes, uint32_t* max_depth, double* seconds, std::string& err);
}

// lr_lbvh.h -- device-side BVH build (lr_lbvh.hip), see SURVEY 8(f4).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include "../../include/lumilly_hip.h"

namespace lr {
// Builds an LBVH over host_prims (n >= 2) on the device and writes the traversal layout:
//   d_nodes  4 float4 per inner node (n - 1 nodes, node 0 = root)
//   d_prims  3 float4 per primitive in leaf order
// height_out = inner nodes on the longest root-to-leaf path (traversal stack need), ms_out = device time.
int lbvh_build(const LrPrimitive* host_prims, int n, const float* extra_point, hipStream_t st,
               float4* d_nodes, float4* d_prims, int* height_out, double* ms_out, std::string& err);
}  // namespace lr

// rt_bvh_device.h — launcher of the GPU BVH build (rt_bvh_device.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtd {

// Builds the threaded BVH over n triangles (host arrays: 9 floats, primitive id and material per triangle) on `stream`:
// nodes_dev receives (ordered ? 8 : 1) arrays of 2n-1 nodes (two float4 each, layout of rt_device.h), cnodes_dev the same
// tree in 16-byte nodes (layout of rt_bvh.h; q_params_host receives q_lo[3], q_scale[3]), tris_dev the n triangle
// records (three float4 each) in leaf order.  order_host[i] = input index of the triangle whose record is
// tris_dev[3i..3i+2].  Returns with the stream idle.  Temporary device memory is allocated and freed inside.
hipError_t build_bvh_device(hipStream_t stream, const float *tri_v_host, const int32_t *prim_host, const int32_t *mat_host, int n, bool ordered,
                            float4 *nodes_dev, uint4 *cnodes_dev, float4 *tris_dev, uint32_t *order_host, float *q_params_host);

} // namespace rtd

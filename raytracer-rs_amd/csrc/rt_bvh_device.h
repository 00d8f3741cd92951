// rt_bvh_device.h — launcher of the GPU BVH build (rt_bvh_device.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtd {

// Builds the threaded BVH over n triangles (host arrays: 9 floats, primitive id and material per triangle) on `stream`:
// nodes_dev receives (ordered ? 8 : 1) arrays of 2n-1 nodes (two float4 each, layout of rt_device.h), cnodes_dev the same
// tree in 16-byte nodes (layout of rt_bvh.h; q_params_host receives q_lo[3], q_scale[3]), tris_dev the n triangle
// records (three float4 each) in leaf order.  order_host[i] = input index of the triangle whose record is
// tris_dev[3i..3i+2].  The 16-byte copy is a smaller tree than the 2n-1 nodes of the first (pairs of sibling leaves become
// one leaf of two triangles, inner nodes that cover most of their parent are dissolved into it; `widen` = the share of the
// parent's surface from which on, 0 = keep the binary tree, and `pair_leaves`); `rotate`: the bottom-up pass also applies, at
// every node, the tree rotation that shrinks a child's box most (both copies): *n_cnodes receives the node count, which
// is also the stride of its (ordered ? 8 : 1) arrays.  Returns with the stream idle.  Temporary device memory is allocated
// and freed inside.
hipError_t build_bvh_device(hipStream_t stream, const float *tri_v_host, const int32_t *prim_host, const int32_t *mat_host, int n, bool ordered,
                            float4 *nodes_dev, uint4 *cnodes_dev, float4 *tris_dev, uint32_t *order_host, float *q_params_host, float widen,
                            bool pair_leaves, bool rotate, int *n_cnodes);

} // namespace rtd

// rt_bvh.h — host-side builder of the threaded (stackless) BVH the device kernels walk.
//
// Takes the place of BVHTree::build_tree (src/mesh.rs:225-261) for ALL triangles of the world at once.  The
// reference splits at the median of a random axis with one face per leaf; here the split is chosen by a binned
// surface-area heuristic, leaves hold up to 4 faces, and the tree is laid out for a stack-free walk: every node
// carries the index to continue with after a hit (its first child) and after a miss.  Only the set of triangles
// matters for results (the kernels apply the reference's tie rule themselves), so the tree shape is free.
//
// A stack-free walk visits children in a fixed order.  To still look at the nearer child first, the links are written
// once per sign octant of the ray direction (8 copies of the node array, same node numbering in each): a ray picks
// the copy of its octant and then walks front to back along every split axis.
#pragma once

#include <cstdint>
#include <vector>

namespace rtd {

struct BvhBuild {
    std::vector<float> nodes; // n_orders arrays of n_nodes nodes, 8 floats each: xmin, xmax, ymin, ymax, zmin, zmax, link, miss (int bits);
                              // with 8 orders the pair of an axis is (max, min) in the arrays of octants whose direction is negative along it;
                              // link >= 0: first child, link < 0: ~(first_triangle * 8 + count - 1); miss: next node, -1 = walk finished
    std::vector<float> tris;  // 12 floats per triangle in leaf order: v0.xyz, prim, e1.xyz, mat, e2.xyz, 0
    // The same tree once more in 16-byte nodes, for the walk that is bound by the number of cache lines it touches, not by
    // arithmetic: n_orders arrays of n_nodes nodes in the DEPTH-FIRST ORDER OF THAT OCTANT's visiting order, so a node's
    // first child is the next node and only the index to continue with after a miss is stored.  Four words:
    // x, y, z = entry plane | exit plane << 16, quantised to 16 bits over the root box (q_lo + q * q_scale; rounded
    // outwards by one step); w = inner node: index to go on with when the box is missed (n_nodes = walk finished);
    // leaf: 0x80000000 | first_triangle * 8 + count - 1 (a leaf is always followed by the next node).
    std::vector<uint32_t> cnodes;
    float q_lo[3] = {0, 0, 0}, q_scale[3] = {1, 1, 1};
    int n_nodes = 0;
    int n_orders = 1;         // 8: one node array per direction octant (bit a set = component a negative); 1: direction-independent order
    int n_top = 0;            // nodes [0, n_top) are the part of the tree rays meet most often (largest boxes, parents before children)
    int max_depth = 0;
};

// tri_v: 9 floats per triangle; prim_id / mat per triangle.  max_top bounds n_top, max_leaf (1..4) the triangles per leaf,
// max_arity the children per node after widening (2 = keep the binary tree).
void build_bvh(const float *tri_v, const int32_t *prim_id, const int32_t *mat, int n_tri, int max_top, bool ordered, int max_leaf, int max_arity,
               BvhBuild *out);

} // namespace rtd

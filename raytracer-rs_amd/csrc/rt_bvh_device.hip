// rt_bvh_device.hip — the acceleration structure built on the GPU (gfx950).
//
// Takes the place of BVHTree::build_tree (src/mesh.rs:225-261), which the reference runs on the CPU after every
// Mesh::transform (src/mesh.rs:42): a linear BVH over all world triangles — Morton codes of the triangle centroids,
// one radix sort, one thread per inner node finding its range and split from the sorted codes (Karras 2012), a bottom-up
// refit — emitted in the layout the traversal kernels walk (rt_device.h): threaded nodes with hit and miss links, one
// copy of the links per sign octant of the ray direction, triangles in leaf order.  One triangle per leaf; node index
// i in [0, n-1) is inner node i (0 = root), n-1+j is the leaf of the j-th triangle in sorted order.  The tree's shape
// differs from the host builder's (binned SAH, rt_bvh.cpp); results cannot: boxes only cull, the exact triangle tests
// and the tie rule are the kernels'.

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdint>

#include "rt_bvh_device.h"

namespace rtd {
namespace {

#define BB 256

__device__ __forceinline__ int order_int(float f) { // monotonic float -> int
    const int i = __float_as_int(f);
    return i >= 0 ? i : i ^ 0x7FFFFFFF;
}
__device__ __forceinline__ float order_float(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

struct Box6 {
    float mn[3], mx[3];
};

// padded box of every triangle (same padding as the host builder: the exact triangle test may accept a hit a rounding
// error outside the exact box, and flat axis-aligned faces have zero-thickness boxes) + bounds of the centroids
__global__ __launch_bounds__(BB) void k_bvh_prepare(const float *tri_v, int n, Box6 *tbox, float *centroid, int *bounds /* min xyz, max xyz as ordered ints */) {
    __shared__ int s_b[6];
    if (threadIdx.x < 3) s_b[threadIdx.x] = 0x7FFFFFFF;
    else if (threadIdx.x < 6) s_b[threadIdx.x] = (int)0x80000000;
    __syncthreads();
    const int t = blockIdx.x * BB + threadIdx.x;
    if (t < n) {
        const float *v = tri_v + 9 * (size_t)t;
        Box6 b;
        float ext = 0.0f, mag = 0.0f;
        for (int a = 0; a < 3; a++) {
            b.mn[a] = fminf(fminf(v[a], v[3 + a]), v[6 + a]);
            b.mx[a] = fmaxf(fmaxf(v[a], v[3 + a]), v[6 + a]);
            ext = fmaxf(ext, b.mx[a] - b.mn[a]);
            mag = fmaxf(mag, fmaxf(fabsf(b.mn[a]), fabsf(b.mx[a])));
        }
        const float pad = 1e-4f * ext + 1e-5f * mag + 1e-6f;
        for (int a = 0; a < 3; a++) {
            b.mn[a] -= pad;
            b.mx[a] += pad;
            const float c = (v[a] + v[3 + a] + v[6 + a]) * (1.0f / 3.0f);
            centroid[3 * (size_t)t + a] = c;
            atomicMin(&s_b[a], order_int(c));
            atomicMax(&s_b[3 + a], order_int(c));
        }
        tbox[t] = b;
    }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&bounds[threadIdx.x], s_b[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&bounds[threadIdx.x], s_b[threadIdx.x]);
}

__device__ __forceinline__ uint32_t spread3(uint32_t v) { // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

// key = 30-bit Morton code of the centroid << 32 | triangle index: unique, so equal codes need no special case
__global__ __launch_bounds__(BB) void k_bvh_keys(const float *centroid, int n, const int *bounds, unsigned long long *keys) {
    const int t = blockIdx.x * BB + threadIdx.x;
    if (t >= n) return;
    uint32_t q[3];
    for (int a = 0; a < 3; a++) {
        const float lo = order_float(bounds[a]), hi = order_float(bounds[3 + a]);
        const float ext = hi - lo;
        float u = ext > 0.0f ? (centroid[3 * (size_t)t + a] - lo) / ext : 0.0f;
        u = fminf(fmaxf(u * 1024.0f, 0.0f), 1023.0f);
        q[a] = (uint32_t)u;
    }
    const uint32_t code = (spread3(q[0]) << 2) | (spread3(q[1]) << 1) | spread3(q[2]);
    keys[t] = ((unsigned long long)code << 32) | (unsigned long long)(uint32_t)t;
}

// triangle records in sorted (= leaf) order: (v0, prim), (v1 - v0, material), (v2 - v0, 0) — geometry.rs:30-31's two
// subtractions, done once; leaf boxes in node-index space
__global__ __launch_bounds__(BB) void k_bvh_emit_tris(const float *tri_v, const int32_t *prim, const int32_t *mat, const unsigned long long *keys, int n,
                                                      const Box6 *tbox, float4 *tris, Box6 *nbox) {
    const int i = blockIdx.x * BB + threadIdx.x;
    if (i >= n) return;
    const uint32_t t = (uint32_t)keys[i];
    const float *v = tri_v + 9 * (size_t)t;
    tris[3 * (size_t)i] = make_float4(v[0], v[1], v[2], __int_as_float(prim[t]));
    tris[3 * (size_t)i + 1] = make_float4(v[3] - v[0], v[4] - v[1], v[5] - v[2], __int_as_float(mat[t]));
    tris[3 * (size_t)i + 2] = make_float4(v[6] - v[0], v[7] - v[1], v[8] - v[2], 0.0f);
    nbox[(size_t)(n - 1) + i] = tbox[t];
}

__device__ __forceinline__ int common_prefix(const unsigned long long *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return __clzll((long long)(keys[i] ^ keys[j]));
}

// one thread per inner node: its range of sorted keys and where the range splits (highest differing bit)
__global__ __launch_bounds__(BB) void k_bvh_hierarchy(const unsigned long long *keys, int n, int2 *children, int *parent, int2 *range) {
    const int i = blockIdx.x * BB + threadIdx.x;
    if (i >= n - 1) return;
    const int d = common_prefix(keys, n, i, i + 1) - common_prefix(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = common_prefix(keys, n, i, i - d);
    int lmax = 2;
    while (common_prefix(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (common_prefix(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = common_prefix(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) / 2;
        if (common_prefix(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + min(d, 0);
    const int left = min(i, j) == gamma ? (n - 1) + gamma : gamma;
    const int right = max(i, j) == gamma + 1 ? (n - 1) + gamma + 1 : gamma + 1;
    children[i] = make_int2(left, right);
    range[i] = make_int2(min(i, j), max(i, j)); // leaves (sorted positions) under this node
    parent[left] = i;
    parent[right] = i;
    if (i == 0) parent[0] = -1;
}

// Node flags of the thinned tree (the 16-byte copy)
#define NF_DISSOLVED 1u /* inner node: its children stand in its parent's child list, the node itself is not emitted */
#define NF_PAIR 2u      /* inner node over two leaves: emitted as ONE leaf of two triangles */
#define NF_DEAD 4u      /* leaf under such a node: not emitted */
#define NF_WIDE 8u      /* inner node that has taken a dissolved child's children: not dissolved itself (at most four children) */

// bottom-up: the second thread to arrive at an inner node owns it.  Release before the counter, acquire after it: the
// children's boxes (and sizes) were written by other compute units.  The owner
//  (1) may ROTATE the tree below the node (`rotate`): Morton splits know nothing of surfaces, and swapping a child with a
//      grandchild on the other side — the four candidates of a tree rotation — changes one box only, that of the child
//      whose children change; the swap that shrinks that box most is applied (the surface-area cost of the tree falls
//      by the difference).  Both subtrees are complete when the owner arrives, so nothing else touches them;
//  (2) decides how the node's children appear in the thinned tree — a child node whose surface exceeds `widen` x the
//      node's own is dissolved (the host builder's rule for two children: a box test that fails less than half the time
//      costs more than it saves), unless that child has itself taken in a dissolved node's children (chains of dissolved
//      nodes would make lists of dozens of boxes, each tested by every ray that meets the first: measured 3x slower); two
//      leaf children that are neighbours in sorted order become one leaf of two triangles;
//  (3) counts the nodes of its thinned subtree, itself included: sz.
__global__ __launch_bounds__(BB) void k_bvh_refit(int n, int2 *children, int *parent, int *arrived, Box6 *nbox, int *sz, unsigned char *flags, float widen,
                                                  int pair_leaves, int rotate) {
    const int j = blockIdx.x * BB + threadIdx.x;
    if (j >= n) return;
    auto load_children = [&](int i) {
        const volatile int *cp = (const volatile int *)&children[i];
        return make_int2(cp[0], cp[1]);
    };
    auto load_box = [&](int i) {
        const volatile Box6 *b = &nbox[i];
        Box6 r;
        for (int k = 0; k < 3; k++) { r.mn[k] = b->mn[k]; r.mx[k] = b->mx[k]; }
        return r;
    };
    auto unite = [](const Box6 &a, const Box6 &b) {
        Box6 u;
        for (int k = 0; k < 3; k++) { u.mn[k] = fminf(a.mn[k], b.mn[k]); u.mx[k] = fmaxf(a.mx[k], b.mx[k]); }
        return u;
    };
    auto area_of = [](const Box6 &b) {
        const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
        return 2.0f * (dx * dy + dy * dz + dz * dx);
    };
    // flags of x's children and x's own flags and size, from x's current children and box (steps 2 and 3)
    auto finish_node = [&](int x, const Box6 &xbox) {
        const volatile int *cp = (const volatile int *)&children[x];
        const int ch[2] = {cp[0], cp[1]};
        const bool leaf_a = ch[0] >= n - 1, leaf_b = ch[1] >= n - 1;
        const bool pair = pair_leaves && leaf_a && leaf_b && (ch[0] - ch[1] == 1 || ch[1] - ch[0] == 1);
        const unsigned char keep = (unsigned char)(((volatile unsigned char *)flags)[x] & NF_DISSOLVED); // decided by x's parent, later
        if (pair) {
            flags[x] = (unsigned char)(keep | NF_PAIR);
            flags[ch[0]] = NF_DEAD;
            flags[ch[1]] = NF_DEAD;
            sz[x] = 1;
            return;
        }
        const float area = area_of(xbox);
        int total = 1;
        unsigned wide = 0u;
        for (int k = 0; k < 2; k++) {
            if (ch[k] >= n - 1) { flags[ch[k]] = 0; total += 1; continue; }
            const unsigned char f = (unsigned char)(((volatile unsigned char *)flags)[ch[k]] & ~NF_DISSOLVED);
            const int s = ((volatile int *)sz)[ch[k]];
            if (!(f & (NF_PAIR | NF_WIDE)) && widen > 0.0f && area_of(load_box(ch[k])) > widen * area) {
                flags[ch[k]] = (unsigned char)(f | NF_DISSOLVED);
                total += s - 1;
                wide = NF_WIDE;
            } else {
                flags[ch[k]] = f;
                total += s;
            }
        }
        flags[x] = (unsigned char)(keep | wide);
        sz[x] = total;
    };
    int cur = parent[(n - 1) + j];
    while (cur >= 0) {
        __threadfence();
        if (atomicAdd(&arrived[cur], 1) == 0) return;
        __threadfence();
        int2 c = load_children(cur);
        const Box6 ba = load_box(c.x), bb = load_box(c.y);
        const Box6 u = unite(ba, bb);
        nbox[cur] = u;
        if (rotate) {
            // X = the child whose children change, Y = the other child of cur, which changes places with X's child `which`
            float best = 0.0f;
            int best_x = -1, best_which = 0;
            for (int side = 0; side < 2; side++) {
                const int X = side ? c.y : c.x;
                if (X >= n - 1) continue; // a leaf has no children to trade
                const Box6 &by = side ? ba : bb;
                const int2 xc = load_children(X);
                const float ax = area_of(side ? bb : ba);
                const float d0 = area_of(unite(by, load_box(xc.y))) - ax; // Y takes the place of X's first child
                const float d1 = area_of(unite(load_box(xc.x), by)) - ax; // ... of X's second child
                if (d0 < best) { best = d0; best_x = X; best_which = 0; }
                if (d1 < best) { best = d1; best_x = X; best_which = 1; }
            }
            if (best_x >= 0 && best < -1e-6f * area_of(u)) {
                const int X = best_x, Y = (X == c.x) ? c.y : c.x;
                int2 xc = load_children(X);
                const int up = best_which ? xc.y : xc.x; // goes up to cur
                if (best_which) xc.y = Y; else xc.x = Y;
                children[X] = xc;
                if (X == c.x) c.y = up; else c.x = up;
                children[cur] = c;
                parent[Y] = X;
                parent[up] = cur;
                const Box6 xb = unite(load_box(xc.x), load_box(xc.y));
                nbox[X] = xb;
                __threadfence();
                finish_node(X, xb);
                __threadfence();
            }
        }
        finish_node(cur, u);
        cur = parent[cur];
    }
}

// for every inner node and direction octant: is the right child visited first?  Same key as the host builder: children
// sorted along the octant's diagonal, the side the ray comes from first; equal keys keep (left, right).
__global__ __launch_bounds__(BB) void k_bvh_order(int n, const int2 *children, const Box6 *nbox, unsigned char *swap_mask) {
    const int i = blockIdx.x * BB + threadIdx.x;
    if (i >= n - 1) return;
    const int2 c = children[i];
    const Box6 a = nbox[c.x], b = nbox[c.y];
    uint32_t m = 0;
    for (int k = 0; k < 8; k++) {
        float ka = 0.0f, kb = 0.0f;
        for (int ax = 0; ax < 3; ax++) {
            const float sgn = (k >> ax) & 1 ? -1.0f : 1.0f;
            ka += sgn * (a.mn[ax] + a.mx[ax]);
            kb += sgn * (b.mn[ax] + b.mx[ax]);
        }
        if (kb < ka) m |= 1u << k;
    }
    swap_mask[i] = (unsigned char)m;
}

// the threaded node arrays: node `idx` of order k gets its box, its hit link (first child in that order, or the
// leaf's triangle) and its miss link (where the walk goes on after the node's subtree: up until the node is a first
// child, then across to the second)
__global__ __launch_bounds__(BB) void k_bvh_thread(int n, int n_orders, const int2 *children, const int *parent, const unsigned char *swap_mask,
                                                   const Box6 *nbox, float4 *nodes) {
    const int nn = 2 * n - 1;
    const long long g = (long long)blockIdx.x * BB + threadIdx.x;
    if (g >= (long long)nn * n_orders) return;
    const int k = (int)(g / nn), idx = (int)(g - (long long)k * nn);
    const int ko = n_orders == 8 ? k : 0; // a single order: the one of octant 0
    int link;
    if (idx < n - 1) {
        const int2 c = children[idx];
        link = (swap_mask[idx] >> ko) & 1 ? c.y : c.x;
    } else {
        link = ~((idx - (n - 1)) * 8);
    }
    int miss = -1, cur = idx;
    int guard = nn; // (see k_bvh_compact)
    for (int p = parent[cur]; p >= 0 && guard > 0; p = parent[cur], guard--) {
        const int2 c = children[p];
        const bool sw = (swap_mask[p] >> ko) & 1;
        const int first = sw ? c.y : c.x, second = sw ? c.x : c.y;
        if (cur == first) { miss = second; break; }
        cur = p;
    }
    const Box6 b = nbox[idx];
    // (entry plane, exit plane) per axis for the rays of this octant: (max, min) where the direction is negative
    const bool nx = n_orders == 8 && (k & 1), ny = n_orders == 8 && (k & 2), nz = n_orders == 8 && (k & 4);
    float4 *dst = nodes + 2 * ((size_t)k * (size_t)nn + (size_t)idx);
    dst[0] = make_float4(nx ? b.mx[0] : b.mn[0], nx ? b.mn[0] : b.mx[0], ny ? b.mx[1] : b.mn[1], ny ? b.mn[1] : b.mx[1]);
    dst[1] = make_float4(nz ? b.mx[2] : b.mn[2], nz ? b.mn[2] : b.mx[2], __int_as_float(link), __int_as_float(miss));
}

// The 16-byte copy of the tree (layout: rt_bvh.h), thinned: per order, a node that is emitted goes to its position in that
// order's depth-first sequence — the number of emitted nodes before it: per ancestor one if the ancestor is emitted, plus
// the thinned subtree of the sibling if that one is visited first — with its planes quantised to 16 bits over the root
// box, one step outwards, and the position to go on with after a miss (its own position + the size of its thinned
// subtree).  `stride` = sz[0], the number of emitted nodes.
__global__ __launch_bounds__(BB) void k_bvh_compact(int n, int n_orders, const int2 *children, const int *parent, const unsigned char *swap_mask,
                                                    const int2 *range, const Box6 *nbox, const int *sz, const unsigned char *flags, int stride, uint4 *cnodes,
                                                    float *q_params) {
    const int nn = 2 * n - 1;
    const long long g = (long long)blockIdx.x * BB + threadIdx.x;
    if (g >= (long long)nn * n_orders) return;
    const int k = (int)(g / nn), idx = (int)(g - (long long)k * nn);
    const int ko = n_orders == 8 ? k : 0;
    if (g == 0) {
        const Box6 root = nbox[0];
        for (int a = 0; a < 3; a++) {
            const float ext = root.mx[a] - root.mn[a];
            q_params[a] = root.mn[a];
            q_params[3 + a] = ext > 0.0f ? ext / 65535.0f : 1.0f;
        }
    }
    const unsigned char f = flags[idx];
    if (f & (NF_DISSOLVED | NF_DEAD)) return; // not emitted
    // emitted nodes under and including c, as its parent's list sees it
    auto size_of = [&](int c) { return c >= n - 1 ? ((flags[c] & NF_DEAD) ? 0 : 1) : sz[c] - ((flags[c] & NF_DISSOLVED) ? 1 : 0); };
    int pos = 0, cur = idx;
    int guard = nn; // (a consistent tree is at most n deep: the bound only keeps a broken one from hanging the GPU)
    for (int p = parent[cur]; p >= 0 && guard > 0; p = parent[cur], guard--) {
        const int2 c = children[p];
        const bool sw = (swap_mask[p] >> ko) & 1;
        const int first = sw ? c.y : c.x;
        pos += ((flags[p] & NF_DISSOLVED) ? 0 : 1) + (cur == first ? 0 : size_of(first));
        cur = p;
    }
    if (pos >= stride) return; // (never in a consistent tree)
    const Box6 root = nbox[0], b = nbox[idx];
    uint32_t w[3];
    for (int a = 0; a < 3; a++) {
        const float ext = root.mx[a] - root.mn[a];
        const float scale = ext > 0.0f ? ext / 65535.0f : 1.0f;
        const float lo = floorf((b.mn[a] - root.mn[a]) / scale) - 1.0f, hi = ceilf((b.mx[a] - root.mn[a]) / scale) + 1.0f;
        const uint32_t qlo = (uint32_t)fminf(fmaxf(lo, 0.0f), 65535.0f), qhi = (uint32_t)fminf(fmaxf(hi, 0.0f), 65535.0f);
        const bool neg = n_orders == 8 && ((k >> a) & 1);
        w[a] = neg ? (qhi | (qlo << 16)) : (qlo | (qhi << 16));
    }
    uint32_t last;
    if (idx >= n - 1) last = 0x80000000u | (uint32_t)((idx - (n - 1)) * 8);
    else if (f & NF_PAIR) last = 0x80000000u | (uint32_t)((min(children[idx].x, children[idx].y) - (n - 1)) * 8 + 1); // two triangles, neighbours in sorted order
    else last = (uint32_t)(pos + sz[idx]);
    cnodes[(size_t)k * (size_t)stride + (size_t)pos] = make_uint4(w[0], w[1], w[2], last);
}

__global__ void k_bvh_init_bounds(int *bounds) {
    if (threadIdx.x < 3) bounds[threadIdx.x] = 0x7FFFFFFF;
    else if (threadIdx.x < 6) bounds[threadIdx.x] = (int)0x80000000;
}

inline unsigned grid_for(long long n) { return (unsigned)((n + BB - 1) / BB); }

} // namespace

#define BVH_TRY(expr)                 \
    do {                              \
        hipError_t e_ = (expr);       \
        if (e_ != hipSuccess) {       \
            cleanup();                \
            return e_;                \
        }                             \
    } while (0)

hipError_t build_bvh_device(hipStream_t stream, const float *tri_v_host, const int32_t *prim_host, const int32_t *mat_host, int n, bool ordered,
                            float4 *nodes_dev, uint4 *cnodes_dev, float4 *tris_dev, uint32_t *order_host, float *q_params_host, float widen,
                            bool pair_leaves, bool rotate, int *n_cnodes) {
    if (n_cnodes) *n_cnodes = 0;
    if (n <= 0) return hipSuccess;
    const int nn = 2 * n - 1, n_orders = ordered ? 8 : 1;
    float *tri_v = nullptr, *centroid = nullptr;
    int32_t *prim = nullptr, *mat = nullptr;
    Box6 *tbox = nullptr, *nbox = nullptr;
    unsigned long long *keys_in = nullptr, *keys_out = nullptr;
    int *bounds = nullptr, *parent = nullptr, *arrived = nullptr;
    int2 *children = nullptr, *range = nullptr;
    float *q_params = nullptr;
    unsigned char *swap_mask = nullptr, *flags = nullptr;
    int *sz = nullptr;
    void *sort_tmp = nullptr;
    auto cleanup = [&]() {
        for (void *p : {(void *)tri_v, (void *)centroid, (void *)prim, (void *)mat, (void *)tbox, (void *)nbox, (void *)keys_in, (void *)keys_out, (void *)bounds,
                        (void *)parent, (void *)arrived, (void *)children, (void *)range, (void *)q_params, (void *)swap_mask, (void *)flags, (void *)sz, sort_tmp})
            if (p) (void)hipFree(p);
    };
    BVH_TRY(hipMalloc((void **)&tri_v, 9 * sizeof(float) * (size_t)n));
    BVH_TRY(hipMalloc((void **)&prim, sizeof(int32_t) * (size_t)n));
    BVH_TRY(hipMalloc((void **)&mat, sizeof(int32_t) * (size_t)n));
    BVH_TRY(hipMalloc((void **)&centroid, 3 * sizeof(float) * (size_t)n));
    BVH_TRY(hipMalloc((void **)&tbox, sizeof(Box6) * (size_t)n));
    BVH_TRY(hipMalloc((void **)&nbox, sizeof(Box6) * (size_t)nn));
    BVH_TRY(hipMalloc((void **)&keys_in, sizeof(unsigned long long) * (size_t)n));
    BVH_TRY(hipMalloc((void **)&keys_out, sizeof(unsigned long long) * (size_t)n));
    BVH_TRY(hipMalloc((void **)&bounds, 6 * sizeof(int)));
    BVH_TRY(hipMalloc((void **)&parent, sizeof(int) * (size_t)nn));
    BVH_TRY(hipMalloc((void **)&arrived, sizeof(int) * (size_t)n));
    BVH_TRY(hipMalloc((void **)&children, sizeof(int2) * (size_t)n));
    BVH_TRY(hipMalloc((void **)&range, sizeof(int2) * (size_t)n));
    BVH_TRY(hipMalloc((void **)&q_params, 6 * sizeof(float)));
    BVH_TRY(hipMalloc((void **)&swap_mask, (size_t)n));
    BVH_TRY(hipMalloc((void **)&flags, (size_t)nn));
    BVH_TRY(hipMalloc((void **)&sz, sizeof(int) * (size_t)nn));
    size_t sort_bytes = 0;
    BVH_TRY(hipcub::DeviceRadixSort::SortKeys(nullptr, sort_bytes, keys_in, keys_out, n, 0, 62, stream));
    BVH_TRY(hipMalloc(&sort_tmp, sort_bytes ? sort_bytes : 16));

    BVH_TRY(hipMemcpyAsync(tri_v, tri_v_host, 9 * sizeof(float) * (size_t)n, hipMemcpyHostToDevice, stream));
    BVH_TRY(hipMemcpyAsync(prim, prim_host, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, stream));
    BVH_TRY(hipMemcpyAsync(mat, mat_host, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, stream));
    BVH_TRY(hipMemsetAsync(arrived, 0, sizeof(int) * (size_t)n, stream));
    BVH_TRY(hipMemsetAsync(flags, 0, (size_t)nn, stream));
    hipLaunchKernelGGL(k_bvh_init_bounds, dim3(1), dim3(64), 0, stream, bounds);
    hipLaunchKernelGGL(k_bvh_prepare, dim3(grid_for(n)), dim3(BB), 0, stream, tri_v, n, tbox, centroid, bounds);
    hipLaunchKernelGGL(k_bvh_keys, dim3(grid_for(n)), dim3(BB), 0, stream, centroid, n, bounds, keys_in);
    BVH_TRY(hipcub::DeviceRadixSort::SortKeys(sort_tmp, sort_bytes, keys_in, keys_out, n, 0, 62, stream));
    hipLaunchKernelGGL(k_bvh_emit_tris, dim3(grid_for(n)), dim3(BB), 0, stream, tri_v, prim, mat, keys_out, n, tbox, tris_dev, nbox);
    if (n > 1) {
        hipLaunchKernelGGL(k_bvh_hierarchy, dim3(grid_for(n - 1)), dim3(BB), 0, stream, keys_out, n, children, parent, range);
        hipLaunchKernelGGL(k_bvh_refit, dim3(grid_for(n)), dim3(BB), 0, stream, n, children, parent, arrived, nbox, sz, flags, widen, pair_leaves ? 1 : 0, rotate ? 1 : 0);
        hipLaunchKernelGGL(k_bvh_order, dim3(grid_for(n - 1)), dim3(BB), 0, stream, n, children, nbox, swap_mask);
    } else {
        const int minus_one = -1;
        BVH_TRY(hipMemcpyAsync(parent, &minus_one, sizeof(int), hipMemcpyHostToDevice, stream));
        BVH_TRY(hipStreamSynchronize(stream)); // minus_one goes out of scope
    }
    hipLaunchKernelGGL(k_bvh_thread, dim3(grid_for((long long)nn * n_orders)), dim3(BB), 0, stream, n, n_orders, children, parent, swap_mask, nbox, nodes_dev);
    // the size of the thinned tree = the stride of its arrays: known once the bottom-up pass is through
    int stride = 1;
    if (n > 1) {
        BVH_TRY(hipMemcpyAsync(&stride, sz, sizeof(int), hipMemcpyDeviceToHost, stream));
        BVH_TRY(hipStreamSynchronize(stream));
        if (stride < 1 || stride > nn) { cleanup(); return hipErrorUnknown; }
    }
    if (n_cnodes) *n_cnodes = stride;
    hipLaunchKernelGGL(k_bvh_compact, dim3(grid_for((long long)nn * n_orders)), dim3(BB), 0, stream, n, n_orders, children, parent, swap_mask, range, nbox, sz, flags,
                       stride, cnodes_dev, q_params);
    BVH_TRY(hipGetLastError());
    BVH_TRY(hipMemcpyAsync(q_params_host, q_params, 6 * sizeof(float), hipMemcpyDeviceToHost, stream));
    // the sorted order goes back to the host: it needs to know where each triangle's record went
    static_assert(sizeof(unsigned long long) == 8, "64-bit keys");
    unsigned long long *keys_host = nullptr;
    BVH_TRY(hipHostMalloc((void **)&keys_host, sizeof(unsigned long long) * (size_t)n, hipHostMallocDefault));
    hipError_t e = hipMemcpyAsync(keys_host, keys_out, sizeof(unsigned long long) * (size_t)n, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e == hipSuccess)
        for (int i = 0; i < n; i++) order_host[i] = (uint32_t)keys_host[i];
    (void)hipHostFree(keys_host);
    cleanup();
    return e;
}

} // namespace rtd

// rt_bvh.cpp — binned-SAH BVH over all world triangles, emitted as threaded node arrays (see rt_bvh.h).

#include "rt_bvh.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <limits>

namespace rtd {
namespace {

struct Box {
    float mn[3], mx[3];
    void reset() {
        for (int a = 0; a < 3; a++) { mn[a] = std::numeric_limits<float>::max(); mx[a] = -std::numeric_limits<float>::max(); }
    }
    void grow(const float *p) {
        for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], p[a]); mx[a] = std::max(mx[a], p[a]); }
    }
    void grow(const Box &b) {
        for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], b.mn[a]); mx[a] = std::max(mx[a], b.mx[a]); }
    }
    float area() const {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (dx < 0) return 0.0f;
        return 2.0f * (dx * dy + dy * dz + dz * dx);
    }
};

struct TmpNode {
    Box box;
    int left = -1, right = -1; // children (inner): left holds the lower centroids along `axis`
    int axis = -1;             // split axis (-1: the range was halved without looking at positions)
    int first = 0, count = 0;  // triangle range in the permuted order (leaf)
    int depth = 0;
};

struct Builder {
    const float *tri_v;
    int n;
    std::vector<Box> tbox;
    std::vector<float> centroid; // 3 per triangle
    std::vector<int> order;
    std::vector<TmpNode> nodes;
    int max_depth = 0;

    int max_leaf = 2; // measured on MI355X: 1 or 2 triangles per leaf beat 4 (a wave runs the triangle tests of every leaf any of its lanes enters)
    static constexpr int BINS = 16;

    int build(int first, int count, int depth) {
        int id = (int)nodes.size();
        nodes.emplace_back();
        max_depth = std::max(max_depth, depth);
        Box box, cbox;
        box.reset();
        cbox.reset();
        for (int i = first; i < first + count; i++) {
            box.grow(tbox[order[i]]);
            cbox.grow(&centroid[3 * order[i]]);
        }
        nodes[id].box = box;
        nodes[id].depth = depth;
        if (count <= max_leaf && (count == 1 || depth > 48 || !worth_splitting(first, count, box, cbox))) {
            nodes[id].first = first;
            nodes[id].count = count;
            return id;
        }
        int axis = -1;
        int mid = split(first, count, box, cbox, &axis);
        int l = build(first, mid - first, depth + 1);
        int r = build(mid, first + count - mid, depth + 1);
        nodes[id].left = l;
        nodes[id].right = r;
        nodes[id].axis = axis;
        return id;
    }

    // for small leaves: does the best SAH split beat intersecting everything?
    bool worth_splitting(int first, int count, const Box &box, const Box &cbox) {
        float best_cost;
        int axis, bin;
        if (!best_split(first, count, box, cbox, &best_cost, &axis, &bin)) return false;
        return best_cost < (float)count;
    }

    bool best_split(int first, int count, const Box &box, const Box &cbox, float *best_cost, int *best_axis, int *best_bin) {
        float parent_area = box.area();
        *best_cost = std::numeric_limits<float>::max();
        *best_axis = -1;
        if (parent_area <= 0.0f) return false;
        for (int a = 0; a < 3; a++) {
            float ext = cbox.mx[a] - cbox.mn[a];
            if (!(ext > 0.0f)) continue;
            Box bb[BINS];
            int bc[BINS];
            for (int b = 0; b < BINS; b++) { bb[b].reset(); bc[b] = 0; }
            float scale = (float)BINS / ext;
            for (int i = first; i < first + count; i++) {
                int t = order[i];
                int b = std::min(BINS - 1, std::max(0, (int)((centroid[3 * t + a] - cbox.mn[a]) * scale)));
                bb[b].grow(tbox[t]);
                bc[b]++;
            }
            float right_area[BINS];
            int right_cnt[BINS];
            Box acc;
            acc.reset();
            int cnt = 0;
            for (int b = BINS - 1; b > 0; b--) {
                acc.grow(bb[b]);
                cnt += bc[b];
                right_area[b] = acc.area();
                right_cnt[b] = cnt;
            }
            acc.reset();
            cnt = 0;
            for (int b = 0; b < BINS - 1; b++) {
                acc.grow(bb[b]);
                cnt += bc[b];
                if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                float cost = 1.0f + (acc.area() * (float)cnt + right_area[b + 1] * (float)right_cnt[b + 1]) / parent_area;
                if (cost < *best_cost) { *best_cost = cost; *best_axis = a; *best_bin = b; }
            }
        }
        return *best_axis >= 0;
    }

    int split(int first, int count, const Box &box, const Box &cbox, int *split_axis) {
        float cost;
        int axis, bin;
        *split_axis = -1;
        if (best_split(first, count, box, cbox, &cost, &axis, &bin)) {
            float ext = cbox.mx[axis] - cbox.mn[axis];
            float scale = (float)BINS / ext;
            auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](int t) {
                int b = std::min(BINS - 1, std::max(0, (int)((centroid[3 * t + axis] - cbox.mn[axis]) * scale)));
                return b <= bin;
            });
            int mid = (int)(it - order.begin());
            if (mid > first && mid < first + count) { *split_axis = axis; return mid; }
        }
        // all centroids coincide (or a degenerate box): halve the range in its current order
        return first + count / 2;
    }
};

// Tree rotations after the top-down build: a binned split looks at one axis and sixteen planes; swapping a child of a
// node with a grandchild on the other side, or two grandchildren of different sides, sometimes gives smaller boxes.
// Bottom-up over the tree, the swap that shrinks the surface of the boxes it changes most is applied; repeated until
// a pass finds (almost) nothing.  Leaves keep their triangle ranges, inner nodes never relied on theirs.
struct Rotator {
    std::vector<TmpNode> &nodes;
    long long applied = 0;
    static Box unite(const Box &a, const Box &b) {
        Box u = a;
        u.grow(b);
        return u;
    }
    bool inner(int i) const { return nodes[i].left >= 0; }
    void visit(int id) {
        // post-order without recursion: children first
        std::vector<std::pair<int, int>> st{{id, 0}};
        while (!st.empty()) {
            auto &[n, phase] = st.back();
            if (!inner(n)) { st.pop_back(); continue; }
            if (phase == 0) { phase = 1; st.push_back({nodes[n].left, 0}); continue; }
            if (phase == 1) { phase = 2; st.push_back({nodes[n].right, 0}); continue; }
            rotate_at(n);
            st.pop_back();
        }
    }
    void rotate_at(int n) {
        TmpNode &N = nodes[n];
        N.box = unite(nodes[N.left].box, nodes[N.right].box);
        const int c[2] = {N.left, N.right};
        float best = 0.0f;
        int kind = -1, bi = 0, bj = 0;
        // child c[1 - s] <-> grandchild g of c[s]: only c[s]'s box changes
        for (int s = 0; s < 2; s++) {
            if (!inner(c[s])) continue;
            const int g[2] = {nodes[c[s]].left, nodes[c[s]].right};
            for (int j = 0; j < 2; j++) {
                const float d = unite(nodes[c[1 - s]].box, nodes[g[1 - j]].box).area() - nodes[c[s]].box.area();
                if (d < best) { best = d; kind = 0; bi = s; bj = j; }
            }
        }
        // grandchild i of the left child <-> grandchild j of the right child: both children's boxes change
        if (inner(c[0]) && inner(c[1])) {
            const int gl[2] = {nodes[c[0]].left, nodes[c[0]].right}, gr[2] = {nodes[c[1]].left, nodes[c[1]].right};
            for (int i = 0; i < 2; i++)
                for (int j = 0; j < 2; j++) {
                    const float d = unite(nodes[gr[j]].box, nodes[gl[1 - i]].box).area() + unite(nodes[gl[i]].box, nodes[gr[1 - j]].box).area() -
                                    nodes[c[0]].box.area() - nodes[c[1]].box.area();
                    if (d < best) { best = d; kind = 1; bi = i; bj = j; }
                }
        }
        if (kind < 0 || !(best < -1e-6f * N.box.area())) return;
        auto child = [&](int x, int k) -> int & { return k ? nodes[x].right : nodes[x].left; };
        if (kind == 0) {
            int &up = child(c[bi], bj);             // the grandchild that comes up
            int &down = bi ? N.left : N.right;      // the child that goes down (c[1 - bi])
            std::swap(up, down);
            nodes[c[bi]].box = unite(nodes[nodes[c[bi]].left].box, nodes[nodes[c[bi]].right].box);
            nodes[c[bi]].axis = -1;
        } else {
            std::swap(child(c[0], bi), child(c[1], bj));
            for (int s = 0; s < 2; s++) {
                nodes[c[s]].box = unite(nodes[nodes[c[s]].left].box, nodes[nodes[c[s]].right].box);
                nodes[c[s]].axis = -1;
            }
        }
        N.axis = -1;
        applied++;
    }
};

// surface-area cost of the binary tree: box tests weighted by the chance of a random ray meeting the box, leaves by their triangles
inline double sah_cost(const std::vector<TmpNode> &nodes, int root) {
    const double ra = nodes[root].box.area();
    double c = 0.0;
    std::vector<int> st{root};
    while (!st.empty()) {
        const int i = st.back();
        st.pop_back();
        const double a = nodes[i].box.area() / ra;
        if (nodes[i].left >= 0) { c += a; st.push_back(nodes[i].left); st.push_back(nodes[i].right); }
        else c += a * nodes[i].count;
    }
    return c;
}

inline float int_bits(int v) {
    float f;
    std::memcpy(&f, &v, 4);
    return f;
}

} // namespace

void build_bvh(const float *tri_v, const int32_t *prim_id, const int32_t *mat, int n_tri, int max_top, bool ordered, int max_leaf, int max_arity,
               BvhBuild *out) {
    out->nodes.clear();
    out->tris.clear();
    out->n_nodes = out->n_top = out->max_depth = 0;
    out->n_orders = 1;
    if (n_tri <= 0) return;

    Builder b;
    b.tri_v = tri_v;
    b.n = n_tri;
    b.max_leaf = std::min(4, std::max(1, max_leaf));
    b.tbox.resize(n_tri);
    b.centroid.resize(3 * (size_t)n_tri);
    b.order.resize(n_tri);
    for (int t = 0; t < n_tri; t++) {
        const float *v = tri_v + 9 * (size_t)t;
        Box bx;
        bx.reset();
        bx.grow(v);
        bx.grow(v + 3);
        bx.grow(v + 6);
        // Pad every triangle box: the exact triangle test may accept a hit that lies a rounding error outside the
        // exact box (flat, axis-aligned faces have zero-thickness boxes), and culling must never lose such a hit.
        float ext = std::max(std::max(bx.mx[0] - bx.mn[0], bx.mx[1] - bx.mn[1]), bx.mx[2] - bx.mn[2]);
        float mag = 0.0f;
        for (int a = 0; a < 3; a++) mag = std::max(mag, std::max(std::fabs(bx.mn[a]), std::fabs(bx.mx[a])));
        float pad = 1e-4f * ext + 1e-5f * mag + 1e-6f;
        for (int a = 0; a < 3; a++) { bx.mn[a] -= pad; bx.mx[a] += pad; }
        b.tbox[t] = bx;
        for (int a = 0; a < 3; a++) b.centroid[3 * (size_t)t + a] = (v[a] + v[3 + a] + v[6 + a]) * (1.0f / 3.0f);
        b.order[t] = t;
    }
    b.nodes.reserve(2 * (size_t)n_tri);
    int root = b.build(0, n_tri, 0);
    {
        // passes of tree rotations (RT_BVH_ROTATE, 0 = none).  Measured: the surface-area cost falls by 12-14 % on the two
        // big-mesh scenes, the nodes a ray visits by 2 % (the widening below already removes most of what a rotation
        // repairs), the frame by 1 %; the passes take a few milliseconds.
        const char *rv = std::getenv("RT_BVH_ROTATE");
        const int passes = rv && *rv ? std::max(0, std::atoi(rv)) : 8;
        const bool stats = std::getenv("RT_BVH_STATS") != nullptr;
        if (stats) std::fprintf(stderr, "[rt] bvh: %d triangles, %zu nodes, surface-area cost %.3f\n", n_tri, b.nodes.size(), sah_cost(b.nodes, root));
        Rotator rot{b.nodes};
        for (int pass = 0; pass < passes; pass++) {
            const long long before = rot.applied;
            rot.visit(root);
            if (stats) std::fprintf(stderr, "[rt] bvh: rotation pass %d: %lld swaps, cost %.3f\n", pass, rot.applied - before, sah_cost(b.nodes, root));
            if (rot.applied - before < (long long)b.nodes.size() / 1000) break;
        }
        if (passes) { // depths changed
            std::vector<std::pair<int, int>> st{{root, 0}};
            b.max_depth = 0;
            while (!st.empty()) {
                auto [i, d] = st.back();
                st.pop_back();
                b.nodes[i].depth = d;
                b.max_depth = std::max(b.max_depth, d);
                if (b.nodes[i].left >= 0) { st.push_back({b.nodes[i].left, d + 1}); st.push_back({b.nodes[i].right, d + 1}); }
            }
        }
    }

    // Widening.  An inner node X under P only pays for its box test if that test often fails: with k children the walk
    // does 1 + p*k tests through X (p = chance that a ray meeting P also meets X, about area(X) / area(P)) against k
    // without it, so X is dissolved into P's child list when p > (k-1)/k.  The threaded layout does not care how many
    // children a node has.  Rays that start inside a room meet every box that spans the room: those nodes all go.
    // (Measured: thresholds 0.7x to 2.5x this one are all slower, on a wall-dominated and on a mesh-dominated scene.)
    std::vector<std::vector<int>> kids(b.nodes.size());
    for (size_t i = 0; i < b.nodes.size(); i++)
        if (b.nodes[i].left >= 0) kids[i] = {b.nodes[i].left, b.nodes[i].right};
    if (max_arity > 2) {
        std::vector<int> todo{root};
        while (!todo.empty()) {
            const int pnode = todo.back();
            todo.pop_back();
            std::vector<int> &list = kids[pnode];
            const float pa = b.nodes[pnode].box.area();
            for (size_t i = 0; i < list.size();) {
                const int x = list[i];
                const size_t k = kids[x].size();
                if (k >= 2 && list.size() + k - 1 <= (size_t)max_arity && b.nodes[x].box.area() * (float)k > pa * (float)(k - 1)) {
                    const std::vector<int> grand = kids[x]; // x's children take x's place and are looked at in turn
                    kids[x].clear();
                    list.erase(list.begin() + (long)i);
                    list.insert(list.begin() + (long)i, grand.begin(), grand.end());
                } else {
                    i++;
                }
            }
            for (int c : list)
                if (!kids[c].empty()) todo.push_back(c);
        }
    }

    // Visiting order of a node's children for a ray of direction-sign class `order`.  Classes 0..7 are the sign octants
    // (bit a set: the direction's component a is negative): children sorted along the octant's diagonal, the side the
    // ray comes from first, so that a hit found there culls the rest.  Class 8 ignores the direction: larger surface first.
    auto sorted_kids = [&](int id, int order) {
        std::vector<int> v = kids[id];
        auto key = [&](int c) {
            const Box &bx = b.nodes[c].box;
            if (order >= 8) return -bx.area();
            float kx = 0.0f;
            for (int a = 0; a < 3; a++) kx += ((order >> a) & 1 ? -1.0f : 1.0f) * (bx.mn[a] + bx.mx[a]);
            return kx;
        };
        std::stable_sort(v.begin(), v.end(), [&](int x, int y) { return key(x) < key(y); });
        return v;
    };

    // numbering (the same for every order): the first max_top nodes are the ones rays meet most often — starting at
    // the root, always the not yet numbered child with the largest surface (a random ray crosses a box with probability
    // proportional to its surface) — the rest follow depth-first.  The kernels keep the first group in LDS.
    std::vector<int> index(b.nodes.size(), -1);
    int next_index = 0;
    {
        auto cmp = [&](int x, int y) { return b.nodes[x].box.area() < b.nodes[y].box.area(); };
        std::priority_queue<int, std::vector<int>, decltype(cmp)> q(cmp);
        q.push(root);
        while (!q.empty() && next_index < max_top) {
            int id = q.top();
            q.pop();
            index[id] = next_index++;
            for (int c : kids[id]) q.push(c);
        }
    }
    out->n_top = next_index;
    {
        std::vector<int> stack;
        stack.push_back(root);
        while (!stack.empty()) {
            int id = stack.back();
            stack.pop_back();
            if (index[id] < 0) index[id] = next_index++;
            const std::vector<int> v = sorted_kids(id, 8);
            for (size_t i = v.size(); i-- > 0;) stack.push_back(v[i]);
        }
    }
    const int nn = next_index; // dissolved nodes are gone

    // threading, once per order: miss(node) = first node of whatever follows the node's subtree in visiting order
    out->n_orders = ordered ? 8 : 1;
    out->nodes.assign(8 * (size_t)nn * (size_t)out->n_orders, 0.0f);
    struct Item { int id, next; };
    std::vector<Item> stack;
    for (int k = 0; k < out->n_orders; k++) {
        const int order = ordered ? k : 8;
        float *base = &out->nodes[8 * (size_t)nn * (size_t)k];
        stack.push_back({root, -1});
        while (!stack.empty()) {
            Item it = stack.back();
            stack.pop_back();
            const TmpNode &nd = b.nodes[it.id];
            float *dst = base + 8 * (size_t)index[it.id];
            // each pair is (plane the rays of this octant enter through, plane they leave through): (min, max) along an
            // axis whose direction component is positive, (max, min) along a negative one
            for (int a = 0; a < 3; a++) {
                const bool neg = ordered && ((k >> a) & 1);
                dst[2 * a] = neg ? nd.box.mx[a] : nd.box.mn[a];
                dst[2 * a + 1] = neg ? nd.box.mn[a] : nd.box.mx[a];
            }
            dst[7] = int_bits(it.next);
            if (!kids[it.id].empty()) {
                const std::vector<int> v = sorted_kids(it.id, order);
                dst[6] = int_bits(index[v[0]]);
                for (size_t i = 0; i < v.size(); i++) stack.push_back({v[i], i + 1 < v.size() ? index[v[i + 1]] : it.next});
            } else {
                dst[6] = int_bits(~(nd.first * 8 + (nd.count - 1)));
            }
        }
    }

    // ---- the compact copy: per order, nodes in that order's depth-first sequence
    {
        const Box &rb = b.nodes[root].box;
        for (int a = 0; a < 3; a++) {
            out->q_lo[a] = rb.mn[a];
            const float ext = rb.mx[a] - rb.mn[a];
            out->q_scale[a] = ext > 0.0f ? ext / 65535.0f : 1.0f;
        }
        auto quant = [&](float v, int a, bool up) {
            double q = ((double)v - (double)out->q_lo[a]) / (double)out->q_scale[a];
            long long i = up ? (long long)std::ceil(q) + 1 : (long long)std::floor(q) - 1; // one step outwards: the kernel's float evaluation of q_lo + q * q_scale rounds
            return (uint32_t)std::min<long long>(65535, std::max<long long>(0, i));
        };
        out->cnodes.assign(4 * (size_t)nn * (size_t)out->n_orders, 0u);
        std::vector<int> dfs; // nodes of the current order in visiting sequence
        std::vector<int> subtree(b.nodes.size(), 0);
        for (int k = 0; k < out->n_orders; k++) {
            const int order = ordered ? k : 8;
            uint32_t *base = &out->cnodes[4 * (size_t)nn * (size_t)k];
            // iterative pre-order walk; subtree sizes from a post-order pass over the same sequence
            dfs.clear();
            std::vector<int> st{root};
            while (!st.empty()) {
                const int id = st.back();
                st.pop_back();
                dfs.push_back(id);
                if (!kids[id].empty()) {
                    const std::vector<int> v = sorted_kids(id, order);
                    for (size_t i = v.size(); i-- > 0;) st.push_back(v[i]);
                }
            }
            for (size_t i = dfs.size(); i-- > 0;) {
                int sz = 1;
                for (int c : kids[dfs[i]]) sz += subtree[c];
                subtree[dfs[i]] = sz;
            }
            for (size_t i = 0; i < dfs.size(); i++) {
                const TmpNode &nd = b.nodes[dfs[i]];
                uint32_t *dst = base + 4 * i;
                for (int a = 0; a < 3; a++) {
                    const bool neg = ordered && ((k >> a) & 1);
                    const uint32_t lo = quant(nd.box.mn[a], a, false), hi = quant(nd.box.mx[a], a, true);
                    dst[a] = neg ? (hi | (lo << 16)) : (lo | (hi << 16));
                }
                if (kids[dfs[i]].empty()) dst[3] = 0x80000000u | (uint32_t)(nd.first * 8 + (nd.count - 1));
                else dst[3] = (uint32_t)(i + (size_t)subtree[dfs[i]]);
            }
        }
    }

    out->tris.resize(12 * (size_t)n_tri);
    for (int i = 0; i < n_tri; i++) {
        int t = b.order[i];
        const float *v = tri_v + 9 * (size_t)t;
        float *dst = &out->tris[12 * (size_t)i];
        dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2];
        dst[3] = int_bits(prim_id[t]);
        dst[4] = v[3] - v[0]; dst[5] = v[4] - v[1]; dst[6] = v[5] - v[2]; // E1 = v1 - v0 (geometry.rs:30)
        dst[7] = int_bits(mat[t]);
        dst[8] = v[6] - v[0]; dst[9] = v[7] - v[1]; dst[10] = v[8] - v[2]; // E2 = v2 - v0 (geometry.rs:31)
        dst[11] = 0.0f;
    }
    out->n_nodes = nn;
    out->max_depth = b.max_depth;
}

} // namespace rtd

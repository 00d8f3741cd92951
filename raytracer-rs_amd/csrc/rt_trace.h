// rt_trace.h — persistent BVH traversal for wave64: lanes that finish their ray pull the next one.
//
// A ray's walk through the threaded BVH takes anywhere from a handful to a hundred steps.  With one ray per lane for
// the lifetime of a wave, the wave runs as long as its slowest ray and most lanes idle (measured: 24 % of lanes active
// per vector instruction).  Here a wave owns a strided list of 64-ray chunks of the queue and keeps its lanes fed:
//   * when at least REFILL lanes are idle, the finished rays are finalised TOGETHER (spheres / cylinders, result
//     store or shadow accumulation) and the idle lanes take the next unissued rays of the wave's chunks;
//   * a lane that reaches a leaf parks it until at least LEAF_T lanes hold one (or nobody has inner nodes left), so
//     the triangle tests, the expensive divergent part, run with many lanes at once.
// Which primitive a ray reports is independent of all this: the walk visits the same node sequence per ray.
#pragma once

#include "rt_device.h"
#include "rt_kernels.h"

namespace rtd {

#define TRACE_BLOCK 256

struct TraceCursor { // wave-uniform
    int chunk; // current chunk (-1: no work left)
    int base;  // queue slot of the chunk's first ray
    int cnt;   // rays in the chunk
    int pos;   // next unissued ray of the chunk
};

// chunk -> (segment, first slot, count) through the per-launch prefix of chunks per segment (LDS, RT_NSEG+1 entries)
__device__ __forceinline__ void cursor_seek(TraceCursor &w, int chunk, const int *chunk_prefix, const uint32_t *seg_count, uint32_t seg_capacity) {
    if (chunk >= chunk_prefix[RT_NSEG]) {
        w.chunk = -1;
        w.base = 0;
        w.cnt = 0;
        w.pos = 0;
        return;
    }
    int s = 0;
    while (chunk >= chunk_prefix[s + 1]) s++;
    const int local = chunk - chunk_prefix[s];
    w.chunk = chunk;
    w.base = (int)(s * seg_capacity) + local * 64;
    const int left = (int)seg_count[s] - local * 64;
    w.cnt = left < 64 ? left : 64;
    w.pos = 0;
}

// ANY_HIT = false: closest hit of world.objects (hitable.rs:35-62), result (t, code) per ray into hit[slot].
// ANY_HIT = true : occlusion query (light.rs:74,170,200,320); unoccluded rays add their contribution to the frame.
template <bool ANY_HIT, bool COUNT, int REFILL, int LEAF_T>
__device__ __forceinline__ void trace_persistent(const DevScene &sc, const RenderConst &rc, Ctl *ctl, const uint32_t *seg_count_global,
                                                 uint32_t seg_count_stride, uint32_t seg_capacity, const float4 *ray_a, const float4 *ray_b, const float2 *ray_c,
                                                 float4 *hit, float *planes, uint32_t n_pixels, float4 *lds_nodes, int *chunk_prefix,
                                                 uint32_t *seg_count) {
    // ---- per-workgroup prologue: segment counts, chunk prefix, top of the tree
    if (threadIdx.x < RT_NSEG) {
        uint32_t c = seg_count_global[threadIdx.x * seg_count_stride];
        seg_count[threadIdx.x] = c < seg_capacity ? c : seg_capacity;
    }
    for (int j = threadIdx.x; j < 2 * sc.n_top; j += TRACE_BLOCK) lds_nodes[j] = sc.nodes[j];
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int s = 0; s < RT_NSEG; s++) {
            chunk_prefix[s] = acc;
            acc += (int)((seg_count[s] + 63u) >> 6);
        }
        chunk_prefix[RT_NSEG] = acc;
    }
    __syncthreads();

    const int stride = (int)(gridDim.x * (TRACE_BLOCK / 64));
    TraceCursor w;
    cursor_seek(w, __builtin_amdgcn_readfirstlane((int)(blockIdx.x * (TRACE_BLOCK / 64) + (threadIdx.x >> 6))), chunk_prefix, seg_count, seg_capacity);

    const float t_min = rc.t_min;
    const float t_max0 = ANY_HIT ? 1.0f - rc.t_min : rc.t_max;

    // per-lane ray state.  slot < 0: lane holds no ray.  node >= 0: next node to visit; node == -1 with slot >= 0: done, waiting for finalisation
    int slot = -1, node = -1, leaf = -1;
    F3 o = f3(0, 0, 0), d = f3(0, 0, 0), inv = f3(0, 0, 0);
    float best_t = 0.0f;
    int best_tri = -1, best_prim = -1;
    uint32_t nn = 0, nt = 0;

    for (;;) {
        const bool walking = slot >= 0 && (node >= 0 || leaf >= 0);
        const unsigned long long idle_mask = __ballot(!walking);
        const int n_idle = __popcll(idle_mask);
        // refill when enough lanes are idle and there is something to do for them: rays to fetch or results to store
        if (n_idle == 64 || (n_idle >= REFILL && (w.chunk >= 0 || __ballot(!walking && slot >= 0) != 0ull))) {
            // ---- finalise the rays that finished since the last refill (all idle lanes holding a slot)
            if (!walking && slot >= 0) {
                if (ANY_HIT) {
                    bool occluded = best_tri >= 0;
                    for (int i = 0; i < sc.n_sph && !occluded; i++) {
                        float t;
                        occluded = sph_hit(sc.sph[i], o, d, t_min, t_max0, &t);
                    }
                    for (int i = 0; i < sc.n_cyl && !occluded; i++) {
                        float t;
                        int kind;
                        occluded = cyl_hit(sc.cyl + 5 * i, o, d, t_min, t_max0, &t, &kind);
                    }
                    if (!occluded) {
                        const float4 b = ray_b[slot];
                        const float2 c = ray_c[slot];
                        const uint32_t pixel = __float_as_uint(c.y);
                        atomicAdd(planes + pixel, b.z);
                        atomicAdd(planes + n_pixels + pixel, b.w);
                        atomicAdd(planes + 2u * n_pixels + pixel, c.x);
                    }
                } else {
                    // spheres and cylinders in world order; they need t < closest, so a triangle at equal t keeps the hit
                    float closest = best_t;
                    int code = best_tri >= 0 ? hit_code(HIT_KIND_TRI, best_tri) : HIT_NONE;
                    for (int i = 0; i < sc.n_sph; i++) {
                        float t;
                        if (sph_hit(sc.sph[i], o, d, t_min, closest, &t)) { closest = t; code = hit_code(HIT_KIND_SPH, i); }
                    }
                    for (int i = 0; i < sc.n_cyl; i++) {
                        float t;
                        int kind;
                        if (cyl_hit(sc.cyl + 5 * i, o, d, t_min, closest, &t, &kind)) { closest = t; code = hit_code(kind, i); }
                    }
                    hit[slot] = make_float4(closest, __int_as_float(code), 0.0f, 0.0f);
                }
                if (COUNT) {
                    atomicAdd(&ctl->h.nodes, (unsigned long long)nn);
                    atomicAdd(&ctl->h.tris, (unsigned long long)nt);
                    nn = nt = 0;
                }
                slot = -1;
            }
            // ---- hand the idle lanes the next unissued rays of this wave's chunks
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(idle_mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle_mask, 0u));
            int need = n_idle, given = 0;
            while (need > 0 && w.chunk >= 0) {
                int take = w.cnt - w.pos;
                take = take < need ? take : need;
                if (!walking && rank >= given && rank < given + take) slot = w.base + w.pos + (rank - given);
                w.pos += take;
                given += take;
                need -= take;
                if (w.pos >= w.cnt) cursor_seek(w, w.chunk + stride, chunk_prefix, seg_count, seg_capacity);
            }
            if (!walking && slot >= 0) {
                const float4 a = ray_a[slot], b = ray_b[slot];
                o = f3(a.x, a.y, a.z);
                d = f3(a.w, b.x, b.y);
                inv = f3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
                best_t = t_max0;
                best_tri = -1;
                best_prim = -1;
                node = sc.n_nodes > 0 ? 0 : -1;
                leaf = -1;
            }
            if (__ballot(slot >= 0) == 0ull) break; // no ray anywhere in the wave and none left to fetch
            if (__ballot(slot >= 0 && (node >= 0 || leaf >= 0)) == 0ull) continue; // e.g. empty BVH: finalise right away
        }

        // ---- one step down the threaded BVH for every lane that has an inner node to look at
        const bool stepping = slot >= 0 && leaf < 0 && node >= 0;
        if (stepping) {
            float4 na, nb;
            if (node < sc.n_top) {
                na = lds_nodes[2 * node];
                nb = lds_nodes[2 * node + 1];
            } else {
                na = sc.nodes[2 * node];
                nb = sc.nodes[2 * node + 1];
            }
            if (COUNT) nn++;
            const float x0 = (na.x - o.x) * inv.x, x1 = (nb.x - o.x) * inv.x;
            const float y0 = (na.y - o.y) * inv.y, y1 = (nb.y - o.y) * inv.y;
            const float z0 = (na.z - o.z) * inv.z, z1 = (nb.z - o.z) * inv.z;
            const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), t_min));
            const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), best_t));
            int next = __float_as_int(nb.w);
            if (tn <= tf * 1.00001f + 1e-7f) {
                const int link = __float_as_int(na.w);
                if (link >= 0) next = link;
                else leaf = ~link; // park the leaf; the walk resumes at `next` once its triangles are tested
            }
            node = next;
        }

        // ---- triangle tests, once enough lanes hold a leaf (or nothing else is left to do in this wave)
        const unsigned long long leaf_mask = __ballot(leaf >= 0);
        if (leaf_mask != 0ull) {
            const unsigned long long inner_mask = __ballot(slot >= 0 && leaf < 0 && node >= 0);
            if (__popcll(leaf_mask) >= LEAF_T || inner_mask == 0ull) {
                if (leaf >= 0) {
                    const int first = leaf >> 3, count = (leaf & 7) + 1;
                    for (int k = 0; k < count; k++) {
                        const float4 *tp = sc.tris + 3 * (first + k);
                        const float4 a = tp[0], b = tp[1], c = tp[2];
                        if (COUNT) nt++;
                        float t;
                        if (tri_hit(f3(a.x, a.y, a.z), f3(b.x, b.y, b.z), f3(c.x, c.y, c.z), o, d, t_min, t_max0, false, &t)) {
                            const int prim = __float_as_int(a.w);
                            if (ANY_HIT) {
                                best_tri = first + k;
                                node = -1; // occluded: the walk is over
                                break;
                            }
                            // equal t: the later primitive of the world list wins (hitable.rs:44-50 + geometry.rs:53)
                            if (t < best_t || (t == best_t && prim > best_prim)) { best_t = t; best_tri = first + k; best_prim = prim; }
                        }
                    }
                    leaf = -1;
                }
            }
        }
    }
}

} // namespace rtd

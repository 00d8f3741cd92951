// rt_kernels.hip — the wavefront stages as HIP kernels for gfx950 (wave64).
//
//   k_plan      one thread: closes the previous iteration's counters and decides how many camera rays to add
//   k_raygen    camera ray generation (src/bin/main.rs:35-39, src/camera.rs:29-34), appended to the live queue
//   k_extend    closest hit of world.objects + first light hit (src/shader.rs:99): stackless BVH walk with the top of
//               the tree in LDS, then spheres / cylinders, then the light geometry
//   k_shade     the body of the shaders (src/shader.rs:22-115, src/material.rs:400-484) in throughput form:
//               emission, BSDF sampling + evaluation, next-event estimation; emits child nodes and shadow rays,
//               compacted with wave ballots and one atomic per wave
//   k_shadow    occlusion queries of the queued shadow rays (src/light.rs:74,170,200,320) + accumulation
//
// Built with -ffp-contract=off: see rt_device.h.

#include "rt_kernels.h"

#include <cstddef>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace rtd {

#define BLOCK 256

// ---- workgroup-level queue append ---------------------------------------------------------------------------------
// Ballot + mbcnt give every lane its rank inside the wave; the four wave totals meet in LDS and ONE returning atomic
// per workgroup reserves the slots in the segment's counter.  Every thread of the workgroup must call this, the
// same number of times, with no divergent exit before it.  K = number of independent element streams appended at
// once (child 0 + child 1 share a reservation; shadow rays use their own counter).
// wave-level variant: one returning atomic per wave.  Every lane of the wave must call it.
__device__ __forceinline__ uint32_t wave_append(bool valid, uint32_t *counter) {
    unsigned long long mask = __ballot(valid);
    uint32_t cnt = (uint32_t)__popcll(mask);
    uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    uint32_t base = 0;
    if (cnt != 0) {
        if ((threadIdx.x & 63) == 0) base = atomicAdd(counter, cnt);
        base = __builtin_amdgcn_readfirstlane(base);
    }
    return base + below;
}

// Append of two path nodes (next queue) and one shadow ray (shadow queue) per lane, each light or heavy: four groups
// (light nodes, heavy nodes, light shadow rays, heavy shadow rays), one counter each in segment `seg_l` of its class.
// Inside a wave the elements keep lane order (ballot + mbcnt) and a wave's elements are contiguous: a wave of camera rays
// is one 8x8 pixel tile, and rays that stay together keep their stores, their later loads and the atomic adds of their
// shadow rays in few cache lines.  The waves of the workgroup are ordered by an LDS counter, and one returning global
// atomic per non-empty group reserves the workgroup's slots.  Every thread must call it.
struct BlockAppendSeg {
    uint32_t cnt[5];
    uint32_t base[4];
};

// `answered` = this lane's occlusion query was answered in place (counted for the statistics, no slot).
// The caller zeroes sm.cnt[0..4] and passes a __syncthreads() before the call; one call per kernel (sm is not reset).
// `meanwhile` runs between the issue of the workgroup's reserving atomics and the first use of what they return: work
// that does not depend on the slots hides their round trip through the memory side.
template <class F>
__device__ __forceinline__ void block_append_seg(BlockAppendSeg &sm, Ctl *ctl, const bool (&valid)[3], const bool (&heavy)[3], bool answered,
                                                 uint32_t seg_l, uint32_t (&slot)[3], F &&meanwhile) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_answered = (uint32_t)__popcll(__ballot(answered));
    unsigned long long m[3][2];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        m[k][0] = __ballot(valid[k] && !heavy[k]);
        m[k][1] = __ballot(valid[k] && heavy[k]);
    }
    const uint32_t n00 = (uint32_t)__popcll(m[0][0]), n01 = (uint32_t)__popcll(m[0][1]);
    const uint32_t cnt0 = n00 + (uint32_t)__popcll(m[1][0]), cnt1 = n01 + (uint32_t)__popcll(m[1][1]);
    const uint32_t cnt2 = (uint32_t)__popcll(m[2][0]), cnt3 = (uint32_t)__popcll(m[2][1]);
    uint32_t wave_base = 0;
    if (lane < 5u) {
        const uint32_t c = lane == 0u ? cnt0 : lane == 1u ? cnt1 : lane == 2u ? cnt2 : lane == 3u ? cnt3 : n_answered;
        if (c) wave_base = atomicAdd(&sm.cnt[lane], c);
    }
    __syncthreads();
    if (threadIdx.x == 4u && sm.cnt[4] != 0u) atomicAdd(&ctl->seg_inline[seg_l].v, sm.cnt[4]);
    uint32_t reserved = 0u;
    if (threadIdx.x < 4u) {
        const uint32_t c = sm.cnt[threadIdx.x];
        uint32_t *counter = threadIdx.x == 0u ? &ctl->seg_out[seg_l].v : threadIdx.x == 1u ? &ctl->seg_out[RT_NSEG_HALF + seg_l].v
                          : threadIdx.x == 2u ? &ctl->seg_shadow[seg_l].v : &ctl->seg_shadow[RT_NSEG_HALF + seg_l].v;
        if (c) reserved = atomicAdd(counter, c);
    }
    meanwhile();
    if (threadIdx.x < 4u) sm.base[threadIdx.x] = reserved;
    __syncthreads();
    const uint32_t b0 = sm.base[0] + (uint32_t)__builtin_amdgcn_readlane((int)wave_base, 0);
    const uint32_t b1 = sm.base[1] + (uint32_t)__builtin_amdgcn_readlane((int)wave_base, 1);
    const uint32_t b2 = sm.base[2] + (uint32_t)__builtin_amdgcn_readlane((int)wave_base, 2);
    const uint32_t b3 = sm.base[3] + (uint32_t)__builtin_amdgcn_readlane((int)wave_base, 3);
    auto below = [&](unsigned long long mask) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u)); };
    slot[0] = heavy[0] ? b1 + below(m[0][1]) : b0 + below(m[0][0]);
    slot[1] = heavy[1] ? b1 + n01 + below(m[1][1]) : b0 + n00 + below(m[1][0]);
    slot[2] = heavy[2] ? b3 + below(m[2][1]) : b2 + below(m[2][0]);
}

// segment of its class a workgroup appends to (any choice is correct; spreading by workgroup keeps the counters apart)
__device__ __forceinline__ uint32_t ray_segment(bool heavy, uint32_t spread) { return (heavy ? RT_NSEG_HALF : 0u) + spread % RT_NSEG_HALF; }

__device__ __forceinline__ void add_pixel(float *planes, uint32_t n_pixels, uint32_t pixel, F3 c) {
    atomicAdd(planes + pixel, c.x);
    atomicAdd(planes + n_pixels + pixel, c.y);
    atomicAdd(planes + 2u * n_pixels + pixel, c.z);
}

// one-dimensional consumer grid -> (segment, workgroup index inside the segment): heavy segments first
__device__ __forceinline__ void decode_block(uint32_t b, uint32_t heavy_blocks, uint32_t *seg, uint32_t *bx) {
    const uint32_t nh = heavy_blocks * RT_NSEG_HALF;
    if (b < nh) {
        *seg = RT_NSEG_HALF + b % RT_NSEG_HALF;
        *bx = b / RT_NSEG_HALF;
    } else {
        b -= nh;
        *seg = b % RT_NSEG_HALF;
        *bx = b / RT_NSEG_HALF;
    }
}

struct Node {
    F3 o, d, w;
    float b_pdf;
    uint32_t meta, nk;
};
__device__ __forceinline__ uint32_t make_meta(uint32_t pixel, uint32_t depth, uint32_t root) { return pixel | (depth << 24) | (root << 30); }
__device__ __forceinline__ F3 load_light(const NodeQueue &q, uint32_t i, uint32_t meta) { // what node i's path has gathered; nothing for a camera ray
    if ((meta >> 30) & 1u) return f3(0.0f, 0.0f, 0.0f);
    const float4 l = q.l[i];
    return f3(l.x, l.y, l.z);
}
// `carry`: the node's gathered light travels with it.  Camera rays (root nodes) carry nothing: k_raygen does not write
// l, and a reader ignores what it finds there.
__device__ __forceinline__ void store_node(const NodeQueue &q, uint32_t i, const Node &n, bool carry = false, F3 light = F3{0.0f, 0.0f, 0.0f}) {
    q.a[i] = make_float4(n.o.x, n.o.y, n.o.z, n.d.x);
    q.b[i] = make_float4(n.d.y, n.d.z, n.w.x, n.w.y);
    q.c[i] = make_float4(n.w.z, n.b_pdf, __uint_as_float(n.meta), __uint_as_float(n.nk));
    if (carry) q.l[i] = make_float4(light.x, light.y, light.z, 0.0f);
}
__device__ __forceinline__ Node load_node(const NodeQueue &q, uint32_t i) {
    float4 a = q.a[i], b = q.b[i], c = q.c[i];
    Node n;
    n.o = f3(a.x, a.y, a.z);
    n.d = f3(a.w, b.x, b.y);
    n.w = f3(b.z, b.w, c.x);
    n.b_pdf = c.y;
    n.meta = __float_as_uint(c.z);
    n.nk = __float_as_uint(c.w);
    return n;
}

// ---- k_plan ---------------------------------------------------------------------------------------------------------
// the bookkeeping of k_plan, by one thread: closes the iteration that just ran and decides the next one's camera rays
__device__ __forceinline__ void plan_head(Ctl *ctl, uint32_t target, const uint32_t *s_in, const uint32_t *s_out, const uint32_t *s_sh, const uint32_t *s_inl) {
    uint32_t prev_in = 0, alive = 0, prev_shadow = 0, max_in[2] = {0, 0}, max_sh[2] = {0, 0};
    for (int k = 0; k < RT_NSEG; k++) {
        const int cls = k / RT_NSEG_HALF;
        prev_in += s_in[k];
        alive += s_out[k];
        prev_shadow += s_sh[k];
        max_in[cls] = s_out[k] > max_in[cls] ? s_out[k] : max_in[cls];
        max_sh[cls] = s_sh[k] > max_sh[cls] ? s_sh[k] : max_sh[cls];
    }
    CtlHead &h = ctl->h;
    // close the iteration that just ran: it extended prev_in nodes, n_alive of which were carried in, the rest new camera rays
    h.rays_closest += prev_in;
    h.rays_shadow += prev_shadow;
    for (int k = 0; k < RT_NSEG_HALF; k++) {
        h.rays_shadow += s_inl[k];
        h.rays_shadow_inline += s_inl[k];
    }
    h.cam_samples += prev_in - h.n_alive;
    unsigned long long remaining = h.total - h.cursor;
    uint32_t n_new = 0;
    if (alive < target && remaining > 0) {
        unsigned long long room = target - alive;
        // whole raygen workgroups, or at least whole waves (a wave is one 8x8 block of one sample: k_raygen relies on it;
        // the total is a multiple of 64 as well)
        room &= room >= RT_RAYGEN_PER_WG ? ~(unsigned long long)(RT_RAYGEN_PER_WG - 1) : ~63ull;
        n_new = (uint32_t)(remaining < room ? remaining : room);
    }
    const unsigned long long chunk_candidates = (unsigned long long)h.per_sample * h.chunk_samples;
    h.first_sample = (uint32_t)(h.cursor / chunk_candidates) * h.chunk_samples;
    h.first_rem = (uint32_t)(h.cursor % chunk_candidates);
    h.cursor += n_new;
    h.n_new = n_new;
    h.n_alive = alive;
    h.max_in_light = max_in[0];
    h.max_in_heavy = max_in[1];
    h.max_sh_light = max_sh[0];
    h.max_sh_heavy = max_sh[1];
    h.done = (alive == 0 && n_new == 0) ? 1u : 0u;
    if (!h.done) h.iterations++;
    h.seq++;
}

__global__ void k_plan(Ctl *ctl, uint32_t target, CtlHead *host_head) {
    if (blockIdx.x != 0) return;
    __shared__ uint32_t s_in[RT_NSEG], s_out[RT_NSEG], s_sh[RT_NSEG], s_inl[RT_NSEG_HALF];
    const uint32_t t = threadIdx.x;
    if (t < RT_NSEG_HALF) {
        s_inl[t] = ctl->seg_inline[t].v;
        ctl->seg_inline[t].v = 0;
    }
    if (t < RT_NSEG) {
        s_in[t] = ctl->seg_in[t].v;
        s_out[t] = ctl->seg_out[t].v;
        s_sh[t] = ctl->seg_shadow[t].v;
        ctl->seg_in[t].v = s_out[t]; // the queue filled by the last iteration becomes the current one
        ctl->seg_out[t].v = 0;
        ctl->seg_shadow_cur[t] = s_sh[t]; // k_shadow of the last iteration runs after this kernel, with exact counts
        ctl->seg_shadow[t].v = 0;
    }
    __syncthreads();
    if (t == 0) plan_head(ctl, target, s_in, s_out, s_sh, s_inl);
    if (!host_head) return;
    // the host waits for seq in its pinned copy: everything else first (a word per thread: stores to host memory are slow
    // one after the other), then a fence, then seq
    __syncthreads();
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&ctl->h);
    volatile uint32_t *dst = reinterpret_cast<volatile uint32_t *>(host_head);
    const uint32_t seq_word = (uint32_t)(offsetof(CtlHead, seq) / sizeof(uint32_t));
    for (uint32_t i = t; i < sizeof(CtlHead) / sizeof(uint32_t); i += blockDim.x)
        if (i != seq_word) dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (t == 0) __hip_atomic_store(&host_head->seq, ctl->h.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- k_raygen -------------------------------------------------------------------------------------------------------
// Camera candidates are numbered chunk -> 8x8 pixel block -> sample inside the chunk -> pixel inside the block (see
// CtlHead).  One wave = one 8x8 block of one sample; a workgroup takes RT_RAYGEN_PER_WG consecutive candidates, every
// thread RT_RAYGEN_PER_THREAD of them (the same pixel in consecutive samples while the chunk lasts), and reserves
// its queue slots ONCE: one returning atomic per class and workgroup and two barriers for 1024 rays.  The rays wait in
// registers for their slots.
struct RaygenAppend {
    uint32_t cnt[2];  // light / heavy rays of the workgroup
    uint32_t base[2]; // the workgroup's first slot per class
};

__global__ __launch_bounds__(BLOCK) void k_raygen(DevScene sc, RenderConst rc, Ctl *ctl, NodeQueue in, const int32_t *blocks, uint32_t blocks_w,
                                                  uint32_t seg_capacity) {
    __shared__ RaygenAppend sm;
    const uint32_t n_new = ctl->h.n_new;
    const uint32_t wg_first = blockIdx.x * RT_RAYGEN_PER_WG;
    if (wg_first >= n_new) return; // whole workgroup idle
    if (threadIdx.x < 2u) sm.cnt[threadIdx.x] = 0u;
    const uint32_t per_sample = ctl->h.per_sample, n_samples = ctl->h.n_samples, chunk_samples = ctl->h.chunk_samples;
    const uint32_t first_rem = ctl->h.first_rem, first_sample = ctl->h.first_sample;
    const uint32_t lane = threadIdx.x & 63u;
    Node n[RT_RAYGEN_PER_THREAD];
    unsigned long long m_light[RT_RAYGEN_PER_THREAD], m_heavy[RT_RAYGEN_PER_THREAD];
    uint32_t n_light = 0, n_heavy = 0; // of this wave
#pragma unroll
    for (int j = 0; j < RT_RAYGEN_PER_THREAD; j++) {
        // the wave's 64 candidates are one sample of one block: everything up to the pixel is the same for its lanes
        const uint32_t g0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wg_first + (uint32_t)j * BLOCK + (threadIdx.x & ~63u)));
        bool valid = false;
        if (g0 < n_new) {
            uint32_t r = first_rem + g0; // < 2^31: the host bounds per_sample * chunk_samples and the refill level
            uint32_t s0 = first_sample;
            uint32_t in_chunk = s0 < n_samples ? min(chunk_samples, n_samples - s0) : 0u;
            while (in_chunk != 0u && r >= per_sample * in_chunk) { // the batch runs over into the next chunk(s)
                r -= per_sample * in_chunk;
                s0 += in_chunk;
                in_chunk = s0 < n_samples ? min(chunk_samples, n_samples - s0) : 0u;
            }
            const uint32_t div = 64u * max(in_chunk, 1u);
            const uint32_t bi = min(r / div, per_sample / 64u - 1u); // in range even if the counters were inconsistent
            const uint32_t within = r - bi * div;                      // a multiple of 64
            const uint32_t s_rel = s0 + (within >> 6);
            const uint32_t blk = blocks ? (uint32_t)blocks[bi] : bi;
            const uint32_t by = blk / blocks_w, bx = blk - by * blocks_w;
            const int x = (int)(bx * 8u + (lane & 7u));
            const int y = (int)(by * 8u + (lane >> 3));
            const int k = rc.spp_begin + (int)s_rel;
            if (in_chunk != 0u && within < div && x < rc.width && y < rc.height && pixel_selected(rc, x, y)) {
                const uint32_t pixel = (uint32_t)(y * rc.width + x);
                const uint32_t pk = pixel_key(rc.seed, pixel);
                const uint32_t nk = root_node_key((uint32_t)k);
                if (camera_sample(rc, x, y, k, pk, nk, &n[j].o, &n[j].d)) {
                    valid = true;
                    n[j].meta = make_meta(pixel, 0u, 1u);
                    n[j].nk = nk;
                }
            }
        }
        const bool heavy = valid && ray_is_heavy(sc, n[j].o, n[j].d, rc.t_min, rc.t_max);
        m_light[j] = __ballot(valid && !heavy);
        m_heavy[j] = __ballot(heavy);
        n_light += (uint32_t)__popcll(m_light[j]);
        n_heavy += (uint32_t)__popcll(m_heavy[j]);
    }
    __syncthreads(); // sm.cnt is zero
    uint32_t wave_base = 0;
    if (lane < 2u) {
        const uint32_t c = lane == 0u ? n_light : n_heavy;
        if (c) wave_base = atomicAdd(&sm.cnt[lane], c);
    }
    __syncthreads();
    const uint32_t seg_l = blockIdx.x % RT_NSEG_HALF, seg_h = RT_NSEG_HALF + seg_l;
    if (threadIdx.x < 2u) {
        const uint32_t c = sm.cnt[threadIdx.x];
        sm.base[threadIdx.x] = c ? atomicAdd(&ctl->seg_in[threadIdx.x == 0u ? seg_l : seg_h].v, c) : 0u;
    }
    __syncthreads();
    uint32_t at_light = sm.base[0] + (uint32_t)__builtin_amdgcn_readlane((int)wave_base, 0);
    uint32_t at_heavy = sm.base[1] + (uint32_t)__builtin_amdgcn_readlane((int)wave_base, 1);
    auto below = [&](unsigned long long mask) { return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u)); };
#pragma unroll
    for (int j = 0; j < RT_RAYGEN_PER_THREAD; j++) {
        const bool l = (m_light[j] >> lane) & 1ull, h = (m_heavy[j] >> lane) & 1ull;
        if (l || h) {
            const uint32_t sl = h ? at_heavy + below(m_heavy[j]) : at_light + below(m_light[j]);
            n[j].w = f3(1.0f, 1.0f, 1.0f);
            n[j].b_pdf = 1.0f;
            if (sl < seg_capacity) store_node(in, (h ? seg_h : seg_l) * seg_capacity + sl, n[j]);
            else ctl->h.overflow = 1;
        }
        at_light += (uint32_t)__popcll(m_light[j]);
        at_heavy += (uint32_t)__popcll(m_heavy[j]);
    }
}

// ---- k_extend -------------------------------------------------------------------------------------------------------
// One-dimensional grid, heavy segments first (decode_block).  Light segments: workgroup (bx, s) handles nodes
// [bx*256, bx*256+256) of segment s, one per thread.  Heavy segments (when sc.deferred): every wave owns `heavy_span`
// consecutive nodes and walks them with walk_range.  hit record: (t_obj, object code)
struct ExtendClient {
    const DevScene &sc;
    const RenderConst &rc;
    const NodeQueue &in;
    float2 *hit;
    uint32_t base; // first slot of the segment
    mutable uint32_t scanned = 0; // exact tests of the scanned list (work counting)
    __device__ __forceinline__ bool fetch(uint32_t k, F3 *o, F3 *d, TriHit *init) const {
        const float4 a = in.a[base + k], b = in.b[base + k];
        *o = f3(a.x, a.y, a.z);
        *d = f3(a.w, b.x, b.y);
        // the triangles outside the big meshes are not in the tree when the loose list is in use: scanned here, and the
        // walk starts with their closest hit (in a room: the wall behind the mesh) as its bound
        if (sc.n_loose >= 0) *init = loose_traverse<false, true>(sc, *o, *d, rc.t_min, rc.t_max, &scanned);
        return true;
    }
    // world.objects.hit (hitable.rs:35-62) once the triangles are done: spheres and cylinders need t < closest
    __device__ __forceinline__ void done(uint32_t k, F3 o, F3 d, const TriHit &h) const {
        float closest = h.t;
        int code = h.prim >= 0 ? hit_code(HIT_KIND_TRI, sc.tri_of_prim[h.prim]) : HIT_NONE;
        for (int i = 0; i < sc.n_sph; i++) {
            float t;
            if (sph_hit(sc.sph[i], o, d, rc.t_min, closest, &t)) { closest = t; code = hit_code(HIT_KIND_SPH, i); }
        }
        for (int i = 0; i < sc.n_cyl; i++) {
            float t;
            int kind;
            if (cyl_hit(sc.cyl + 5 * i, o, d, rc.t_min, closest, &t, &kind)) { closest = t; code = hit_code(kind, i); }
        }
        hit[base + k] = make_float2(closest, __int_as_float(code));
    }
};

template <bool COUNT, bool USE_LDS>
__global__ __launch_bounds__(BLOCK) void k_extend(DevScene sc, RenderConst rc, Ctl *ctl, NodeQueue in, float2 *hit, uint32_t seg_capacity,
                                                  uint32_t heavy_blocks, uint32_t heavy_span) {
    extern __shared__ float4 lds_nodes[];
    uint32_t seg, bx;
    decode_block(blockIdx.x, heavy_blocks, &seg, &bx);
    if (seg >= RT_NSEG_HALF && sc.deferred) {
        const uint32_t n_seg = min(ctl->seg_in[seg].v, seg_capacity);
        const uint32_t wg_first = bx * (BLOCK / 64) * heavy_span;
        if (wg_first >= n_seg) return;
        const uint32_t wave = threadIdx.x >> 6, first = wg_first + wave * heavy_span;
        const uint32_t total = first < n_seg ? min(heavy_span, n_seg - first) : 0u;
        WaveWalkLds *areas = reinterpret_cast<WaveWalkLds *>(reinterpret_cast<char *>(lds_nodes) + wave_area_offset(sc, USE_LDS));
        ExtendClient client{sc, rc, in, hit, seg * seg_capacity};
        uint32_t nn = 0, nt = 0;
        walk_range<false, COUNT>(sc, areas[wave], client, first, total, rc.t_min, rc.t_max, &nn, &nt);
        if (COUNT) {
            atomicAdd(&ctl->h.nodes, (unsigned long long)nn);
            atomicAdd(&ctl->h.tris, (unsigned long long)nt);
            atomicAdd(&ctl->h.scanned, (unsigned long long)client.scanned);
        }
        return;
    }
    const uint32_t il = bx * BLOCK + threadIdx.x;
    const uint32_t i = seg * seg_capacity + il;
    // The ray is fetched before the segment's fill count is known (a wave lives about 2 us, and waiting for the count
    // first adds a round trip to that): the slot exists whether it is filled or not.  The loads are then in flight while
    // the top of the tree is copied into LDS.
    const uint32_t i_safe = seg * seg_capacity + min(il, seg_capacity - 1u);
    const float4 a = in.a[i_safe], b = in.b[i_safe];
    const uint32_t n_seg = min(ctl->seg_in[seg].v, seg_capacity);
    if (bx * BLOCK >= n_seg) return;
    const bool loose = seg < RT_NSEG_HALF && sc.n_loose >= 0; // light rays scan the few triangles outside the big meshes
    if (USE_LDS && !loose) fill_top_cache<BLOCK>(sc, lds_nodes);
    const bool stage_loose = loose && sc.n_loose_boxes > 0; // every lane fetches its own candidates: from LDS (read only when staged)
    if (stage_loose) {
        if (threadIdx.x < 3u * (uint32_t)sc.n_loose) lds_nodes[threadIdx.x] = sc.loose[threadIdx.x];
        __syncthreads();
    }
    if (il >= n_seg) return;
    const F3 o = f3(a.x, a.y, a.z), d = f3(a.w, b.x, b.y);
    float t_obj;
    int code, prim;
    uint32_t nn = 0, nt = 0, ns = 0;
    // light rays (first half of the segments) postpone their leaves, heavy rays do not: see bvh_traverse
    if (loose) world_hit<COUNT, false, false, true>(sc, lds_nodes, o, d, rc.t_min, rc.t_max, &t_obj, &code, &prim, &nn, &nt, &ns);
    else if (seg < RT_NSEG_HALF) world_hit<COUNT, USE_LDS, true>(sc, lds_nodes, o, d, rc.t_min, rc.t_max, &t_obj, &code, &prim, &nn, &nt, &ns);
    else world_hit<COUNT, USE_LDS, false>(sc, lds_nodes, o, d, rc.t_min, rc.t_max, &t_obj, &code, &prim, &nn, &nt, &ns);
    hit[i] = make_float2(t_obj, __int_as_float(code));
    if (COUNT) {
        atomicAdd(&ctl->h.scanned, (unsigned long long)ns);
        atomicAdd(&ctl->h.nodes, (unsigned long long)nn);
        atomicAdd(&ctl->h.tris, (unsigned long long)(nt & 0xFFFFu));
        atomicAdd(&ctl->h.wave_tri_iters, (unsigned long long)(nt >> 16));
        atomicAdd(&ctl->h.hist_ray[min(15, 32 - __clz((int)nn))], 1ull);
        uint32_t wmax = nn;
        for (int sh = 32; sh > 0; sh >>= 1) wmax = max(wmax, (uint32_t)__shfl_xor((int)wmax, sh));
        if (__ffsll(__ballot(1)) - 1 == (int)(threadIdx.x & 63u)) {
            atomicAdd(&ctl->h.hist_wave[min(15, 32 - __clz((int)wmax))], 1ull);
            atomicAdd(&ctl->h.wave_steps, (unsigned long long)wmax);
        }
    }
}

// ---- surface point ---------------------------------------------------------------------------------------------------
template <class SC>
__device__ __forceinline__ void surface_at(const SC &sc, int code, F3 p, F3 *n, int *mat) {
    int idx = hit_index(code);
    switch (hit_kind(code)) {
    case HIT_KIND_TRI: {
        const float4 nm = sc.tri_nrm[idx]; // normalize(E1 x E2) (geometry.rs:59) and the material, from k_tri_normals
        *n = f3(nm.x, nm.y, nm.z);
        *mat = __float_as_int(nm.w);
        break;
    }
    case HIT_KIND_SPH: {
        float4 s = sc.sph[idx];
        *n = (p - f3(s.x, s.y, s.z)) / s.w; // geometry.rs:94
        *mat = sc.sph_info[idx].x;
        break;
    }
    case HIT_KIND_CYL_SIDE: {
        const float *cy = sc.cyl + 5 * idx;
        *n = f3(p.x - cy[0], 0.0f, p.z - cy[1]) / cy[4]; // geometry.rs:155-156
        *mat = sc.cyl_info[idx].x;
        break;
    }
    case HIT_KIND_CYL_TOP:
        *n = f3(0.0f, 1.0f, 0.0f);
        *mat = sc.cyl_info[idx].x;
        break;
    default:
        *n = f3(0.0f, -1.0f, 0.0f);
        *mat = sc.cyl_info[idx].x;
        break;
    }
}

template <class SC>
__device__ __forceinline__ F3 sample_in_triangle(const SC &sc, int tri, Stream &st) { // light.rs:274-288
    float e1 = st.next_f32();
    float e2 = st.next_f32();
    float se1 = sqrtf(e1);
    float bx = 1.0f - se1, by = e2 * se1;
    return bx * ltri_vertex(sc, tri, 0) + by * ltri_vertex(sc, tri, 1) + (1.0f - bx - by) * ltri_vertex(sc, tri, 2);
}

// LightList::visible_d without the occlusion query (light.rs:387-397 -> 83-95, 183-213, 303-332): picks a light,
// samples it, returns direction, radiance and the light's pdf.  The caller queues the shadow ray.
template <class SC>
__device__ __forceinline__ bool nee_sample(const SC &sc, const RenderConst &rc, F3 p, F3 n, uint32_t pk, uint32_t nk, F3 *dir_out,
                                           F3 *radiance, float *p_light) {
    if (sc.n_light == 0) return false;
    Stream st(pk, nk, P_NEE);
    const DevLight L = sc.lights[st.next_index((uint32_t)sc.n_light)];
    F3 spectrum = f3(L.sx, L.sy, L.sz);
    if (L.kind == RT_LIGHT_POINT) {
        F3 dir = f3(L.ox, L.oy, L.oz) - p;
        *dir_out = dir;
        *radiance = div_w(spectrum * dot(normalize(dir), normalize(n)), dot(dir, dir)); // light.rs:79
        *p_light = RTD_ZERO;
        return true;
    }
    if (L.kind == RT_LIGHT_DISK) {
        float a, b;
        if (!disk_sample_d(L.sampler_kind, L.radius, rc.light_samples, rc.ns2_edge, st, &a, &b)) return false;
        F3 origin = f3(L.ox, L.oy, L.oz) + f3(a - L.radius, 0.0f, b - L.radius);
        F3 dir = origin - p;
        float cs = dot(normalize(dir), normalize(n));
        *dir_out = dir;
        *radiance = cs > 0.0f ? spectrum * cs : f3(0.0f, 0.0f, 0.0f);
        *p_light = disk_pdf(L, dir, dot(dir, dir));
        return true;
    }
    if (L.tri_count <= 0) return false;
    int tri = L.tri_first + (int)st.next_index((uint32_t)L.tri_count);
    F3 origin = sample_in_triangle(sc, tri, st);
    F3 dir = origin - p;
    float cs = dot(normalize(dir), normalize(n));
    *dir_out = dir;
    *radiance = cs > 0.0f ? spectrum * cs : f3(0.0f, 0.0f, 0.0f);
    float t_l;
    bool h = light_mesh_hit_sampled(sc, L, tri, p, dir, rc.t_min, rc.t_max, &t_l); // self.pdf(&r), light.rs:328,334-342
    *p_light = light_pdf_with_t(L, dir, h, t_l);
    return true;
}

// What shading one node yields: up to two children, one next-event shadow ray (or, for the Whitted shader, the surface
// point whose light samples the caller expands), and the light its ray met.  Nothing is added to the pixel here: the
// caller adds, or lets the path carry.  Of a child only origin, direction, weight and density are filled in; see
// finish_children for the rest.
struct ShadeResult {
    // light this node adds to its path: emission met by its ray, or the 0 * inf / 0 * NaN of a miss with a non-finite
    // weight; the value sits in c0.w (such a node has no children), `ended` says so.
    bool ended = false;
    bool have0 = false, have1 = false, have_s = false;
    Node c0, c1;
    F3 s_o = {0.0f, 0.0f, 0.0f}, s_d = {0.0f, 0.0f, 1.0f}, s_c = {0.0f, 0.0f, 0.0f};
    // Whitted's LightList::visible needs several shadow rays per surface point; they are produced after the main body
    bool vis_pending = false;
    F3 vis_p, vis_n, vis_w;
    uint32_t vis_pk = 0, vis_nk = 0;
};

// Russian roulette (RT_FLAG_RUSSIAN_ROULETTE; the reference has none, shader.rs:94-96 only stops at depth 40): a child at
// depth >= RT_RR_MIN_DEPTH survives with probability q = the largest component of the material's attenuation (at least
// 0.05, at most 1: mirrors and glass always survive) and carries its weight divided by q.  q depends on the material
// alone and the draw on (pixel, sample, node, branch), so the recursive form makes the same decisions.
__device__ __forceinline__ float roulette_q(const rt_material &m) {
    const F3 a = mat_attenuation(m);
    return fminf(1.0f, fmaxf(fmaxf(a.x, fmaxf(a.y, a.z)), 0.05f));
}
__device__ __forceinline__ bool roulette(const rt_material &m, uint32_t pk, uint32_t nk, uint32_t dim, uint32_t child_depth, F3 *w) {
    if (child_depth < RT_RR_MIN_DEPTH) return true;
    const float q = roulette_q(m);
    Stream st(pk, nk, dim);
    if (!(st.next_f32() < q)) return false;
    *w = div_w(*w, q);
    return true;
}

// Cosine-weighted scattering for Diffuse surfaces (RT_FLAG_COSINE_DIFFUSE; the reference samples the hemisphere
// uniformly and calls the density 1/pi, material.rs:85-103,144-151): direction = normalize(n + point on the unit
// sphere), density cos/pi, and the cosine enters the weight of the scattered ray.
__device__ __forceinline__ F3 cosine_scatter(F3 n, uint32_t pk, uint32_t nk) {
    Stream st(pk, nk, P_SCATTER);
    const F3 nn = normalize(n);
    const F3 d = nn + unit_vec_on_sphere(st);
    return dot(d, d) < 1e-12f ? nn : normalize(d);
}
__device__ __forceinline__ float cosine_of(F3 d, F3 n) { return fmaxf(dot(normalize(d), normalize(n)), 0.0f); }

template <int SHADER, bool EXT, class SC>
__device__ __forceinline__ void shade_node(const SC &sc, const RenderConst &rc, float *planes, uint32_t n_pixels, const Node &nd, float2 h,
                                           ShadeResult &r) {
    const float t_obj = h.x;
    const int code = __float_as_int(h.y);
    // world.lights.hit(r) (shader.rs:99): tested here, where every lane of the wave is busy, not in the traversal kernel
    float t_l = 0.0f;
    const int li = lights_hit(sc, nd.o, nd.d, rc.t_min, rc.t_max, &t_l);
    const uint32_t pixel = nd.meta & 0xFFFFFFu, depth = (nd.meta >> 24) & 63u, root = (nd.meta >> 30) & 1u;

    // weight of this node: MIS divides by (light pdf + BSDF pdf) of the ray that led here (material.rs:465-467)
    F3 beta = nd.w;
    if (SHADER == RT_SHADER_MIS && !root) {
        float l_pdf = li >= 0 ? light_pdf_with_t(sc.lights[li], nd.d, true, t_l) : RTD_ZERO; // LightList::pdf, light.rs:399-405
        beta = div_w(nd.w, l_pdf + nd.b_pdf);
    }
    const bool obj = code != HIT_NONE;
    const bool light_wins = li >= 0 && (!obj || !(t_l > t_obj)); // shader.rs:101,108-112
    if (light_wins) {
        const DevLight &L = sc.lights[li];
        r.ended = true;
        r.c0.w = mul_v(beta, f3(L.sx, L.sy, L.sz));
    } else if (!obj) {
        if (!all_finite(beta)) { r.ended = true; r.c0.w = beta * 0.0f; } // 0 * inf, 0 * NaN of the recursive form
    } else {
        const F3 p = nd.o + t_obj * nd.d; // ray.rs:9-11
        F3 n;
        int mat_id;
        surface_at(sc, code, p, &n, &mat_id);
        const rt_material &m = sc.mats[mat_id]; // read where used (LDS when the tables are staged): a copy costs 10 registers for the whole function
        const uint32_t pk = pixel_key(rc.seed, pixel);
        const bool rr = EXT && (rc.flags & RT_FLAG_RUSSIAN_ROULETTE) != 0u;
        const bool cosine = EXT && (rc.flags & RT_FLAG_COSINE_DIFFUSE) != 0u && m.kind == RT_MAT_DIFFUSE;

        if (SHADER == RT_SHADER_MIS) {
            // (next-event estimation first: see the register note at k_shade; the order does not matter to the results,
            // every random number is a function of (pixel, sample, node, purpose))
            F3 ldir, radiance;
            float p_l;
            if (nee_sample(sc, rc, p, n, pk, nd.nk, &ldir, &radiance, &p_l)) { // material.rs:441-447, 473-480
                F3 brdf;
                float pdf;
                mat_brdf_pdf(m, nd.d, ldir, n, rc.t_min, &brdf, &pdf);
                if (cosine) pdf = cosine_of(ldir, n) * (1.0f / RTD_PI);
                F3 direct = div_w(mul_v(radiance, brdf), p_l + pdf);
                F3 c = mul_v(beta, direct);
                if (!is_black(c)) {
                    r.have_s = true;
                    r.s_o = p; r.s_d = ldir; r.s_c = c;
                }
            }
            F3 sd;
            bool scat;
            if (cosine) { sd = cosine_scatter(n, pk, nd.nk); scat = true; }
            else scat = mat_scatter_d(m, nd.d, n, rc.t_min, pk, nd.nk, &sd);
            if (scat) {
                F3 brdf;
                float pdf;
                mat_brdf_pdf(m, nd.d, sd, n, rc.t_min, &brdf, &pdf);
                if (cosine) {
                    const float c = cosine_of(sd, n);
                    brdf = brdf * c;
                    pdf = c * (1.0f / RTD_PI);
                }
                F3 w = mul_v(beta, brdf);
                if (rr && !roulette(m, pk, nd.nk, P_ROULETTE, depth + 1u, &w)) w = f3(0.0f, 0.0f, 0.0f);
                if (depth + 1u <= (uint32_t)rc.max_depth && !is_black(w)) {
                    r.have0 = true;
                    r.c0.o = p; r.c0.d = sd; r.c0.w = w; r.c0.b_pdf = pdf;
                }
            }
            if (m.kind == RT_MAT_MICROFACET) { // the diffuse lobe of material.rs:416-438
                Stream st(pk, nd.nk, P_DIFFUSE);
                F3 dd = unit_vec_on_sphere(st);
                if (dot(dd, n) < 0.0f) dd = -dd;
                F3 f0 = lerp(f0_of(m), albedo_of(m), m.metallic);
                F3 f = fresnel_schlick(dd, n, f0);
                F3 kd = (f3(1.0f, 1.0f, 1.0f) - f) * (1.0f - m.metallic);
                F3 w = mul_v(beta, mul_v(kd, albedo_of(m) * (1.0f / RTD_PI)));
                uint32_t dmin = (uint32_t)max(rc.max_depth - 6, 0); // depth.max(34) for the reference's 40
                uint32_t d1 = max(depth, dmin) + 1u;
                if (rr && !roulette(m, pk, nd.nk, P_ROULETTE1, d1, &w)) w = f3(0.0f, 0.0f, 0.0f);
                if (d1 <= (uint32_t)rc.max_depth && !is_black(w)) {
                    r.have1 = true;
                    r.c1.o = p; r.c1.d = dd; r.c1.w = w;
                }
            }
        } else if (SHADER == RT_SHADER_PATH) { // shader.rs:68-80
            F3 sd;
            bool scat;
            if (cosine) { sd = cosine_scatter(n, pk, nd.nk); scat = true; } // weight albedo/pi * cos / (cos/pi) = the attenuation
            else scat = mat_scatter_d(m, nd.d, n, rc.t_min, pk, nd.nk, &sd);
            if (scat) {
                F3 w = mul_v(beta, mat_attenuation(m));
                if (rr && !roulette(m, pk, nd.nk, P_ROULETTE, depth + 1u, &w)) w = f3(0.0f, 0.0f, 0.0f);
                if (depth + 1u <= (uint32_t)rc.max_depth && !is_black(w)) {
                    r.have0 = true;
                    r.c0.o = p; r.c0.d = sd; r.c0.w = w; r.c0.b_pdf = 1.0f;
                }
            }
        } else { // RT_SHADER_WHITTED, shader.rs:32-45
            F3 sd;
            bool scat = (m.kind == RT_MAT_METAL || m.kind == RT_MAT_DIELECTRIC) && mat_scatter_specular(m, nd.d, n, rc.t_min, &sd);
            F3 w = mul_v(beta, mat_attenuation(m));
            if (scat) {
                if (depth + 1u <= (uint32_t)rc.max_depth && !is_black(w)) {
                    r.have0 = true;
                    r.c0.o = p; r.c0.d = sd; r.c0.w = w; r.c0.b_pdf = 1.0f;
                }
            } else if (!is_black(w)) {
                r.vis_pending = true;
                r.vis_p = p; r.vis_n = n; r.vis_w = w; r.vis_pk = pk; r.vis_nk = nd.nk;
            }
        }
    }
}

// What shade_node leaves out of the children it makes, because it follows from the parent alone and would only occupy
// registers until the children are stored: depth and pixel (meta), the node keys, the constant density of the diffuse
// lobe (material.rs:433-437).  Called where the children are about to be used.
__device__ __forceinline__ void finish_children(const RenderConst &rc, const Node &nd, ShadeResult &r) {
    const uint32_t pixel = nd.meta & 0xFFFFFFu, depth = (nd.meta >> 24) & 63u;
    r.c0.meta = make_meta(pixel, depth + 1u, 0u);
    r.c0.nk = child_node_key(nd.nk, 0u);
    r.c1.meta = make_meta(pixel, max(depth, (uint32_t)max(rc.max_depth - 6, 0)) + 1u, 0u);
    r.c1.nk = child_node_key(nd.nk, 1u);
    r.c1.b_pdf = 1.0f / RTD_PI;
}

// One shadow ray per lane for the Whitted shader's LightList::visible (light.rs:368-385), appended to the light or the
// heavy segment `seg_l` of the shadow queue according to the ray's own class: the occlusion kernel scans only the
// triangles outside the big meshes for the rays of a light segment.  One returning atomic per wave and class; every
// lane of the wave must call it.
__device__ __forceinline__ void append_shadow_classified(const DevScene &sc, const RenderConst &rc, Ctl *ctl, const ShadowQueue &sh, bool valid, F3 o,
                                                         F3 d, F3 c, uint32_t pixel_index, uint32_t seg_l, uint32_t shadow_seg_capacity) {
    const bool heavy = valid && ray_is_heavy(sc, o, d, rc.t_min, 1.0f - rc.t_min);
    const uint32_t slot_l = wave_append(valid && !heavy, &ctl->seg_shadow[seg_l].v);
    const uint32_t slot_h = wave_append(heavy, &ctl->seg_shadow[RT_NSEG_HALF + seg_l].v);
    if (!valid) return;
    const uint32_t slot = heavy ? slot_h : slot_l;
    if (slot >= shadow_seg_capacity) { ctl->h.overflow = 1; return; }
    const uint32_t i = ray_segment(heavy, seg_l) * shadow_seg_capacity + slot;
    sh.a[i] = make_float4(o.x, o.y, o.z, d.x);
    sh.b[i] = make_float4(d.y, d.z, c.x, c.y);
    sh.c[i] = make_float2(c.z, __uint_as_float(pixel_index));
}

// ---- k_shade ---------------------------------------------------------------------------------------------------------
// grid (blocks, RT_NSEG) like k_extend.  Children go to segment (bx + s) % RT_NSEG of the next queue, which spreads
// every segment's offspring over all segments and keeps them level.
// The small tables of a scene — lights and their triangles, materials, spheres, cylinders — are indexed per lane by the
// shading code (the light a path picked, the material it hit).  From global memory every such access is one more vector
// memory instruction through the texture addresser, about thirty per wave where five are needed for the path state; the
// shading kernel was measured bound by exactly that.  When they fit (almost always) a workgroup copies them into LDS once.
#define RT_TAB_LIGHTS 8
#define RT_TAB_LTRIS 32
#define RT_TAB_MATS 32
#define RT_TAB_SPH 16
#define RT_TAB_CYL 8
struct ShadeTables {
    DevLight lights[RT_TAB_LIGHTS];
    float ltris[9 * RT_TAB_LTRIS];
    rt_material mats[RT_TAB_MATS];
    float4 sph[RT_TAB_SPH];
    int2 sph_info[RT_TAB_SPH];
    float cyl[5 * RT_TAB_CYL];
    int2 cyl_info[RT_TAB_CYL];
};
static inline bool tables_fit(const DevScene &sc) {
    return sc.n_light <= RT_TAB_LIGHTS && sc.n_light_tri <= RT_TAB_LTRIS && sc.n_mat <= RT_TAB_MATS && sc.n_sph <= RT_TAB_SPH && sc.n_cyl <= RT_TAB_CYL;
}
// The scene as the shading code sees it when the tables are staged: same member names as DevScene for what the
// table-reading functions touch, everything else through the conversion to the scene itself.
struct ShadeScene {
    const DevScene &g;
    const DevLight *lights;
    const float *ltris;
    const rt_material *mats;
    const float4 *sph;
    const int2 *sph_info;
    const float *cyl;
    const int2 *cyl_info;
    const float4 *tris;
    const float4 *tri_nrm;
    int32_t n_light, n_sph, n_cyl;
    __device__ __forceinline__ operator const DevScene &() const { return g; }
};
// Staging: stage_issue reads every thread's share of the tables into registers (at most 9 words: the table limits above
// fit two rounds of BLOCK words for light triangles and materials, one for the rest), all loads in flight at once;
// stage_commit writes them to LDS and synchronises.  Every thread of the workgroup must call both.  (Issuing the loads
// before the workgroup knows whether it has any node to shade was measured slower: +3.6 % on the teapot frame.)
struct StagedWords { uint32_t w[9]; };
__device__ __forceinline__ StagedWords stage_issue(const DevScene &sc) {
    static_assert(RT_TAB_LIGHTS * sizeof(DevLight) <= 4 * BLOCK && 36 * RT_TAB_LTRIS <= 8 * BLOCK && RT_TAB_MATS * sizeof(rt_material) <= 8 * BLOCK &&
                  16 * RT_TAB_SPH <= 4 * BLOCK && 20 * RT_TAB_CYL <= 4 * BLOCK, "a thread stages at most two words per table");
    const uint32_t t = threadIdx.x;
    auto ld = [](const void *src, uint32_t bytes, uint32_t i) { return i < bytes / 4u ? static_cast<const uint32_t *>(src)[i] : 0u; };
    StagedWords r;
    r.w[0] = ld(sc.lights, (uint32_t)sc.n_light * (uint32_t)sizeof(DevLight), t);
    r.w[1] = ld(sc.ltris, (uint32_t)sc.n_light_tri * 36u, t);
    r.w[2] = ld(sc.ltris, (uint32_t)sc.n_light_tri * 36u, t + BLOCK);
    r.w[3] = ld(sc.mats, (uint32_t)sc.n_mat * (uint32_t)sizeof(rt_material), t);
    r.w[4] = ld(sc.mats, (uint32_t)sc.n_mat * (uint32_t)sizeof(rt_material), t + BLOCK);
    r.w[5] = ld(sc.sph, (uint32_t)sc.n_sph * 16u, t);
    r.w[6] = ld(sc.sph_info, (uint32_t)sc.n_sph * 8u, t);
    r.w[7] = ld(sc.cyl, (uint32_t)sc.n_cyl * 20u, t);
    r.w[8] = ld(sc.cyl_info, (uint32_t)sc.n_cyl * 8u, t);
    return r;
}
__device__ __forceinline__ ShadeScene stage_commit(const DevScene &sc, ShadeTables &tb, const StagedWords &r) {
    const uint32_t t = threadIdx.x;
    auto st = [](void *dst, uint32_t bytes, uint32_t i, uint32_t v) { if (i < bytes / 4u) static_cast<uint32_t *>(dst)[i] = v; };
    st(tb.lights, (uint32_t)sc.n_light * (uint32_t)sizeof(DevLight), t, r.w[0]);
    st(tb.ltris, (uint32_t)sc.n_light_tri * 36u, t, r.w[1]);
    st(tb.ltris, (uint32_t)sc.n_light_tri * 36u, t + BLOCK, r.w[2]);
    st(tb.mats, (uint32_t)sc.n_mat * (uint32_t)sizeof(rt_material), t, r.w[3]);
    st(tb.mats, (uint32_t)sc.n_mat * (uint32_t)sizeof(rt_material), t + BLOCK, r.w[4]);
    st(tb.sph, (uint32_t)sc.n_sph * 16u, t, r.w[5]);
    st(tb.sph_info, (uint32_t)sc.n_sph * 8u, t, r.w[6]);
    st(tb.cyl, (uint32_t)sc.n_cyl * 20u, t, r.w[7]);
    st(tb.cyl_info, (uint32_t)sc.n_cyl * 8u, t, r.w[8]);
    __syncthreads();
    return ShadeScene{sc, tb.lights, tb.ltris, tb.mats, tb.sph, tb.sph_info, tb.cyl, tb.cyl_info, sc.tris, sc.tri_nrm, sc.n_light, sc.n_sph, sc.n_cyl};
}

// Six waves per SIMD (80 registers): the kernel waits on memory at its start (node, tables), at the queue append and
// behind its atomic adds, and further waves fill those gaps: four waves (102 registers) -> five (material read by
// reference instead of copied) -5 % per frame -> six (next-event estimation evaluated before the scattered rays, so
// that its temporaries are not live together with both children) another -4 %.  The Whitted variant, with its
// loop over light samples, needs 86 registers: five waves.
template <int SHADER, bool LDS_TABLES, bool EXT = false>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(SHADER == RT_SHADER_WHITTED ? 5 : 6, SHADER == RT_SHADER_WHITTED ? 5 : 6))) void k_shade(DevScene sc, RenderConst rc, Ctl *ctl, NodeQueue in, NodeQueue out, const float2 *hit,
                                                 ShadowQueue sh, float *planes, uint32_t n_pixels, uint32_t seg_capacity,
                                                 uint32_t shadow_seg_capacity, uint32_t heavy_blocks) {
    __shared__ BlockAppendSeg sm;
    uint32_t seg, bx;
    decode_block(blockIdx.x, heavy_blocks, &seg, &bx);
    const uint32_t il = bx * BLOCK + threadIdx.x;
    // node and hit record are fetched before the fill count is known, like the ray of k_extend
    const uint32_t i = seg * seg_capacity + min(il, seg_capacity - 1u);
    // carry_in: the nodes read bring the light their paths have gathered; carry_out: the children take it along (else it
    // is added to the pixel here).  They differ only in the one iteration in which the host gives carrying up.
    const bool carry_in = SHADER != RT_SHADER_WHITTED && sc.carry_in != 0, carry = SHADER != RT_SHADER_WHITTED && sc.carry_out != 0;
    const Node nd_early = load_node(in, i);
    const float2 h_early = hit[i];
    const uint32_t n_seg = min(ctl->seg_in[seg].v, seg_capacity);
    if (bx * BLOCK >= n_seg) return;
    const uint32_t oseg_l = (bx + seg) % RT_NSEG_HALF;
    __shared__ typename std::conditional<LDS_TABLES, ShadeTables, char>::type tb;
    if (threadIdx.x < 5u) sm.cnt[threadIdx.x] = 0u; // for block_append_seg; the barrier below (or stage_commit's) publishes it

    ShadeResult r;
    bool answered = false; // a light occlusion query answered in place, see below
    bool s_heavy = false;  // the occlusion query's ray touches the box of a big mesh
    bool heavy0 = false, heavy1 = false;
    bool lit = false; // the occlusion query answered in place found the light visible
    F3 carried = f3(0.0f, 0.0f, 0.0f);
    uint32_t slot[3];
    auto shade = [&](const auto &view) {
        if (il < n_seg) shade_node<SHADER, EXT>(view, rc, planes, n_pixels, nd_early, h_early, r);
        if (carry_in && il < n_seg) carried = load_light(in, i, nd_early.meta); // fetched here, not with the node: three registers less while shading
        // ---- compaction: children into the next queue, shadow rays into the shadow queue, light and heavy rays apart
        s_heavy = SHADER != RT_SHADER_WHITTED && r.have_s && ray_is_heavy(sc, r.s_o, r.s_d, rc.t_min, 1.0f - rc.t_min);
        heavy0 = r.have0 && ray_is_heavy(sc, r.c0.o, r.c0.d, rc.t_min, rc.t_max);
        heavy1 = r.have1 && ray_is_heavy(sc, r.c1.o, r.c1.d, rc.t_min, rc.t_max);
        // A light occlusion query is a scan of the few loose triangles (and the spheres and cylinders): answered here, it
        // costs neither the 40-byte store, nor the occlusion kernel's 40-byte load, nor a slot in the shadow queue.
        const bool in_place = SHADER != RT_SHADER_WHITTED && sc.inline_shadow && sc.n_loose >= 0;
        answered = in_place && r.have_s && !s_heavy;
        const bool nee = SHADER != RT_SHADER_WHITTED && r.have_s;
        const bool valid3[3] = {r.have0, r.have1, nee && !answered};
        const bool heavy3[3] = {heavy0, heavy1, nee && s_heavy};
        // the scan runs while the workgroup's slot reservations are on their way to the memory side and back
        block_append_seg(sm, ctl, valid3, heavy3, answered, oseg_l, slot, [&]() {
            if (in_place && __ballot(answered) != 0ull) {
                uint32_t nn = 0, nt = 0;
                const bool occ = answered && world_occluded<false, false, true>(view, nullptr, r.s_o, r.s_d, rc.t_min, 1.0f - rc.t_min, &nn, &nt);
                lit = answered && !occ;
                if (lit && !carry) add_pixel(planes, n_pixels, tiled_index(rc, nd_early.meta & 0xFFFFFFu), r.s_c);
            }
        });
    };
    if constexpr (LDS_TABLES) shade(stage_commit(sc, tb, stage_issue(sc)));
    else {
        __syncthreads();
        shade(sc);
    }
    const bool have0 = r.have0, have1 = r.have1, have_s = r.have_s, vis_pending = r.vis_pending;
    const Node &c0 = r.c0, &c1 = r.c1;
    const F3 s_o = r.s_o, s_d = r.s_d, s_c = r.s_c, vis_p = r.vis_p, vis_n = r.vis_n, vis_w = r.vis_w;
    finish_children(rc, nd_early, r);
    const uint32_t s_pixel = tiled_index(rc, nd_early.meta & 0xFFFFFFu), vis_pk = r.vis_pk, vis_nk = r.vis_nk;
    const bool nee = SHADER != RT_SHADER_WHITTED && have_s;
    const bool heavy_s = nee && s_heavy;
    const bool queued = nee && !answered;
    const uint32_t seg3[3] = {ray_segment(heavy0, oseg_l), ray_segment(heavy1, oseg_l), ray_segment(heavy_s, oseg_l)};
    // Light of this node: what its ray met plus, carried, what the path brought along and the next-event sample found
    // visible in place.  It travels on with the first child there is; a path that ends here adds it to its pixel.
    F3 light = f3(0.0f, 0.0f, 0.0f);
    if (il < n_seg) {
        if (r.ended) light = c0.w;
        if (carry_in) light = carried + light;
        if (carry && lit) light = light + s_c;
        if (!(carry && (have0 || have1)) && !is_black(light)) add_pixel(planes, n_pixels, tiled_index(rc, nd_early.meta & 0xFFFFFFu), light);
    }
    const F3 none = f3(0.0f, 0.0f, 0.0f);
    if (have0) {
        if (slot[0] < seg_capacity) store_node(out, seg3[0] * seg_capacity + slot[0], c0, carry, light); else ctl->h.overflow = 1;
    }
    if (have1) {
        if (slot[1] < seg_capacity) store_node(out, seg3[1] * seg_capacity + slot[1], c1, carry, have0 ? none : light); else ctl->h.overflow = 1;
    }
    if (SHADER == RT_SHADER_WHITTED) {
        // LightList::visible (light.rs:368-385): every light; a disk light is sampled light_samples times (light.rs:150-181)
        for (int l = 0; l < sc.n_light; l++) {
            const DevLight L = sc.lights[l];
            const F3 spectrum = f3(L.sx, L.sy, L.sz);
            if (L.kind == RT_LIGHT_POINT) {
                F3 dir = f3(L.ox, L.oy, L.oz) - vis_p;
                F3 c = f3(0.0f, 0.0f, 0.0f);
                if (vis_pending) c = mul_v(vis_w, spectrum * dot(normalize(dir), normalize(vis_n)) / dot(dir, dir)); // light.rs:79
                const bool v = vis_pending && !is_black(c);
                append_shadow_classified(sc, rc, ctl, sh, v, vis_p, dir, c, s_pixel, oseg_l, shadow_seg_capacity);
            } else if (L.kind == RT_LIGHT_DISK) {
                const float range = L.radius * 2.0f;
                const bool white = L.sampler_kind == RT_SAMPLER_WHITE || L.sampler_kind == RT_SAMPLER_BLUE;
                // first pass: how many of the sampler's points lie in the disk (actual_sample_num, light.rs:161-163)
                int actual = 0;
                if (vis_pending) {
                    Stream st(vis_pk, vis_nk, P_VISIBLE0 + (uint32_t)l);
                    for (int k = 0; k < rc.light_samples; k++) {
                        float a, b;
                        bool have = false;
                        if (white) {
                            for (int tries = 0; tries < 16 && !have; tries++) {
                                a = st.next_f32() * range;
                                b = st.next_f32() * range;
                                have = in_disk(range, a, b);
                            }
                        } else {
                            have = sampler_point(L.sampler_kind, range, rc.light_samples, rc.ns2_edge, k, st, &a, &b) && in_disk(range, a, b);
                        }
                        if (have) actual++;
                    }
                }
                Stream st(vis_pk, vis_nk, P_VISIBLE0 + (uint32_t)l);
                const float scale = RTD_PI * (L.radius * L.radius) / (float)actual;
                for (int k = 0; k < rc.light_samples; k++) {
                    float a = 0.0f, b = 0.0f;
                    bool have = false;
                    if (vis_pending) {
                        if (white) {
                            for (int tries = 0; tries < 16 && !have; tries++) {
                                a = st.next_f32() * range;
                                b = st.next_f32() * range;
                                have = in_disk(range, a, b);
                            }
                        } else {
                            have = sampler_point(L.sampler_kind, range, rc.light_samples, rc.ns2_edge, k, st, &a, &b) && in_disk(range, a, b);
                        }
                    }
                    F3 origin = f3(L.ox, L.oy, L.oz) + f3(a - L.radius, 0.0f, b - L.radius);
                    F3 dir = origin - vis_p;
                    F3 c = f3(0.0f, 0.0f, 0.0f);
                    if (have) c = mul_v(vis_w, (spectrum * dot(normalize(dir), normalize(vis_n)) / dot(dir, dir)) * scale); // light.rs:171,175-177
                    const bool v = have && !is_black(c);
                    append_shadow_classified(sc, rc, ctl, sh, v, vis_p, dir, c, s_pixel, oseg_l, shadow_seg_capacity);
                }
            }
        }
    } else {
        if (queued) {
            const uint32_t slot_s = slot[2];
            if (slot_s < shadow_seg_capacity) {
                const uint32_t si = seg3[2] * shadow_seg_capacity + slot_s;
                sh.a[si] = make_float4(s_o.x, s_o.y, s_o.z, s_d.x);
                sh.b[si] = make_float4(s_d.y, s_d.z, s_c.x, s_c.y);
                sh.c[si] = make_float2(s_c.z, __uint_as_float(s_pixel));
            } else ctl->h.overflow = 1;
        }
    }
}

// ---- k_shadow --------------------------------------------------------------------------------------------------------
struct ShadowClient {
    const DevScene &sc;
    const RenderConst &rc;
    const ShadowQueue &sh;
    float *planes;
    uint32_t n_pixels;
    uint32_t base; // first slot of the segment
    mutable uint32_t scanned = 0; // exact tests of the scanned list (work counting)
    // world.hit(r, T_MIN, 1 - T_MIN): spheres and cylinders first; false = already occluded, no walk
    __device__ __forceinline__ bool fetch(uint32_t k, F3 *o, F3 *d, TriHit *) const {
        const float4 a = sh.a[base + k], b = sh.b[base + k];
        *o = f3(a.x, a.y, a.z);
        *d = f3(a.w, b.x, b.y);
        bool occ = false;
        for (int i = 0; i < sc.n_sph && !occ; i++) {
            float t;
            occ = sph_hit(sc.sph[i], *o, *d, rc.t_min, 1.0f - rc.t_min, &t);
        }
        for (int i = 0; i < sc.n_cyl && !occ; i++) {
            float t;
            int kind;
            occ = cyl_hit(sc.cyl + 5 * i, *o, *d, rc.t_min, 1.0f - rc.t_min, &t, &kind);
        }
        if (!occ && sc.n_loose >= 0) occ = loose_traverse<true, true>(sc, *o, *d, rc.t_min, 1.0f - rc.t_min, &scanned).tri >= 0; // not in the tree
        return !occ;
    }
    __device__ __forceinline__ void done(uint32_t k, F3, F3, const TriHit &h) const {
        if (h.tri >= 0) return; // occluded
        const float4 b = sh.b[base + k];
        const float2 c = sh.c[base + k];
        add_pixel(planes, n_pixels, __float_as_uint(c.y), f3(b.z, b.w, c.x));
    }
};

template <bool COUNT, bool USE_LDS>
__global__ __launch_bounds__(BLOCK) void k_shadow(DevScene sc, RenderConst rc, Ctl *ctl, ShadowQueue sh, float *planes, uint32_t n_pixels,
                                                  uint32_t shadow_seg_capacity, uint32_t heavy_blocks, uint32_t heavy_span) {
    extern __shared__ float4 lds_nodes[];
    uint32_t seg, bx;
    decode_block(blockIdx.x, heavy_blocks, &seg, &bx);
    if (seg >= RT_NSEG_HALF && sc.deferred) {
        const uint32_t n = min(ctl->seg_shadow_cur[seg], shadow_seg_capacity);
        const uint32_t wg_first = bx * (BLOCK / 64) * heavy_span;
        if (wg_first >= n) return;
        const uint32_t wave = threadIdx.x >> 6, first = wg_first + wave * heavy_span;
        const uint32_t total = first < n ? min(heavy_span, n - first) : 0u;
        WaveWalkLds *areas = reinterpret_cast<WaveWalkLds *>(reinterpret_cast<char *>(lds_nodes) + wave_area_offset(sc, USE_LDS));
        ShadowClient client{sc, rc, sh, planes, n_pixels, seg * shadow_seg_capacity};
        uint32_t nn = 0, nt = 0;
        walk_range<true, COUNT>(sc, areas[wave], client, first, total, rc.t_min, 1.0f - rc.t_min, &nn, &nt);
        if (COUNT) {
            atomicAdd(&ctl->h.nodes_sh, (unsigned long long)nn);
            atomicAdd(&ctl->h.tris_sh, (unsigned long long)nt);
            atomicAdd(&ctl->h.scanned, (unsigned long long)client.scanned);
        }
        return;
    }
    const uint32_t il = bx * BLOCK + threadIdx.x;
    // fetched before the fill count is known, like the ray of k_extend
    const uint32_t i = seg * shadow_seg_capacity + min(il, shadow_seg_capacity - 1u);
    const float4 a = sh.a[i], b = sh.b[i];
    const float2 c = sh.c[i];
    const uint32_t n = min(ctl->seg_shadow_cur[seg], shadow_seg_capacity);
    if (bx * BLOCK >= n) return;
    const bool loose = seg < RT_NSEG_HALF && sc.n_loose >= 0;
    if (USE_LDS && !loose) fill_top_cache<BLOCK>(sc, lds_nodes);
    if (il >= n) return;
    const F3 o = f3(a.x, a.y, a.z), d = f3(a.w, b.x, b.y);
    uint32_t nn = 0, nt = 0, ns = 0;
    const bool occ = loose ? world_occluded<COUNT, false, true>(sc, nullptr, o, d, rc.t_min, 1.0f - rc.t_min, &nn, &nt, &ns)
                           : world_occluded<COUNT, USE_LDS>(sc, lds_nodes, o, d, rc.t_min, 1.0f - rc.t_min, &nn, &nt, &ns);
    if (!occ) add_pixel(planes, n_pixels, __float_as_uint(c.y), f3(b.z, b.w, c.x));
    if (COUNT) {
        atomicAdd(&ctl->h.nodes_sh, (unsigned long long)nn);
        atomicAdd(&ctl->h.tris_sh, (unsigned long long)(nt & 0xFFFFu));
        atomicAdd(&ctl->h.scanned, (unsigned long long)ns);
    }
}

// ---- k_tail ----------------------------------------------------------------------------------------------------------
// The end of a call: few paths are left and every wavefront iteration (plan, extend, shade, shadow, a host round trip)
// costs its fixed ~90 us whatever it carries, for up to max_depth more iterations.  Here one thread takes one node of
// the current queue and follows the path to its end: closest hit, shading, the occlusion query of its shadow ray, next
// bounce.  Second children (Microfacet: they start at depth >= max_depth - 5, so at most six are ever pending) wait on
// a small stack of the thread; should it overflow they go to the next queue, which a further launch of this kernel
// takes up.  Same random numbers, same arithmetic as the staged kernels; only the order of the atomic adds differs.
template <int SHADER, bool EXT = false>
__global__ __launch_bounds__(BLOCK) void k_tail(DevScene sc, RenderConst rc, Ctl *ctl, NodeQueue in, NodeQueue out, float *planes, uint32_t n_pixels,
                                                uint32_t seg_capacity, uint32_t heavy_blocks) {
    uint32_t seg, bx;
    decode_block(blockIdx.x, heavy_blocks, &seg, &bx);
    const uint32_t n_seg = min(ctl->seg_in[seg].v, seg_capacity);
    const uint32_t il = bx * BLOCK + threadIdx.x;
    if (il >= n_seg) return;
    const bool carry = sc.carry_out != 0;
    // every node this thread shades belongs to one pixel: what they gather is summed here and added once at the end
    Node nd = load_node(in, seg * seg_capacity + il);
    F3 light = sc.carry_in ? load_light(in, seg * seg_capacity + il, nd.meta) : f3(0.0f, 0.0f, 0.0f);
    const uint32_t pixel_index = tiled_index(rc, nd.meta & 0xFFFFFFu);
    const uint32_t oseg_l = (bx + seg) % RT_NSEG_HALF;
    uint32_t more_closest = 0, shadows = 0;
    Node pending[6];
    int n_pending = 0;
    for (;;) { // ends: a child is deeper than its parent and needs depth <= max_depth
        float t_obj;
        int code, prim;
        uint32_t nn = 0, nt = 0;
        // (the plain walk: with the top of the tree in LDS and the scan of the loose triangles this kernel was measured no faster)
        world_hit<false>(sc, nullptr, nd.o, nd.d, rc.t_min, rc.t_max, &t_obj, &code, &prim, &nn, &nt);
        ShadeResult r;
        shade_node<SHADER, EXT>(sc, rc, planes, n_pixels, nd, make_float2(t_obj, __int_as_float(code)), r);
        finish_children(rc, nd, r);
        if (r.ended) light = light + r.c0.w;
        if (r.have_s) {
            shadows++;
            if (!world_occluded<false>(sc, nullptr, r.s_o, r.s_d, rc.t_min, 1.0f - rc.t_min, &nn, &nt)) light = light + r.s_c;
        }
        if (r.have1) {
            if (n_pending < 6) pending[n_pending++] = r.c1;
            else {
                const uint32_t oseg = ray_segment(ray_is_heavy(sc, r.c1.o, r.c1.d, rc.t_min, rc.t_max), oseg_l);
                const uint32_t slot = atomicAdd(&ctl->seg_out[oseg].v, 1u);
                if (slot < seg_capacity) store_node(out, oseg * seg_capacity + slot, r.c1, carry); else ctl->h.overflow = 1; // (carries nothing: L = 0)
            }
        }
        if (r.have0) nd = r.c0;
        else if (n_pending > 0) nd = pending[--n_pending];
        else break;
        more_closest++;
    }
    if (!is_black(light)) add_pixel(planes, n_pixels, pixel_index, light);
    // the first ray of every path is counted by k_plan with the queue, the others here
    if (more_closest) atomicAdd(&ctl->h.rays_closest, (unsigned long long)more_closest);
    if (shadows) atomicAdd(&ctl->h.rays_shadow, (unsigned long long)shadows);
}

// One record per triangle for the shading kernel: the geometric normal of geometry.rs:59 and the material id, so that
// a shaded hit costs one 16-byte load instead of two and no cross product, square root and division.  Computed here,
// by the arithmetic the shading code used to run per hit, once per scene.
__global__ __launch_bounds__(BLOCK) void k_tri_normals(const float4 *tris, int n, float4 *out) {
    const int i = (int)(blockIdx.x * BLOCK + threadIdx.x);
    if (i >= n) return;
    const float4 b = tris[3 * (size_t)i + 1], c = tris[3 * (size_t)i + 2];
    const F3 nm = normalize(cross(f3(b.x, b.y, b.z), f3(c.x, c.y, c.z)));
    out[i] = make_float4(nm.x, nm.y, nm.z, b.w);
}

// ---- normal shader (shader.rs:7-13): no queues, one thread per pixel looping over its samples ---------------------------
__global__ __launch_bounds__(BLOCK) void k_normal_shader(DevScene sc, RenderConst rc, float *planes, uint32_t n_pixels) {
    const uint32_t pixel = blockIdx.x * BLOCK + threadIdx.x;
    if (pixel >= n_pixels) return;
    int x = (int)(pixel % (uint32_t)rc.width), y = (int)(pixel / (uint32_t)rc.width);
    if (!pixel_selected(rc, x, y)) return;
    uint32_t pk = pixel_key(rc.seed, pixel);
    F3 col = f3(0.0f, 0.0f, 0.0f);
    for (int k = rc.spp_begin; k < rc.spp_end; k++) {
        F3 o, d;
        if (!camera_sample(rc, x, y, k, pk, root_node_key((uint32_t)k), &o, &d)) continue;
        float t;
        int code, prim;
        uint32_t nn, nt;
        world_hit<false>(sc, nullptr, o, d, rc.t_min, rc.t_max, &t, &code, &prim, &nn, &nt);
        if (code != HIT_NONE) {
            F3 n;
            int mat;
            surface_at(sc, code, o + t * d, &n, &mat);
            col = col + 0.5f * f3(n.x + 1.0f, n.y + 1.0f, n.z + 1.0f);
        } else {
            col = col + f3(0.8f, 0.8f, 0.8f);
        }
    }
    const uint32_t ti = tiled_index(rc, pixel);
    planes[ti] += col.x;
    planes[rc.plane_stride + ti] += col.y;
    planes[2u * rc.plane_stride + ti] += col.z;
}

// ---- primary-hit AOV -----------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_primary_hits(DevScene sc, RenderConst rc, int32_t *prim_out, float *t_out) {
    const uint32_t pixel = blockIdx.x * BLOCK + threadIdx.x;
    if (pixel >= (uint32_t)(rc.width * rc.height)) return;
    int x = (int)(pixel % (uint32_t)rc.width), y = (int)(pixel / (uint32_t)rc.width);
    int32_t prim = -1;
    float t = 0.0f;
    if (pixel_selected(rc, x, y)) {
        F3 o, d;
        if (camera_sample(rc, x, y, rc.spp_begin, pixel_key(rc.seed, pixel), root_node_key((uint32_t)rc.spp_begin), &o, &d)) {
            int code, pr;
            uint32_t nn, nt;
            float th;
            world_hit<false>(sc, nullptr, o, d, rc.t_min, rc.t_max, &th, &code, &pr, &nn, &nt);
            if (code != HIT_NONE) { prim = pr; t = th; }
        }
    }
    prim_out[pixel] = prim;
    t_out[pixel] = t;
}

// ---- accumulation buffer plumbing -------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_planes_to_rgb(RenderConst rc, const float *planes, float *accum_rgb, uint32_t n_pixels) {
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n_pixels) return;
    const uint32_t ti = tiled_index(rc, i);
    accum_rgb[3u * i] += planes[ti];
    accum_rgb[3u * i + 1u] += planes[rc.plane_stride + ti];
    accum_rgb[3u * i + 2u] += planes[2u * rc.plane_stride + ti];
}

// main.rs:45-50: /NS, sqrt, (255.99 * c) as i32, then clamped to a byte
__global__ __launch_bounds__(BLOCK) void k_resolve(const float *accum_rgb, long long n, float spp, uint8_t *rgb8) {
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n) return;
    float c = sqrtf(accum_rgb[i] / spp);
    float v = 255.99f * c;
    int iv = (v != v) ? 0 : (v >= 2147483648.0f ? 2147483647 : (v <= -2147483648.0f ? (-2147483647 - 1) : (int)v));
    rgb8[i] = (uint8_t)(iv < 0 ? 0 : (iv > 255 ? 255 : iv));
}

// dst += src (logical shards of one frame rendered on the same GPU)
__global__ __launch_bounds__(BLOCK) void k_accum_add(float *dst, const float *src, long long n) {
    const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

// ---- launchers ------------------------------------------------------------------------------------------------------------------
static inline uint32_t nblocks(uint32_t n) { return (n + BLOCK - 1) / BLOCK; }
// dynamic LDS of the traversal kernels: the top of the tree (when cached) + one queue area per wave
static inline size_t lds_bytes(const DevScene &sc, bool use_lds) {
    size_t need = wave_area_offset(sc, use_lds) + (BLOCK / 64) * sizeof(WaveWalkLds);
    // experiment: a larger request lowers the number of resident workgroups per CU
    if (const char *v = std::getenv("RT_WALK_LDS_KB")) need = std::max(need, (size_t)std::atoi(v) * 1024u);
    return need;
}

void launch_plan(const LaunchCfg &cfg, Ctl *ctl, CtlHead *host_head) { hipLaunchKernelGGL(k_plan, dim3(1), dim3(64), 0, cfg.stream, ctl, cfg.target, host_head); }

void launch_raygen(const LaunchCfg &cfg, const DevScene &sc, const RenderConst &rc, Ctl *ctl, NodeQueue in, const int32_t *blocks, uint32_t blocks_w,
                   uint32_t n_threads) {
    if (!n_threads) return;
    hipLaunchKernelGGL(k_raygen, dim3((n_threads + RT_RAYGEN_PER_WG - 1) / RT_RAYGEN_PER_WG), dim3(BLOCK), 0, cfg.stream, sc, rc, ctl, in, blocks, blocks_w, cfg.seg_capacity);
}

static inline uint32_t class_grid(uint32_t light_threads, uint32_t heavy_threads, uint32_t *heavy_blocks, uint32_t heavy_span = 64) {
    const uint32_t per_block = (BLOCK / 64) * heavy_span; // heavy rays one workgroup takes
    *heavy_blocks = (heavy_threads + per_block - 1) / per_block;
    return RT_NSEG_HALF * (*heavy_blocks + nblocks(light_threads));
}

// Rays per wave of the heavy segments (walk_range): longer ranges keep the lanes busier, shorter ones give more waves.
// Enough waves to fill the chip several times over come first.
static inline uint32_t heavy_span_for(const DevScene &sc, uint32_t heavy_threads, bool count) {
    if (!sc.deferred || count) return 64u;
    const uint32_t waves_wanted = 256u * 4u * 8u; // 8 per SIMD
    uint32_t k = (uint32_t)(((unsigned long long)heavy_threads * RT_NSEG_HALF / 64u) / waves_wanted);
    const uint32_t k_max = (uint32_t)sc.walk_span;
    k = k < 1u ? 1u : (k > k_max ? k_max : k);
    return 64u * k;
}

void launch_extend(const LaunchCfg &cfg, const DevScene &sc, const RenderConst &rc, Ctl *ctl, NodeQueue in, float2 *hit, uint32_t light_threads,
                   uint32_t heavy_threads, bool count) {
    uint32_t hb;
    const uint32_t span = heavy_span_for(sc, heavy_threads, count);
    const uint32_t grid = class_grid(light_threads, heavy_threads, &hb, span);
    if (!grid) return;
    dim3 g(grid), b(BLOCK);
    if (count) hipLaunchKernelGGL((k_extend<true, false>), g, b, lds_bytes(sc, false), cfg.stream, sc, rc, ctl, in, hit, cfg.seg_capacity, hb, span);
    else if (sc.n_top > 0) hipLaunchKernelGGL((k_extend<false, true>), g, b, lds_bytes(sc, true), cfg.stream, sc, rc, ctl, in, hit, cfg.seg_capacity, hb, span);
    else hipLaunchKernelGGL((k_extend<false, false>), g, b, lds_bytes(sc, false), cfg.stream, sc, rc, ctl, in, hit, cfg.seg_capacity, hb, span);
}

void launch_shade(const LaunchCfg &cfg, int shader, const DevScene &sc, const RenderConst &rc, Ctl *ctl, NodeQueue in, NodeQueue out, const float2 *hit,
                  ShadowQueue sh, float *planes, uint32_t n_pixels, uint32_t light_threads, uint32_t heavy_threads) {
    uint32_t hb;
    const uint32_t grid = class_grid(light_threads, heavy_threads, &hb);
    if (!grid) return;
    dim3 g(grid), b(BLOCK);
    const bool tab = tables_fit(sc);
    const bool ext = (rc.flags & (RT_FLAG_RUSSIAN_ROULETTE | RT_FLAG_COSINE_DIFFUSE)) != 0u; // options the reference does not have: own instantiations
    if (ext && shader == RT_SHADER_MIS && tab)
        hipLaunchKernelGGL((k_shade<RT_SHADER_MIS, true, true>), g, b, 0, cfg.stream, sc, rc, ctl, in, out, hit, sh, planes, n_pixels, cfg.seg_capacity, cfg.shadow_seg_capacity, hb);
    else if (ext && shader == RT_SHADER_PATH && tab)
        hipLaunchKernelGGL((k_shade<RT_SHADER_PATH, true, true>), g, b, 0, cfg.stream, sc, rc, ctl, in, out, hit, sh, planes, n_pixels, cfg.seg_capacity, cfg.shadow_seg_capacity, hb);
    else if (ext && shader == RT_SHADER_MIS)
        hipLaunchKernelGGL((k_shade<RT_SHADER_MIS, false, true>), g, b, 0, cfg.stream, sc, rc, ctl, in, out, hit, sh, planes, n_pixels, cfg.seg_capacity, cfg.shadow_seg_capacity, hb);
    else if (ext && shader == RT_SHADER_PATH)
        hipLaunchKernelGGL((k_shade<RT_SHADER_PATH, false, true>), g, b, 0, cfg.stream, sc, rc, ctl, in, out, hit, sh, planes, n_pixels, cfg.seg_capacity, cfg.shadow_seg_capacity, hb);
    else if (shader == RT_SHADER_MIS && tab)
        hipLaunchKernelGGL((k_shade<RT_SHADER_MIS, true>), g, b, 0, cfg.stream, sc, rc, ctl, in, out, hit, sh, planes, n_pixels, cfg.seg_capacity, cfg.shadow_seg_capacity, hb);
    else if (shader == RT_SHADER_PATH && tab)
        hipLaunchKernelGGL((k_shade<RT_SHADER_PATH, true>), g, b, 0, cfg.stream, sc, rc, ctl, in, out, hit, sh, planes, n_pixels, cfg.seg_capacity, cfg.shadow_seg_capacity, hb);
    else if (tab)
        hipLaunchKernelGGL((k_shade<RT_SHADER_WHITTED, true>), g, b, 0, cfg.stream, sc, rc, ctl, in, out, hit, sh, planes, n_pixels, cfg.seg_capacity, cfg.shadow_seg_capacity, hb);
    else if (shader == RT_SHADER_MIS)
        hipLaunchKernelGGL((k_shade<RT_SHADER_MIS, false>), g, b, 0, cfg.stream, sc, rc, ctl, in, out, hit, sh, planes, n_pixels, cfg.seg_capacity, cfg.shadow_seg_capacity, hb);
    else if (shader == RT_SHADER_PATH)
        hipLaunchKernelGGL((k_shade<RT_SHADER_PATH, false>), g, b, 0, cfg.stream, sc, rc, ctl, in, out, hit, sh, planes, n_pixels, cfg.seg_capacity, cfg.shadow_seg_capacity, hb);
    else
        hipLaunchKernelGGL((k_shade<RT_SHADER_WHITTED, false>), g, b, 0, cfg.stream, sc, rc, ctl, in, out, hit, sh, planes, n_pixels, cfg.seg_capacity, cfg.shadow_seg_capacity, hb);
}

void launch_shadow(const LaunchCfg &cfg, const DevScene &sc, const RenderConst &rc, Ctl *ctl, ShadowQueue sh, float *planes, uint32_t n_pixels,
                   uint32_t light_threads, uint32_t heavy_threads, bool count) {
    uint32_t hb;
    const uint32_t span = heavy_span_for(sc, heavy_threads, count);
    const uint32_t grid = class_grid(light_threads, heavy_threads, &hb, span);
    if (!grid) return;
    dim3 g(grid), b(BLOCK);
    if (count) hipLaunchKernelGGL((k_shadow<true, false>), g, b, lds_bytes(sc, false), cfg.stream, sc, rc, ctl, sh, planes, n_pixels, cfg.shadow_seg_capacity, hb, span);
    else if (sc.n_top > 0) hipLaunchKernelGGL((k_shadow<false, true>), g, b, lds_bytes(sc, true), cfg.stream, sc, rc, ctl, sh, planes, n_pixels, cfg.shadow_seg_capacity, hb, span);
    else hipLaunchKernelGGL((k_shadow<false, false>), g, b, lds_bytes(sc, false), cfg.stream, sc, rc, ctl, sh, planes, n_pixels, cfg.shadow_seg_capacity, hb, span);
}

void launch_tail(const LaunchCfg &cfg, int shader, const DevScene &sc, const RenderConst &rc, Ctl *ctl, NodeQueue in, NodeQueue out, float *planes,
                 uint32_t n_pixels, uint32_t light_threads, uint32_t heavy_threads) {
    uint32_t hb;
    const uint32_t grid = class_grid(light_threads, heavy_threads, &hb);
    if (!grid) return;
    dim3 g(grid), b(BLOCK);
    const bool ext = (rc.flags & (RT_FLAG_RUSSIAN_ROULETTE | RT_FLAG_COSINE_DIFFUSE)) != 0u;
    if (ext && shader == RT_SHADER_MIS) hipLaunchKernelGGL((k_tail<RT_SHADER_MIS, true>), g, b, 0, cfg.stream, sc, rc, ctl, in, out, planes, n_pixels, cfg.seg_capacity, hb);
    else if (ext) hipLaunchKernelGGL((k_tail<RT_SHADER_PATH, true>), g, b, 0, cfg.stream, sc, rc, ctl, in, out, planes, n_pixels, cfg.seg_capacity, hb);
    else if (shader == RT_SHADER_MIS) hipLaunchKernelGGL(k_tail<RT_SHADER_MIS>, g, b, 0, cfg.stream, sc, rc, ctl, in, out, planes, n_pixels, cfg.seg_capacity, hb);
    else hipLaunchKernelGGL(k_tail<RT_SHADER_PATH>, g, b, 0, cfg.stream, sc, rc, ctl, in, out, planes, n_pixels, cfg.seg_capacity, hb);
}

void launch_normal_shader(const LaunchCfg &cfg, const DevScene &sc, const RenderConst &rc, float *planes, uint32_t n_pixels) {
    hipLaunchKernelGGL(k_normal_shader, dim3(nblocks(n_pixels)), dim3(BLOCK), 0, cfg.stream, sc, rc, planes, n_pixels);
}

void launch_primary_hits(const LaunchCfg &cfg, const DevScene &sc, const RenderConst &rc, int32_t *prim, float *t) {
    uint32_t n = (uint32_t)(rc.width * rc.height);
    hipLaunchKernelGGL(k_primary_hits, dim3(nblocks(n)), dim3(BLOCK), 0, cfg.stream, sc, rc, prim, t);
}

void launch_planes_to_rgb(const LaunchCfg &cfg, const RenderConst &rc, const float *planes, float *accum_rgb, uint32_t n_pixels) {
    hipLaunchKernelGGL(k_planes_to_rgb, dim3(nblocks(n_pixels)), dim3(BLOCK), 0, cfg.stream, rc, planes, accum_rgb, n_pixels);
}

void launch_tri_normals(hipStream_t stream, const float4 *tris, int n, float4 *out) {
    if (n > 0) hipLaunchKernelGGL(k_tri_normals, dim3(nblocks((uint32_t)n)), dim3(BLOCK), 0, stream, tris, n, out);
}

void launch_accum_add(hipStream_t stream, float *dst, const float *src, int64_t n) {
    if (n > 0) hipLaunchKernelGGL(k_accum_add, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream, dst, src, (long long)n);
}

void launch_resolve(hipStream_t stream, const float *accum_rgb, int64_t n_pixels, int32_t spp, uint8_t *rgb8) {
    long long n = (long long)n_pixels * 3;
    hipLaunchKernelGGL(k_resolve, dim3((unsigned)((n + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, stream, accum_rgb, n, (float)spp, rgb8);
}

} // namespace rtd

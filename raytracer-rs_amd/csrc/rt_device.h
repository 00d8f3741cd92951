// rt_device.h — device-side building blocks of the wavefront path tracer (gfx950).
//
// Everything that decides WHERE a path goes (hit tests, hit points, normals, sampled directions, random numbers) is
// written with a fixed operation order, IEEE division/sqrt and no FMA contraction (the file is compiled with
// -ffp-contract=off), so those values are bit-identical with a scalar CPU evaluation of the same formulas.
// Citations are paths under the reference root (raytracer-rs).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_api.h"

namespace rtd {

// ------------------------------------------------------------------------------------------------ vector maths
struct F3 {
    float x, y, z;
};
__device__ __forceinline__ F3 f3(float x, float y, float z) { return F3{x, y, z}; }
__device__ __forceinline__ F3 operator+(F3 a, F3 b) { return F3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ F3 operator-(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ F3 operator-(F3 a) { return F3{-a.x, -a.y, -a.z}; }
__device__ __forceinline__ F3 operator*(F3 a, float s) { return F3{a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ F3 operator*(float s, F3 a) { return F3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ F3 operator/(F3 a, float s) { return F3{a.x / s, a.y / s, a.z / s}; }
// cgmath 0.18: dot = (x*x' + y*y') + z*z'; normalize = v * (1/|v|); lerp = a + (b-a)*t
__device__ __forceinline__ float dot(F3 a, F3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ F3 cross(F3 a, F3 b) { return F3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ F3 normalize(F3 a) { return a * (1.0f / sqrtf(dot(a, a))); }
__device__ __forceinline__ F3 lerp(F3 a, F3 b, float t) { return a + (b - a) * t; }
__device__ __forceinline__ F3 mul_v(F3 a, F3 b) { return F3{a.x * b.x, a.y * b.y, a.z * b.z}; } // lib.rs:26-28
__device__ __forceinline__ bool is_black(F3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
__device__ __forceinline__ bool all_finite(F3 a) { return isfinite(a.x) && isfinite(a.y) && isfinite(a.z); }

// ---- approximate division (1 ulp reciprocal) --------------------------------------------------------------------------
// ONLY for the last scaling step of a quantity that weighs a path (BSDF value, pdf, light term) and for scheduling
// hints.  Anything that decides where a path goes (hit tests, hit points, normals, sampled directions) uses the IEEE
// forms above, so paths stay bit-identical with the CPU evaluation.  Unit vectors keep the IEEE normalisation as well:
// they feed cancelling expressions ((1 - |cos|)^2 in the Fresnel term, wi + wo in the half vector) that would amplify
// a 1-ulp error to 1e-3; a 1-ulp error in a final quotient stays 1 ulp.  The cosine of next-event estimation keeps the
// IEEE normalisations too: at grazing angles its three products cancel, and only bit-identical inputs keep the result
// of that cancellation identical (reciprocal square roots there: 4e-4 relative on surfaces a hair from a light).
__device__ __forceinline__ float rcp_w(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float div_w(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float sqrt_w(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ F3 div_w(F3 a, float s) { const float r = __builtin_amdgcn_rcpf(s); return F3{a.x * r, a.y * r, a.z * r}; }

#define RTD_PI 3.14159265358979323846f
#define RTD_ZERO 1.1920928955078125e-07f /* f32::EPSILON, lib.rs:19 */

__device__ __forceinline__ bool vec_eq(F3 a, F3 b, float t_min) { // lib.rs:31-34
    F3 d = a - b;
    return dot(d, d) < t_min;
}

// ------------------------------------------------------------------------------------------------ counter RNG
// value = f(seed, pixel, sample, path node, purpose, index): no state survives between draws.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352dU;
    x ^= x >> 15;
    x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t pixel_key(uint32_t seed, uint32_t pixel) { return mix32(mix32(seed ^ 0x9E3779B9U) + pixel); }
__device__ __forceinline__ uint32_t root_node_key(uint32_t sample) { return mix32(sample + 0x6A09E667U); }
__device__ __forceinline__ uint32_t child_node_key(uint32_t nk, uint32_t branch) { return mix32(nk + 0xBB67AE85U * (branch + 1U)); }

enum : uint32_t { P_CAMERA = 1, P_SCATTER = 2, P_DIFFUSE = 3, P_NEE = 4, P_ROULETTE = 5, P_ROULETTE1 = 6, P_VISIBLE0 = 16 };

struct Stream {
    uint32_t pk, sk, j;
    __device__ __forceinline__ Stream(uint32_t pk_, uint32_t nk, uint32_t purpose) : pk(pk_), sk(mix32(nk + 0x85EBCA6BU * purpose)), j(0) {}
    __device__ __forceinline__ uint32_t next_u32() {
        ++j;
        return mix32((sk + 0x9E3779B9U * j) ^ pk);
    }
    // gen::<f32>() of rand 0.8.5: 24 bits scaled into [0,1)
    __device__ __forceinline__ float next_f32() { return (float)(next_u32() >> 8) * (1.0f / 16777216.0f); }
    __device__ __forceinline__ uint32_t next_index(uint32_t n) { return __umulhi(next_u32(), n); }
};

// sin and cos of 2*pi*r, r in [0,1): exact quadrant reduction + fixed polynomials (plain mul/add, no libm call)
__device__ __forceinline__ void sincos_2pi(float r, float *s_out, float *c_out) {
    int q = (int)(r * 4.0f + 0.5f);
    float f = r - (float)q * 0.25f;
    float a = f * 6.28318530717958647692f;
    float a2 = a * a;
    float sp = -1.6666654611e-1f + a2 * (8.3321608736e-3f + a2 * -1.9515295891e-4f);
    float s = a + a * a2 * sp;
    float cp = 4.166664568298827e-2f + a2 * (-1.388731625493765e-3f + a2 * 2.443315711809948e-5f);
    float c = 1.0f - 0.5f * a2 + a2 * a2 * cp;
    switch (q & 3) {
    case 0: *s_out = s; *c_out = c; break;
    case 1: *s_out = c; *c_out = -s; break;
    case 2: *s_out = -s; *c_out = -c; break;
    default: *s_out = -c; *c_out = s; break;
    }
}

// ------------------------------------------------------------------------------------------------ scene on the device
struct DevLight {
    int32_t kind;
    float ox, oy, oz;
    float radius;
    float sx, sy, sz;
    int32_t tri_first, tri_count;
    int32_t sampler_kind;
    float area;       // PolygonLight::new, light.rs:252-269
    float nx, ny, nz; // normal of face 0
    float bmin[3], bmax[3]; // polygon light: padded box of its faces, a conservative gate in front of the exact tests
};

// BVH node, 32 B: a = (xmin, xmax, ymin, ymax), b = (zmin, zmax, link, miss).  link >= 0: index of the first child
// (inner node); link < 0: leaf holding triangles [first, first+count) with ~link = first*8 + (count-1).  miss = node to
// continue with when the box is missed or the subtree/leaf is done (-1 = traversal finished).  Nodes [0, n_top) are the
// part of the tree rays meet most often (largest boxes first, parents before children: the part each workgroup keeps in LDS); the rest follow in depth-first order.
// `nodes` holds n_orders copies of the array (1, or 8 = one per sign octant of the ray direction) that differ only in
// link and miss, i.e. in the order children are visited.
// Triangle, 48 B: (v0.xyz, prim_id), (e1.xyz, material), (e2.xyz, unused) with e1 = v1-v0, e2 = v2-v0.
struct DevScene {
    // the same tree in 16-byte nodes, one array per order, each in its own depth-first sequence (layout: rt_bvh.h); the
    // planes are quantised to 16 bits over the root box: plane = q_lo + q * q_scale
    const uint4 *cnodes;
    int32_t n_cnodes; // nodes per order in cnodes (the tree built on the GPU drops nodes there that it keeps in `nodes`)
    float q_lo[3], q_scale[3];
    const float4 *nodes;
    const float4 *tris;
    const float4 *tri_nrm;      // per triangle record: unit geometric normal, material id (bits) in w
    const float4 *sph;      // cx cy cz r
    const int2 *sph_info;   // material, prim id
    const float *cyl;       // 5 floats: cx cz ymin ymax r
    const int2 *cyl_info;
    const rt_material *mats;
    const DevLight *lights;
    const float *ltris;     // 9 floats per light triangle
    int32_t n_nodes, n_top, n_orders, n_tri, n_sph, n_cyl, n_light;
    int32_t n_mat, n_light_tri;
    // Padded boxes around the big meshes.  A ray that touches one is expected to walk deep into the BVH ("heavy"), the
    // others ("light") can only meet triangles outside the big meshes; the queues keep the two kinds apart so that the
    // lanes of a wave have similar trip counts.  The test (ray_is_heavy) is conservative like the BVH's box test.
    int32_t n_heavy;
    // The triangles outside the big meshes, in primitive-id order, when they are few (n_loose >= 0; -1 = not in use):
    // (v0.xyz, prim_id), (e1.xyz, material), (e2.xyz, index into tris), (the triangle's plane n.xyz, n.v0, scaled so that a
    // signed distance beyond +-1 is safely off the plane).  A light ray is tested against all of them in a
    // loop the whole wave runs together (the reference's own scan of the world list, hitable.rs:44-50) instead of walking the BVH.
    const float4 *loose;
    int32_t n_loose;
    // Padded boxes of the loose triangles, one per group of triangles with the same box (the two halves of a wall):
    // (min.xyz, the group's triangles as a bit mask), (max.xyz, 0).  A closest-hit ray tests the boxes first, every lane in
    // step, and runs the exact triangle test only for the triangles whose box it meets, each lane for its own.
    const float4 *loose_boxes;
    int32_t n_loose_boxes;
    // light shadow rays (they can only meet the loose triangles, spheres and cylinders) are answered by the shading
    // kernel itself instead of going through the shadow queue when set (and n_loose >= 0)
    int32_t inline_shadow;
    // heavy rays: exact triangle tests are queued per wave and run 64 at a time (walk_range) when set
    int32_t deferred;
    int32_t walk_span; // ... and a wave may own up to 64 * walk_span consecutive rays of a heavy segment
    int32_t walk_refill; // lanes that must be out of work before the wave retires their rays and hands out new ones
    int32_t walk_pairs;  // a step of that walk fetches a node and its successor together
    int32_t walk_batch, walk_batch_any; // queued leaves that trigger the exact tests: closest hit / occlusion query
    int32_t carry_in, carry_out; // the nodes read / the nodes written carry the light their path has gathered (NodeQueue::l)
    const int32_t *tri_of_prim; // primitive id -> index into tris (-1 for spheres and cylinders)
    float heavy_min[4][3];
    float heavy_max[4][3];
};

#define RT_MAX_HEAVY 4

__device__ __forceinline__ bool ray_is_heavy(const DevScene &sc, F3 o, F3 d, float t_min, float t_max) {
    bool heavy = false;
    if (sc.n_heavy == 0) return false;
    const float ix = rcp_w(d.x), iy = rcp_w(d.y), iz = rcp_w(d.z);
    for (int k = 0; k < sc.n_heavy; k++) {
        float x0 = (sc.heavy_min[k][0] - o.x) * ix, x1 = (sc.heavy_max[k][0] - o.x) * ix;
        float y0 = (sc.heavy_min[k][1] - o.y) * iy, y1 = (sc.heavy_max[k][1] - o.y) * iy;
        float z0 = (sc.heavy_min[k][2] - o.z) * iz, z1 = (sc.heavy_max[k][2] - o.z) * iz;
        float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), t_min));
        float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), t_max));
        heavy |= tn <= __builtin_fmaf(tf, 1.00001f, 1e-7f); // widened exit like the BVH's box test: never misses a mesh the ray can hit
    }
    return heavy;
}

struct RenderConst {
    rt_camera cam;
    int32_t width, height, spp, spp_begin, spp_end;
    int32_t tile_size, shard_rank, shard_count;
    int32_t pixel_sampler, light_samples, max_depth;
    int32_t spp_edge, ns2_edge; // floor(sqrt(spp)), floor(sqrt(light_samples)): UniformSampler::new, sampler.rs:104
    float t_min, t_max;
    uint32_t seed;
    uint32_t flags; // RT_FLAG_RUSSIAN_ROULETTE, RT_FLAG_COSINE_DIFFUSE
    // accumulation planes are stored in 8x8 pixel tiles (64 consecutive floats per tile): the lanes of a wave mostly
    // work on one tile of one sample, so a wave's atomic adds fall into two cache lines instead of eight
    uint32_t tiles_w;            // ceil(width / 8)
    uint32_t plane_stride;       // floats per colour plane = 64 * tiles_w * ceil(height / 8)
    unsigned long long w_magic;  // ceil(2^40 / width): y = (pixel * w_magic) >> 40 for pixel < 2^24
};

// linear pixel (y*W + x) -> index inside a tiled colour plane
__device__ __forceinline__ uint32_t tiled_index(const RenderConst &rc, uint32_t pixel) {
    const uint32_t y = (uint32_t)(((unsigned long long)pixel * rc.w_magic) >> 40);
    const uint32_t x = pixel - y * (uint32_t)rc.width;
    return (((y >> 3) * rc.tiles_w + (x >> 3)) << 6) | ((y & 7u) << 3) | (x & 7u);
}

// hit codes written by the extend stage
#define HIT_NONE (-1)
#define HIT_KIND_TRI 0
#define HIT_KIND_SPH 1
#define HIT_KIND_CYL_SIDE 2
#define HIT_KIND_CYL_TOP 3
#define HIT_KIND_CYL_BOTTOM 4
__device__ __forceinline__ int hit_code(int kind, int idx) { return (kind << 28) | idx; }
__device__ __forceinline__ int hit_kind(int code) { return code >> 28; }
__device__ __forceinline__ int hit_index(int code) { return code & 0x0FFFFFFF; }

// ------------------------------------------------------------------------------------------------ primitives
// geometry.rs:29-63 with E1, E2 precomputed (the same two subtractions, done once on the host)
__device__ __forceinline__ bool tri_hit(F3 v0, F3 E1, F3 E2, F3 o, F3 d, float t_min, float t_max, bool both_side, float *t_out) {
    F3 P = cross(d, E2);
    float det = dot(P, E1);
    if (!both_side && det < 0.0f) return false;
    if (det == 0.0f) return false;
    float inv_det = 1.0f / det;
    F3 T = o - v0;
    F3 Q = cross(T, E1);
    float u = inv_det * dot(P, T);
    if (u < 0.0f || u > 1.0f) return false;
    float v = inv_det * dot(Q, d);
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = inv_det * dot(Q, E2);
    if (t < t_min || t > t_max) return false;
    *t_out = t;
    return true;
}

// geometry.rs:82-117
__device__ __forceinline__ bool sph_hit(float4 s, F3 o, F3 d, float t_min, float t_max, float *t_out) {
    F3 oc = o - f3(s.x, s.y, s.z);
    float a = dot(d, d);
    float b = dot(oc, d);
    float c = dot(oc, oc) - s.w * s.w;
    float disc = b * b - a * c;
    if (disc > t_min) {
        float temp = (-b - sqrtf(b * b - a * c)) / a;
        if (temp < t_max && temp > t_min) { *t_out = temp; return true; }
        temp = (-b + sqrtf(b * b - a * c)) / a;
        if (temp < t_max && temp > t_min) { *t_out = temp; return true; }
    }
    return false;
}

// geometry.rs:128-193; *kind receives HIT_KIND_CYL_*
__device__ __forceinline__ bool cyl_hit(const float *cy, F3 o, F3 d, float t_min, float t_max, float *t_out, int *kind) {
    const float cx = cy[0], cz = cy[1], ymin = cy[2], ymax = cy[3], r = cy[4];
    float t_top = (ymax - o.y) / d.y;
    float t_bottom = (ymin - o.y) / d.y;
    F3 p_top = o + t_top * d;
    float dtx = p_top.x - cx, dtz = p_top.z - cz;
    bool hit_top = t_top > t_min && t_top < t_max && (dtx * dtx + dtz * dtz) < r * r;
    F3 p_bottom = o + t_bottom * d;
    float dbx = p_bottom.x - cx, dbz = p_bottom.z - cz;
    bool hit_bottom = t_bottom > t_min && t_bottom < t_max && (dbx * dbx + dbz * dbz) < r * r;
    float dx = d.x, dz = d.z;
    float ox = o.x - cx, oz = o.z - cz;
    float a = dx * dx + dz * dz;
    float b = dx * ox + dz * oz;
    float c = (ox * ox + oz * oz) - r * r;
    float disc = b * b - a * c;
    if (disc > t_min) {
        float temp = (-b - sqrtf(disc)) / a;
        if (temp < t_max && temp > t_min) {
            F3 p = o + temp * d;
            if (p.y > ymin && p.y < ymax) { *t_out = temp; *kind = HIT_KIND_CYL_SIDE; return true; }
        }
        if (hit_top && t_top < t_bottom) { *t_out = t_top; *kind = HIT_KIND_CYL_TOP; return true; }
        else if (hit_bottom) { *t_out = t_bottom; *kind = HIT_KIND_CYL_BOTTOM; return true; }
        temp = (-b + sqrtf(disc)) / a;
        if (temp < t_max && temp > t_min) {
            F3 p = o + temp * d;
            if (p.y > ymin && p.y < ymax) { *t_out = temp; *kind = HIT_KIND_CYL_SIDE; return true; }
        }
    }
    return false;
}

// ------------------------------------------------------------------------------------------------ BVH traversal
// Stackless walk over the threaded node array.  The box test only culls (boxes are padded at build time and the exit
// distance is widened), so it may use reciprocals; the triangle test is the exact one above.  Closest hit keeps, among
// equal t, the triangle with the larger primitive id — the result of the reference's in-order scan, where a later
// triangle replaces an earlier one at equal t (hitable.rs:44-50 with geometry.rs:53).
struct TriHit {
    float t;
    int tri;  // index into DevScene::tris (-1 = none)
    int prim; // primitive id
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 16-byte fetches through buffer descriptors: one instruction per node half / triangle third that the compiler can
// neither split nor sink below the box test (a plain float4 load had its .w lane re-fetched after the test: one more
// dependent memory round trip per visited node), 32-bit offsets, and out-of-range reads return zero instead of faulting.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 load16(__amdgpu_buffer_rsrc_t r, uint32_t byte_offset) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_offset, 0, 0);
}
__device__ __forceinline__ float u2f(unsigned int v) { return __uint_as_float(v); }

typedef float f32x2 __attribute__((ext_vector_type(2)));

// Top-of-tree cache in LDS.  The boxes are kept once, as order 0 stores them, i.e. as (min, max) pairs — the arrays of the
// other orders hold an axis' pair as (max, min) where the octant's direction is negative, so a walk through this cache
// must use the order-independent min / max box test (bvh_traverse does), never the presorted one of walk_range.  Only the
// (link, miss) pairs are kept per order: [n_top x float4 (xmin,xmax,ymin,ymax)][n_top x float2 (zmin,zmax)]
// [n_orders x n_top x int2 (link, miss)] = 24 + 8 * n_orders bytes per node.
__host__ __device__ __forceinline__ uint32_t top_cache_bytes(const DevScene &sc) { return (uint32_t)sc.n_top * (24u + 8u * (uint32_t)sc.n_orders); }
// ... followed, 16-byte aligned, by one WaveWalkLds per wave of the workgroup
__host__ __device__ __forceinline__ uint32_t wave_area_offset(const DevScene &sc, bool use_lds) { return use_lds ? ((top_cache_bytes(sc) + 15u) & ~15u) : 0u; }
// every thread of the workgroup must call it
template <int BLOCK_THREADS>
__device__ __forceinline__ void fill_top_cache(const DevScene &sc, float4 *lds_nodes) {
    float2 *box_b = reinterpret_cast<float2 *>(lds_nodes + sc.n_top);
    float2 *links = box_b + sc.n_top;
    for (int j = threadIdx.x; j < sc.n_top; j += BLOCK_THREADS) {
        const float4 b = sc.nodes[2 * j + 1];
        lds_nodes[j] = sc.nodes[2 * j];
        box_b[j] = make_float2(b.x, b.y);
    }
    for (int j = threadIdx.x; j < sc.n_top * sc.n_orders; j += BLOCK_THREADS) {
        const int k = j / sc.n_top, node = j - k * sc.n_top;
        const float4 b = sc.nodes[2u * ((size_t)k * (size_t)sc.n_nodes + (size_t)node) + 1u];
        links[j] = make_float2(b.z, b.w);
    }
    __syncthreads();
}

// One box test per visited node, 3 packed multiply-adds for the six plane distances: t = plane * (1/d) - o * (1/d).
// A component of d that is zero (or so small that the products overflow) turns that axis' distances into NaN or
// infinities; min/max drop NaN operands, so such an axis never culls.
//
// POSTPONE = false: a lane that reaches a leaf tests its triangles at once (the other lanes of the wave wait).
// POSTPONE = true : a lane that reaches a leaf waits until no lane of the wave has an inner node left, then all lanes
// holding a leaf run the triangle tests together.  Pays for rays that only meet the few big triangles around them (a
// wave of those would otherwise run the test once per leaf ANY lane enters); loses for long walks and occlusion queries.
template <bool ANY_HIT, bool COUNT, bool USE_LDS = false, bool POSTPONE = false>
__device__ __forceinline__ TriHit bvh_traverse(const DevScene &sc, const float4 *lds_nodes, F3 o, F3 d, float t_min, float t_max,
                                               uint32_t *n_nodes, uint32_t *n_tris) {
    TriHit best{t_max, -1, -1};
    if (sc.n_nodes == 0) return best;
    const __amdgpu_buffer_rsrc_t node_r = make_rsrc(sc.nodes, (uint32_t)sc.n_nodes * (uint32_t)sc.n_orders * 32u);
    const __amdgpu_buffer_rsrc_t tri_r = make_rsrc(sc.tris, (uint32_t)sc.n_tri * 48u);
    const float ix = rcp_w(d.x), iy = rcp_w(d.y), iz = rcp_w(d.z); // box tests only cull (padded boxes, widened exit)
    const f32x2 ix2 = {ix, ix}, iy2 = {iy, iy}, iz2 = {iz, iz};
    const f32x2 ox2 = {-o.x * ix, -o.x * ix}, oy2 = {-o.y * iy, -o.y * iy}, oz2 = {-o.z * iz, -o.z * iz};
    // the copy of the node array whose visiting order suits this ray's direction octant
    const uint32_t octant = sc.n_orders == 8 ? ((d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u)) : 0u;
    const uint32_t g_base = octant * (uint32_t)sc.n_nodes * 32u;
    const float2 *l_box_b = reinterpret_cast<const float2 *>(lds_nodes + sc.n_top);
    const float2 *l_links = l_box_b + sc.n_top + octant * (uint32_t)sc.n_top;

    // box test of `node`; returns the node to continue with and, through *leaf, the packed triangle range of a leaf that was hit (else -1)
    auto step = [&](int node, int *leaf) -> int {
        u32x4 na, nb;
        if (USE_LDS && node < sc.n_top) {
            const float4 fa = lds_nodes[node];
            const float2 fb = l_box_b[node], fl = l_links[node];
            na = u32x4{__float_as_uint(fa.x), __float_as_uint(fa.y), __float_as_uint(fa.z), __float_as_uint(fa.w)};
            nb = u32x4{__float_as_uint(fb.x), __float_as_uint(fb.y), __float_as_uint(fl.x), __float_as_uint(fl.y)};
        } else {
            na = load16(node_r, g_base + (uint32_t)node * 32u);
            nb = load16(node_r, g_base + (uint32_t)node * 32u + 16u);
        }
        if (COUNT) (*n_nodes)++;
        const f32x2 tx = __builtin_elementwise_fma(f32x2{u2f(na.x), u2f(na.y)}, ix2, ox2);
        const f32x2 ty = __builtin_elementwise_fma(f32x2{u2f(na.z), u2f(na.w)}, iy2, oy2);
        const f32x2 tz = __builtin_elementwise_fma(f32x2{u2f(nb.x), u2f(nb.y)}, iz2, oz2);
        const float tn = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fmaxf(fminf(tz.x, tz.y), t_min));
        const float tf = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fminf(fmaxf(tz.x, tz.y), best.t));
        const bool inside = tn <= __builtin_fmaf(tf, 1.00001f, 1e-7f);
        const int link = (int)nb.z, miss = (int)nb.w;
        *leaf = (inside && link < 0) ? ~link : -1;
        return (inside && link >= 0) ? link : miss;
    };
    // exact tests of a leaf's triangles; true = an occlusion query is decided
    auto test_leaf = [&](int packed) -> bool {
        const int first = packed >> 3, count = (packed & 7) + 1;
        for (int k = 0; k < count; k++) {
            const uint32_t off = (uint32_t)(first + k) * 48u;
            const u32x4 a = load16(tri_r, off), b = load16(tri_r, off + 16u), c = load16(tri_r, off + 32u);
            if (COUNT) {
                (*n_tris)++;
                // bit 31..16 of the counter: triangle-loop iterations the WAVE runs (counted by its first active lane)
                if (__ffsll(__ballot(1)) - 1 == (int)(threadIdx.x & 63u)) (*n_tris) += 0x10000u;
            }
            float t;
            if (tri_hit(f3(u2f(a.x), u2f(a.y), u2f(a.z)), f3(u2f(b.x), u2f(b.y), u2f(b.z)), f3(u2f(c.x), u2f(c.y), u2f(c.z)), o, d, t_min, t_max, false, &t)) {
                const int prim = (int)a.w;
                // equal t: the later primitive of the world list wins (hitable.rs:44-50 + geometry.rs:53)
                if (ANY_HIT || t < best.t || (t == best.t && prim > best.prim)) {
                    best.t = t; best.tri = first + k; best.prim = prim;
                }
                if (ANY_HIT) return true;
            }
        }
        return false;
    };

    int node = 0;
    if (POSTPONE) {
        for (;;) {
            int leaf = -1;
            while (node >= 0 && leaf < 0) node = step(node, &leaf);
            if (leaf < 0) break;
            if (test_leaf(leaf)) break;
        }
    } else {
        while (node >= 0) {
            int leaf;
            node = step(node, &leaf);
            if (leaf >= 0 && test_leaf(leaf)) break;
        }
    }
    return best;
}

// ---- wave-cooperative walk with deferred triangle tests ---------------------------------------------------------------
// In the walk above a wave runs the exact triangle test (about 50 vector instructions with its IEEE division) every time
// ANY of its lanes enters a leaf, for the two or three lanes that stand at one: on a mesh-dominated frame three quarters
// of a wave's instructions are such iterations at 5 % lane utilisation.  Here a lane that enters a leaf only appends
// (leaf, lane) to a queue of its wave in LDS and walks on; when 64 entries have gathered (or no lane has a node left)
// the 64 lanes take one entry each and test its triangles for the ray of the entry's owner, read from LDS, so the test
// runs with every lane busy.  A hit is merged into the owner's best with one 64-bit LDS minimum on the key
// (t, 0x7FFFFFFF - primitive id): smaller t wins, equal t -> the larger primitive id (hitable.rs:44-50 + geometry.rs:53).
// The owner culls with the best t it read last, which may be stale by a few steps: it then visits boxes it could have
// skipped, never fewer, so the result is the same closest hit.  No workgroup barrier is involved: a wave only talks
// to itself, and LDS operations of one wave complete in order.
struct WaveWalkLds {
    float4 ray_a[64];             // o.xyz, d.x
    float2 ray_b[64];             // d.y, d.z
    unsigned long long best[64];  // closest hit: key of the best hit so far; occlusion query: 0 once a hit is known
    static constexpr uint32_t RING = 256u;
    uint2 q[RING];                // ring of (packed leaf, owner lane): a step adds at most 128 entries to at most 63
};

__device__ __forceinline__ uint32_t float_order_key(float t) { // monotonic float -> uint
    const uint32_t b = __float_as_uint(t);
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float float_from_order_key(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u);
}
// wave-uniform values kept in scalar registers: tells the compiler what the control flow of the walk relies on
// (integers only: a condition computed from such integers is uniform by itself, and forcing it through a lane read
// costs a v_cndmask + v_readfirstlane pair per use)
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// max / min of values that are never signalling NaNs (results of arithmetic): one instruction each, without the
// canonicalising v_max x, x the compiler puts in front of fmaxf / fminf operands.  A quiet NaN operand is dropped.
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float min3_raw(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float min_raw(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// A wave owns a RANGE of rays [first, first + total), usually more than 64: a lane whose ray is finished (walk done
// and none of its leaves left in the queue) hands its result to the client and takes the next ray of the range, so the
// lanes of a wave stay busy although their walks differ in length by an order of magnitude.  Lanes are retired and
// refilled sc.walk_refill at a time (retiring and starting a ray costs the whole wave about 150 instructions) or when
// nothing is left to do.
//
// Client (one per kernel):  bool fetch(uint32_t k, F3 *o, F3 *d, TriHit *init)  loads ray k; false = already decided, no walk.
//                               *init (preset to "none") may receive a hit found outside the tree (the loose triangles):
//                               the walk starts with it as its best hit, so boxes behind it are never entered
//                           void done(uint32_t k, F3 o, F3 d, const TriHit &h)   closest hit: h.prim (-1 = none), h.t;
//                                                                                 occlusion query: h.tri >= 0 = occluded
// Every lane of the wave must call walk_range; `wl` is the calling wave's own area.
template <bool ANY_HIT, bool COUNT, class Client>
__device__ __forceinline__ void walk_range(const DevScene &sc, WaveWalkLds &wl, Client &client, uint32_t first, uint32_t total,
                                           float t_min, float t_max, uint32_t *n_nodes, uint32_t *n_tris) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t NONE = 0xFFFFFFFFu;
    const __amdgpu_buffer_rsrc_t node_r = make_rsrc(sc.cnodes, (uint32_t)sc.n_cnodes * (uint32_t)sc.n_orders * 16u);
    const __amdgpu_buffer_rsrc_t tri_r = make_rsrc(sc.tris, (uint32_t)sc.n_tri * 48u);
    const unsigned long long key0 = ANY_HIT ? 1ull : (((unsigned long long)float_order_key(t_max) << 32) | 0x80000000ull); // (t_max, prim -1)
    const bool presorted = sc.n_orders == 8;
    const float t_min_v = t_min;
    const bool pairs = sc.walk_pairs != 0;

    // per-ray state of the lane
    uint32_t my_k = NONE;
    f32x2 ix2 = {0.0f, 0.0f}, iy2 = ix2, iz2 = ix2, ox2 = ix2, oy2 = ix2, oz2 = ix2;
    uint32_t g_base = 0;
    float cull_t = t_max;
    int node = -1;
    bool pushed = false;
    uint32_t last_pos = 0;

    uint32_t next = 0;            // rays of the range handed out so far (wave-uniform)
    uint32_t head = 0, count = 0; // queue: absolute position of its first entry, entries (wave-uniform)
    const uint32_t refill = uni((uint32_t)sc.walk_refill);
    const uint32_t batch = uni((uint32_t)(ANY_HIT ? sc.walk_batch_any : sc.walk_batch)); // queued leaves that trigger their tests (<= 64)
    first = uni(first);
    total = uni(total);
    for (;;) {
        // Lanes without work are looked after when enough of them have gathered: their queued leaves are tested at once
        // (even fewer than 64), which makes their results final, and they take new rays.
        const uint32_t n_walk = uni((uint32_t)__popcll(__ballot(node >= 0)));
        const bool starving = next < total && n_walk + refill <= 64u;
        if (n_walk != 0u && count < batch && !starving) {
            // One step of the walk for every lane that has a node; leaves are queued, not tested.  The walk waits for one
            // memory round trip per step, and in depth-first order more than half of all steps go on with the very next
            // node (the first child of a box that was hit, the successor of a leaf): so a step fetches the node AND its
            // successor together, and takes the successor in the same step when that is where the walk goes.
            auto test_node = [&](const u32x4 nd, int at, int *leaf) -> int { // box test of node `at`; returns the node to go on with
                if (COUNT) (*n_nodes)++;
                const f32x2 tx = __builtin_elementwise_fma(f32x2{(float)(nd.x & 0xFFFFu), (float)(nd.x >> 16)}, ix2, ox2);
                const f32x2 ty = __builtin_elementwise_fma(f32x2{(float)(nd.y & 0xFFFFu), (float)(nd.y >> 16)}, iy2, oy2);
                const f32x2 tz = __builtin_elementwise_fma(f32x2{(float)(nd.z & 0xFFFFu), (float)(nd.z >> 16)}, iz2, oz2);
                float tn, tf;
                if (presorted) { // the node array of the ray's octant holds (entry plane, exit plane) per axis
                    tn = max3_raw(tx.x, ty.x, max_raw(tz.x, t_min_v));
                    tf = min3_raw(tx.y, ty.y, min_raw(tz.y, cull_t));
                } else {
                    tn = fmaxf(fmaxf(fminf(tx.x, tx.y), fminf(ty.x, ty.y)), fmaxf(fminf(tz.x, tz.y), t_min));
                    tf = fminf(fminf(fmaxf(tx.x, tx.y), fmaxf(ty.x, ty.y)), fminf(fmaxf(tz.x, tz.y), cull_t));
                }
                const bool inside = tn <= __builtin_fmaf(tf, 1.00001f, 1e-7f);
                const bool is_leaf = (int)nd.w < 0;
                *leaf = (inside && is_leaf) ? (int)(nd.w & 0x7FFFFFFFu) : -1;
                // depth-first order: the first child of an inner node and the successor of a leaf are the next node
                const uint32_t nxt = (inside || is_leaf) ? (uint32_t)at + 1u : nd.w;
                return nxt < (uint32_t)sc.n_cnodes ? (int)nxt : -1;
            };
            auto push_leaves = [&](int leaf) { // every lane of the wave calls it
                const unsigned long long m = __ballot(leaf >= 0);
                if (m != 0ull) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    if (leaf >= 0) {
                        last_pos = head + count + rank;
                        pushed = true;
                        wl.q[last_pos & (WaveWalkLds::RING - 1u)] = make_uint2((uint32_t)leaf, lane);
                    }
                    count = uni(count + (uint32_t)__popcll(m));
                }
            };
            int leaf = -1, leaf1 = -1;
            if (node >= 0) {
                const int at = node;
                const u32x4 nd = load16(node_r, g_base + (uint32_t)at * 16u);
                u32x4 nd1 = u32x4{0u, 0u, 0u, 0u};
                if (pairs) nd1 = load16(node_r, g_base + (uint32_t)at * 16u + 16u); // (past the end of the array: zeros, never used)
                node = test_node(nd, at, &leaf);
                if (pairs && node == at + 1) node = test_node(nd1, at + 1, &leaf1);
            }
            push_leaves(leaf);
            if (pairs) push_leaves(leaf1); // at most 63 + 64 + 64 entries: the ring holds 256
            continue;
        }
        if (count > 0u) {
        // the exact tests of up to 64 queued leaves, one per lane
        wave_lds_fence();
        const uint32_t n = uni(min(count, 64u));
        if (lane < n) {
            const uint2 e = wl.q[(head + lane) & (WaveWalkLds::RING - 1u)];
            const uint32_t owner = e.y;
            const unsigned long long seen = wl.best[owner];
            if (!ANY_HIT || seen != 0ull) {
                const float4 ra = wl.ray_a[owner];
                const float2 rb = wl.ray_b[owner];
                const F3 ro = f3(ra.x, ra.y, ra.z), rd = f3(ra.w, rb.x, rb.y);
                float bound = ANY_HIT ? t_max : float_from_order_key((uint32_t)(seen >> 32));
                const int tri0 = (int)(e.x >> 3), cnt = (int)(e.x & 7u) + 1;
                for (int k = 0; k < cnt; k++) {
                    const uint32_t off = (uint32_t)(tri0 + k) * 48u;
                    const u32x4 a = load16(tri_r, off), b = load16(tri_r, off + 16u), c = load16(tri_r, off + 32u);
                    if (COUNT) (*n_tris)++;
                    float t;
                    if (tri_hit(f3(u2f(a.x), u2f(a.y), u2f(a.z)), f3(u2f(b.x), u2f(b.y), u2f(b.z)), f3(u2f(c.x), u2f(c.y), u2f(c.z)), ro, rd, t_min, t_max, false, &t)) {
                        if (ANY_HIT) {
                            wl.best[owner] = 0ull;
                            break;
                        }
                        if (t <= bound) { // never true for a NaN t
                            bound = t;
                            const unsigned long long key = ((unsigned long long)float_order_key(t) << 32) | (unsigned long long)(uint32_t)(0x7FFFFFFF - (int)a.w);
                            atomicMin(&wl.best[owner], key);
                        }
                    }
                }
            }
        }
            head = uni(head + n);
            count = uni(count - n);
            wave_lds_fence();
            const unsigned long long mine = wl.best[lane];
            if (ANY_HIT) {
                if (mine == 0ull) node = -1;
            } else {
                cull_t = float_from_order_key((uint32_t)(mine >> 32));
            }
        }
        // Retire and refill.  Looked at only here — after a batch of tests, or when nothing is left to do — not at every
        // step: a lane's result becomes final when its walk is over and the queue has moved past its last leaf.
        const bool fin = my_k != NONE && node < 0 && (!pushed || (int32_t)(head - last_pos) > 0);
        const uint32_t n_fin = uni((uint32_t)__popcll(__ballot(fin))), n_idle = uni((uint32_t)__popcll(__ballot(my_k == NONE)));
        const bool drained = count == 0u && __ballot(node >= 0) == 0ull;
        if (drained || (next < total && n_fin + n_idle >= refill)) {
            if (fin) {
                const unsigned long long r = wl.best[lane];
                const float4 ra = wl.ray_a[lane];
                const float2 rb = wl.ray_b[lane];
                TriHit h{t_max, -1, -1};
                if (ANY_HIT) {
                    if (r == 0ull) h.tri = 0;
                } else if (r != key0) {
                    h.t = float_from_order_key((uint32_t)(r >> 32));
                    h.prim = 0x7FFFFFFF - (int)(uint32_t)r;
                }
                client.done(my_k, f3(ra.x, ra.y, ra.z), f3(ra.w, rb.x, rb.y), h);
                my_k = NONE;
            }
            const unsigned long long free_m = __ballot(my_k == NONE);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(free_m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)free_m, 0u));
            const uint32_t remaining = uni(total - next);
            if (my_k == NONE && rank < remaining) {
                my_k = first + next + rank;
                F3 o, d;
                TriHit init{t_max, -1, -1};
                const bool walk = client.fetch(my_k, &o, &d, &init);
                const unsigned long long k_init = (ANY_HIT || init.prim < 0) ? key0
                    : (((unsigned long long)float_order_key(init.t) << 32) | (unsigned long long)(uint32_t)(0x7FFFFFFF - init.prim));
                // box tests only cull (padded boxes, quantised outwards, widened exit): distance to the plane of grid
                // coordinate q along x = q * (q_scale.x / d.x) + (q_lo.x - o.x) / d.x, one packed multiply-add per axis
                const float ix = rcp_w(d.x), iy = rcp_w(d.y), iz = rcp_w(d.z);
                const float ax = sc.q_scale[0] * ix, ay = sc.q_scale[1] * iy, az = sc.q_scale[2] * iz;
                const float bx = (sc.q_lo[0] - o.x) * ix, by = (sc.q_lo[1] - o.y) * iy, bz = (sc.q_lo[2] - o.z) * iz;
                ix2 = f32x2{ax, ax}; iy2 = f32x2{ay, ay}; iz2 = f32x2{az, az};
                ox2 = f32x2{bx, bx}; oy2 = f32x2{by, by}; oz2 = f32x2{bz, bz};
                // the octant is taken from the SIGN BITS (a component of -0 has a reciprocal of -inf): it decides which of a
                // node's two planes per axis is the entry plane, and must agree with the sign of the reciprocal
                const uint32_t octant = presorted ? ((__float_as_uint(d.x) >> 31) | ((__float_as_uint(d.y) >> 31) << 1) | ((__float_as_uint(d.z) >> 31) << 2)) : 0u;
                g_base = octant * (uint32_t)sc.n_cnodes * 16u;
                wl.ray_a[lane] = make_float4(o.x, o.y, o.z, d.x);
                wl.ray_b[lane] = make_float2(d.y, d.z);
                wl.best[lane] = (ANY_HIT && !walk) ? 0ull : k_init;
                cull_t = ANY_HIT ? t_max : init.t;
                node = (walk && sc.n_cnodes > 0) ? 0 : -1;
                pushed = false;
            }
            next = uni(next + min((uint32_t)__popcll(free_m), remaining));
            wave_lds_fence();
            if (drained && __ballot(my_k != NONE) == 0ull) break; // nothing in flight, nothing left to hand out
        }
    }
}

// The loose triangles, one after the other, for a ray that misses the boxes of all big meshes.  The loop is the same for
// every lane of the wave (no divergent walk) and the triangle data is uniform, so the loads are scalar.
// `staged`: the workgroup's LDS copy of the 3 * n_loose record words (closest hit only), or nullptr.
template <bool ANY_HIT, bool COUNT>
__device__ __forceinline__ TriHit loose_traverse(const DevScene &sc, F3 o, F3 d, float t_min, float t_max, uint32_t *n_scan, const float4 *staged = nullptr) {
    TriHit best{t_max, -1, -1};
    auto test = [&](int k) { // exact test of loose triangle k; true = an occlusion query is answered
        const float4 a = sc.loose[3 * k], b = sc.loose[3 * k + 1], c = sc.loose[3 * k + 2];
        if (COUNT) (*n_scan)++;
        float t;
        if (tri_hit(f3(a.x, a.y, a.z), f3(b.x, b.y, b.z), f3(c.x, c.y, c.z), o, d, t_min, t_max, false, &t)) {
            const int prim = __float_as_int(a.w);
            if (ANY_HIT || t < best.t || (t == best.t && prim > best.prim)) {
                best.t = t; best.tri = __float_as_int(c.w); best.prim = prim;
            }
            return ANY_HIT;
        }
        return false;
    };
    // An occlusion query over the segment t in [t_min, t_max] (within [0, 1]: bounded) can only hit a FRONT face whose plane
    // the segment crosses from front to back: with the start behind the plane, or the end in front of it, by the margin the
    // host scaled the plane with (2e-5 of the scene's size, more for slivers: far above the rounding of the exact test,
    // so the exact test would reject as well), the triangle is skipped — and when that holds for every lane of the wave
    // the exact test (50 instructions with its division) is not run at all.  In a closed room that is almost every
    // triangle for almost every wave.  NaN coordinates compare false: never skipped.  The planes lie behind the triangle
    // records, padded to a multiple of four with planes nothing is in front of: four planes come with one scalar load,
    // so a wave waits for memory once per four triangles instead of once per triangle.
    if (ANY_HIT && t_min >= 0.0f && t_max <= 1.0f) {
        const float4 *planes = sc.loose + 3 * sc.n_loose;
        bool found = false;
        for (int k0 = 0; k0 < sc.n_loose; k0 += 4) {
            const float4 p0 = planes[k0], p1 = planes[k0 + 1], p2 = planes[k0 + 2], p3 = planes[k0 + 3];
            auto crosses = [&](const float4 &pl) {
                // signed distances (in margins) of the segment's two ends, o + t_min d and o + t_max d: the light of the
                // reference's room lies IN the ceiling's plane, and only the 1 - T_MIN of the far end keeps it in front
                const float da = __builtin_fmaf(pl.x, o.x, __builtin_fmaf(pl.y, o.y, __builtin_fmaf(pl.z, o.z, -pl.w)));
                const float dn = __builtin_fmaf(pl.x, d.x, __builtin_fmaf(pl.y, d.y, pl.z * d.z));
                return !found && !(__builtin_fmaf(t_min, dn, da) <= -1.0f || __builtin_fmaf(t_max, dn, da) >= 1.0f);
            };
            const bool c0 = crosses(p0), c1 = crosses(p1), c2 = crosses(p2), c3 = crosses(p3);
            if (__ballot(c0) != 0ull && !found) found = test(k0);
            if (__ballot(c1) != 0ull && !found) found = test(k0 + 1);
            if (__ballot(c2) != 0ull && !found) found = test(k0 + 2);
            if (__ballot(c3) != 0ull && !found) found = test(k0 + 3);
            if (__ballot(!found) == 0ull) break;
        }
        return best;
    }
    if (!ANY_HIT && sc.n_loose_boxes > 0) {
        // Closest hit.  A ray in a room meets one or two of the triangles around it, and a wave running the exact test
        // (50 instructions with its IEEE division) for every triangle in turn spends most of them on misses.  The boxes
        // only cull (padded like the leaves of the tree, exit distance widened, NaN distances of a zero direction
        // component dropped by min / max), so the result is the exact tests' own.
        const float ix = rcp_w(d.x), iy = rcp_w(d.y), iz = rcp_w(d.z);
        const float ox = -o.x * ix, oy = -o.y * iy, oz = -o.z * iz;
        uint32_t cand = 0u;
        for (int g = 0; g < sc.n_loose_boxes; g++) {
            const float4 lo = sc.loose_boxes[2 * g], hi = sc.loose_boxes[2 * g + 1]; // (scalar loads: g is the same for every lane)
            const float x0 = __builtin_fmaf(lo.x, ix, ox), x1 = __builtin_fmaf(hi.x, ix, ox);
            const float y0 = __builtin_fmaf(lo.y, iy, oy), y1 = __builtin_fmaf(hi.y, iy, oy);
            const float z0 = __builtin_fmaf(lo.z, iz, oz), z1 = __builtin_fmaf(hi.z, iz, oz);
            const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), t_min));
            const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), t_max));
            if (tn <= __builtin_fmaf(tf, 1.00001f, 1e-7f)) cand |= __float_as_uint(lo.w);
        }
        while (__ballot(cand != 0u) != 0ull) { // every lane its own next candidate, in primitive order
            if (cand != 0u) {
                const int k = __ffs((int)cand) - 1;
                cand &= cand - 1u;
                // a lane's own triangle: from LDS when the workgroup staged the list (two load sites, so that each is compiled
                // for its own address space: one site reading through either pointer becomes a flat load)
                float4 a, b, c;
                if (staged) {
                    typedef float f32x4 __attribute__((ext_vector_type(4)));
                    const __attribute__((address_space(3))) f32x4 *ls = (const __attribute__((address_space(3))) f32x4 *)staged; // LDS: ds_read, not flat
                    const f32x4 va = ls[3 * k], vb = ls[3 * k + 1], vc = ls[3 * k + 2];
                    a = make_float4(va.x, va.y, va.z, va.w); b = make_float4(vb.x, vb.y, vb.z, vb.w); c = make_float4(vc.x, vc.y, vc.z, vc.w);
                } else { a = sc.loose[3 * k]; b = sc.loose[3 * k + 1]; c = sc.loose[3 * k + 2]; }
                if (COUNT) (*n_scan)++;
                float t;
                if (tri_hit(f3(a.x, a.y, a.z), f3(b.x, b.y, b.z), f3(c.x, c.y, c.z), o, d, t_min, t_max, false, &t)) {
                    const int prim = __float_as_int(a.w);
                    if (t < best.t || (t == best.t && prim > best.prim)) {
                        best.t = t; best.tri = __float_as_int(c.w); best.prim = prim;
                    }
                }
            }
        }
        return best;
    }
    for (int k = 0; k < sc.n_loose; k++) {
        if (test(k)) break;
    }
    return best;
}

// world.objects.hit (hitable.rs:35-62): triangles, then spheres and cylinders in order.  A sphere or cylinder needs
// t < closest (geometry.rs:91,148), so a triangle at the same t wins, as in the reference's scan.  Triangles: when the
// loose list is in use (n_loose >= 0) the tree holds the big meshes only, so the loose triangles are scanned first and
// the walk starts with their closest hit as its bound; LOOSE_ONLY: the ray is known to miss every big mesh's box.
template <bool COUNT, bool USE_LDS = false, bool POSTPONE = false, bool LOOSE_ONLY = false>
__device__ __forceinline__ void world_hit(const DevScene &sc, const float4 *lds_nodes, F3 o, F3 d, float t_min, float t_max, float *t_out,
                                          int *code_out, int *prim_out, uint32_t *n_nodes, uint32_t *n_tris, uint32_t *n_scan = nullptr) {
    TriHit th{t_max, -1, -1};
    uint32_t scan_local = 0; // (work counting: tests of the scanned list go to n_scan, tests of tree leaves to n_tris)
    // (LOOSE_ONLY: no node cache is in use, and lds_nodes, when given, is the workgroup's copy of the loose records)
    if (LOOSE_ONLY || sc.n_loose >= 0) th = loose_traverse<false, COUNT>(sc, o, d, t_min, t_max, &scan_local, LOOSE_ONLY ? lds_nodes : nullptr);
    if (COUNT && n_scan) *n_scan += scan_local;
    if (!LOOSE_ONLY) {
        // a tree triangle at exactly the loose hit's distance is still accepted (geometry.rs:53) and wins with the larger id
        const TriHit tb = bvh_traverse<false, COUNT, USE_LDS, POSTPONE>(sc, lds_nodes, o, d, t_min, th.t, n_nodes, n_tris);
        if (tb.tri >= 0 && (tb.t < th.t || tb.prim > th.prim)) th = tb;
    }
    float closest = th.t;
    int code = th.tri >= 0 ? hit_code(HIT_KIND_TRI, th.tri) : HIT_NONE;
    int prim = th.prim;
    for (int i = 0; i < sc.n_sph; i++) {
        float t;
        if (sph_hit(sc.sph[i], o, d, t_min, closest, &t)) { closest = t; code = hit_code(HIT_KIND_SPH, i); prim = sc.sph_info[i].y; }
    }
    for (int i = 0; i < sc.n_cyl; i++) {
        float t;
        int kind;
        if (cyl_hit(sc.cyl + 5 * i, o, d, t_min, closest, &t, &kind)) { closest = t; code = hit_code(kind, i); prim = sc.cyl_info[i].y; }
    }
    *t_out = closest;
    *code_out = code;
    *prim_out = prim;
}

// occlusion query: world.hit(r, T_MIN, 1 - T_MIN).is_some() (light.rs:74,170,200,320)
template <bool COUNT, bool USE_LDS = false, bool LOOSE_ONLY = false, class SC = DevScene>
__device__ __forceinline__ bool world_occluded(const SC &sc, const float4 *lds_nodes, F3 o, F3 d, float t_min, float t_max,
                                               uint32_t *n_nodes, uint32_t *n_tris, uint32_t *n_scan = nullptr) {
    for (int i = 0; i < sc.n_sph; i++) {
        float t;
        if (sph_hit(sc.sph[i], o, d, t_min, t_max, &t)) return true;
    }
    for (int i = 0; i < sc.n_cyl; i++) {
        float t;
        int kind;
        if (cyl_hit(sc.cyl + 5 * i, o, d, t_min, t_max, &t, &kind)) return true;
    }
    const DevScene &g = sc;
    uint32_t scan_local = 0;
    const bool hit_loose = (LOOSE_ONLY || g.n_loose >= 0) && loose_traverse<true, COUNT>(sc, o, d, t_min, t_max, &scan_local).tri >= 0;
    if (COUNT && n_scan) *n_scan += scan_local;
    if (hit_loose) return true;
    if (LOOSE_ONLY) return false;
    return bvh_traverse<true, COUNT, USE_LDS>(sc, lds_nodes, o, d, t_min, t_max, n_nodes, n_tris).tri >= 0;
}

// ------------------------------------------------------------------------------------------------ lights
// (the functions below that only read the small tables — lights, light triangles, materials, spheres, cylinders — take
// the scene as a template parameter: the shading kernel hands them a view whose table pointers lead into LDS)
template <class SC>
__device__ __forceinline__ F3 ltri_vertex(const SC &sc, int tri, int k) {
    const float *p = sc.ltris + 9 * tri + 3 * k;
    return f3(p[0], p[1], p[2]);
}

// NaiveMesh::hit_both_side over a light's faces (mesh.rs:66-82)
template <class SC>
__device__ __forceinline__ bool light_mesh_hit(const SC &sc, const DevLight &L, F3 o, F3 d, float t_min, float t_max, float *t_out) {
    bool any = false;
    float closest = t_max;
    for (int f = 0; f < L.tri_count; f++) {
        int tri = L.tri_first + f;
        F3 v0 = ltri_vertex(sc, tri, 0);
        F3 E1 = ltri_vertex(sc, tri, 1) - v0;
        F3 E2 = ltri_vertex(sc, tri, 2) - v0;
        float t;
        if (tri_hit(v0, E1, E2, o, d, t_min, closest, true, &t)) { any = true; closest = t; }
    }
    *t_out = closest;
    return any;
}

// The same for a ray aimed at a point sampled on face `tri` of the light (PolygonLight::pdf inside visible_d,
// light.rs:328,334-342).  The light is planar, so any other face the ray meets is met at the same point of that plane,
// on the edge it shares with `tri`, at a distance that differs in the last bits at most: the sampled face alone gives
// the distance, and the loop over all faces runs only when rounding puts the sample outside its own face.
template <class SC>
__device__ __forceinline__ bool light_mesh_hit_sampled(const SC &sc, const DevLight &L, int tri, F3 o, F3 d, float t_min, float t_max, float *t_out) {
    const F3 v0 = ltri_vertex(sc, tri, 0);
    const F3 E1 = ltri_vertex(sc, tri, 1) - v0;
    const F3 E2 = ltri_vertex(sc, tri, 2) - v0;
    float t;
    if (tri_hit(v0, E1, E2, o, d, t_min, t_max, true, &t)) { *t_out = t; return true; }
    return light_mesh_hit(sc, L, o, d, t_min, t_max, t_out);
}

// Light::hit: light.rs:101-103, 223-238, 344-346
template <class SC>
__device__ __forceinline__ bool light_hit(const SC &sc, const DevLight &L, F3 o, F3 d, float t_min, float t_max, float *t_out) {
    if (L.kind == RT_LIGHT_DISK) {
        float hit_t = (L.oy - o.y) / d.y;
        if (hit_t < t_min) return false;
        float hx = o.x + hit_t * d.x;
        float hz = o.z + hit_t * d.z;
        float rx = hx - L.ox, rz = hz - L.oz;
        if (rx * rx + rz * rz <= L.radius * L.radius) { *t_out = hit_t; return true; }
        return false;
    }
    if (L.kind == RT_LIGHT_POLYGON) {
        // most rays do not point at the light: a slab test against its padded box (culling only, like a BVH node) skips
        // the two-sided triangle tests with their IEEE divisions
        const float ix = rcp_w(d.x), iy = rcp_w(d.y), iz = rcp_w(d.z);
        const float x0 = (L.bmin[0] - o.x) * ix, x1 = (L.bmax[0] - o.x) * ix;
        const float y0 = (L.bmin[1] - o.y) * iy, y1 = (L.bmax[1] - o.y) * iy;
        const float z0 = (L.bmin[2] - o.z) * iz, z1 = (L.bmax[2] - o.z) * iz;
        const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), t_min));
        const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), t_max));
        if (!(tn <= tf * 1.00001f + 1e-7f)) return false;
        return light_mesh_hit(sc, L, o, d, t_min, t_max, t_out);
    }
    return false;
}

// LightList::_hit (light.rs:355-364): the FIRST light in list order that is hit, not the nearest
template <class SC>
__device__ __forceinline__ int lights_hit(const SC &sc, F3 o, F3 d, float t_min, float t_max, float *t_out) {
    for (int i = 0; i < sc.n_light; i++) {
        float t;
        if (light_hit(sc, sc.lights[i], o, d, t_min, t_max, &t)) { *t_out = t; return i; }
    }
    return -1;
}

// light.rs:142-146
__device__ __forceinline__ float disk_pdf(const DevLight &L, F3 dir, float sq_dist) {
    float area = RTD_PI * (L.radius * L.radius);
    float cosine = fabsf(dot(dir, f3(0.0f, 1.0f, 0.0f)));
    return div_w(sq_dist, cosine * area);
}

// Light::pdf once the light's own hit distance is known: light.rs:97-99, 215-221, 334-342
__device__ __forceinline__ float light_pdf_with_t(const DevLight &L, F3 d, bool hit, float t) {
    if (!hit || L.kind == RT_LIGHT_POINT) return RTD_ZERO;
    float sq_dist = (t * t) * dot(d, d);
    if (L.kind == RT_LIGHT_DISK) return disk_pdf(L, d, sq_dist);
    float cosine = fabsf(dot(d, f3(L.nx, L.ny, L.nz)));
    return div_w(sq_dist, cosine * L.area);
}

// ------------------------------------------------------------------------------------------------ samplers
__device__ __forceinline__ bool in_disk(float range, float a, float b) { // sampler.rs:273-277
    float radius = range / 2.0f;
    float dx = a - radius, dy = b - radius;
    return dx * dx + dy * dy < radius * radius;
}
__device__ __forceinline__ float grid_coord(float interval, int m) { // sampler.rs:111,129-133: repeated addition
    float v = interval / 2.0f;
    for (int k = 0; k < m; k++) v += interval;
    return v;
}
// k-th point of a fresh sampler (sampler.rs:34-269); false when the sampler yields fewer points
__device__ __forceinline__ bool sampler_point(int kind, float range, int rate, int e, int k, Stream &st, float *a, float *b) {
    if (kind == RT_SAMPLER_WHITE || kind == RT_SAMPLER_BLUE) {
        if (k >= rate) return false;
        *a = st.next_f32() * range;
        *b = st.next_f32() * range;
        return true;
    }
    if (k >= e * e) return false;
    float interval = range / (float)e;
    float gi = grid_coord(interval, k % e), gj = grid_coord(interval, k / e);
    if (kind == RT_SAMPLER_UNIFORM) { *a = gi; *b = gj; return true; }
    float jitter = interval / 2.0f;
    *a = gi + st.next_f32() * jitter;
    *b = gj + st.next_f32() * jitter;
    return true;
}
// one point on a disk light for next-event estimation (DiskLight::visible_d, light.rs:183-213): the persistent
// sampler of the reference becomes a stratum drawn from the path node's own stream
__device__ __forceinline__ bool disk_sample_d(int kind, float radius, int ns2, int e, Stream &st, float *a, float *b) {
    float range = radius * 2.0f;
    if (kind == RT_SAMPLER_WHITE || kind == RT_SAMPLER_BLUE) {
        for (int tries = 0; tries < 16; tries++) {
            float pa = st.next_f32() * range, pb = st.next_f32() * range;
            if (in_disk(range, pa, pb)) { *a = pa; *b = pb; return true; }
        }
        return false;
    }
    int n = e * e;
    if (n <= 0) return false;
    int k0 = (int)st.next_index((uint32_t)n);
    for (int k = k0; k < n; k++) {
        float pa, pb;
        sampler_point(kind, range, ns2, e, k, st, &pa, &pb);
        if (in_disk(range, pa, pb)) { *a = pa; *b = pb; return true; }
    }
    return false;
}

// ------------------------------------------------------------------------------------------------ materials
__device__ __forceinline__ F3 reflect(F3 v, F3 n) { return v - 2.0f * dot(v, n) * n; } // material.rs:36-38

__device__ __forceinline__ F3 unit_vec_on_sphere(Stream &st) { // material.rs:123-136, at most 16 tries
    float r1 = 0.0f, r2 = 0.0f, rsq = 0.0f;
    for (int tries = 0; tries < 16; tries++) {
        float a = 1.0f - 2.0f * st.next_f32();
        float b = 1.0f - 2.0f * st.next_f32();
        float q = a * a + b * b;
        if (q < 1.0f) { r1 = a; r2 = b; rsq = q; break; }
    }
    float rh = 2.0f * sqrtf(1.0f - rsq);
    return f3(r1 * rh, r2 * rh, 1.0f - 2.0f * rsq);
}

__device__ __forceinline__ bool refract(F3 v, F3 n, float ni_over_nt, float t_min, F3 *out) { // material.rs:171-180
    F3 uv = normalize(v);
    float dt = dot(uv, n);
    float dd = 1.0f - ni_over_nt * ni_over_nt * (1.0f - dt * dt);
    if (dd > t_min) {
        *out = ni_over_nt * (uv - n * dt) - n * sqrtf(dd);
        return true;
    }
    return false;
}

__device__ __forceinline__ F3 albedo_of(const rt_material &m) { return f3(m.albedo[0], m.albedo[1], m.albedo[2]); }
__device__ __forceinline__ F3 f0_of(const rt_material &m) { return f3(m.f0[0], m.f0[1], m.f0[2]); }

__device__ __forceinline__ bool metal_brdf_ok(F3 din, F3 dout, F3 dnor, float t_min) { // material.rs:52-67
    din = normalize(din);
    dout = normalize(dout);
    return fabsf(dot(din + dout, dnor)) < t_min;
}
__device__ __forceinline__ bool dielectric_brdf_ok(const rt_material &m, F3 din, F3 dout, F3 dnor, float t_min) { // material.rs:188-217
    F3 outward;
    float ni_over_nt;
    if (dot(din, dnor) > t_min) { outward = -dnor; ni_over_nt = m.ref_idx; }
    else { outward = dnor; ni_over_nt = 1.0f / m.ref_idx; }
    din = normalize(din);
    dout = normalize(dout);
    F3 refracted;
    if (refract(din, outward, ni_over_nt, t_min, &refracted)) return vec_eq(refracted, dout, t_min);
    return fabsf(dot(din + dout, dnor)) < t_min;
}

// material.rs:275-303
__device__ __forceinline__ float ggx_distribution(F3 n, F3 h, float alpha) {
    float noh = dot(n, h);
    float noh2 = noh * noh;
    float alpha2 = alpha * alpha;
    float e = noh2 * (alpha2 - 1.0f) + 1.0f;
    return div_w(alpha2, RTD_PI * (e * e));
}
__device__ __forceinline__ float ggx_schlick(float nov, float k) { return div_w(nov, k + (1.0f - k) * nov); }
__device__ __forceinline__ float ggx_partial_geometry_term(F3 l, F3 v, F3 n, float alpha) {
    float nol = fabsf(dot(n, l));
    float nov = fabsf(dot(n, v));
    float r = sqrt_w(alpha);
    float k = ((r + 1.0f) * (r + 1.0f)) * 0.125f;
    return ggx_schlick(nol, k) * ggx_schlick(nov, k);
}
__device__ __forceinline__ F3 fresnel_schlick(F3 din, F3 dnor, F3 f0) {
    float a = 1.0f - fabsf(dot(dnor, din));
    return lerp(f0, f3(1.0f, 1.0f, 1.0f), a * a);
}

__device__ __forceinline__ void get_local_coordinate(F3 v1, F3 *bx, F3 *by, F3 *bz) { // material.rs:314-323
    v1 = normalize(v1);
    F3 v2;
    if (fabsf(v1.x) > fabsf(v1.y)) v2 = f3(-v1.z, 0.0f, v1.x) / sqrtf(v1.x * v1.x + v1.z * v1.z);
    else v2 = f3(0.0f, v1.z, -v1.y) / sqrtf(v1.y * v1.y + v1.z * v1.z);
    *bx = v2;
    *by = v1;
    *bz = cross(v1, v2);
}

// Material::brdf and Material::pdf evaluated together (both are always wanted for the same pair of directions):
// material.rs:89-103, 153-160, 248-262, 361-391
__device__ __forceinline__ void mat_brdf_pdf(const rt_material &m, F3 din, F3 dout, F3 dnor, float t_min, F3 *brdf, float *pdf) {
    switch (m.kind) {
    case RT_MAT_DIFFUSE:
        *brdf = albedo_of(m) * (1.0f / RTD_PI);
        *pdf = 1.0f / RTD_PI;
        return;
    case RT_MAT_METAL: {
        bool ok = metal_brdf_ok(din, dout, dnor, t_min);
        *brdf = ok ? albedo_of(m) : f3(0.0f, 0.0f, 0.0f);
        *pdf = ok ? 1.0f : RTD_ZERO;
        return;
    }
    case RT_MAT_DIELECTRIC: {
        bool ok = dielectric_brdf_ok(m, din, dout, dnor, t_min);
        *brdf = ok ? f3(1.0f, 1.0f, 1.0f) : f3(0.0f, 0.0f, 0.0f);
        *pdf = ok ? 1.0f : RTD_ZERO;
        return;
    }
    default: {
        F3 wg = normalize(dnor);
        F3 wi = normalize(dout);
        F3 wo = -normalize(din);
        F3 wm = normalize(wi + wo);
        float alpha = m.roughness * m.roughness;
        {
            float a2 = alpha * alpha;
            float cos_theta = fabsf(dot(wg, wm));
            float e = (a2 - 1.0f) * (cos_theta * cos_theta) + 1.0f;
            float dd = div_w(a2, RTD_PI * (e * e));
            *pdf = div_w(dd * cos_theta, 4.0f * fabsf(dot(wo, wm)));
        }
        if (dot(wi, wg) <= 0.0f || dot(wi, wm) <= 0.0f) { *brdf = f3(0.0f, 0.0f, 0.0f); return; }
        F3 f0 = lerp(f0_of(m), albedo_of(m), m.metallic);
        F3 f = fresnel_schlick(wi, wg, f0);
        float dterm = ggx_distribution(wg, wm, alpha);
        float g = ggx_partial_geometry_term(wi, wo, wg, alpha);
        F3 res = div_w(f * dterm * g, 4.0f * fabsf(dot(wg, wi)) * fabsf(dot(wg, wo)));
        *brdf = mul_v(albedo_of(m), res);
        return;
    }
    }
}

// Material::attenuation: material.rs:105-107, 162-164, 264-266
__device__ __forceinline__ F3 mat_attenuation(const rt_material &m) {
    if (m.kind == RT_MAT_DIELECTRIC) return f3(1.0f, 1.0f, 1.0f);
    return albedo_of(m);
}

// Material::scatter for Metal and Dielectric: material.rs:71-83, 221-242
__device__ __forceinline__ bool mat_scatter_specular(const rt_material &m, F3 d_in, F3 n, float t_min, F3 *out) {
    if (m.kind == RT_MAT_METAL) {
        F3 reflected = reflect(normalize(d_in), n);
        if (dot(reflected, n) > RTD_ZERO) { *out = reflected; return true; }
        return false;
    }
    F3 reflected = reflect(d_in, n);
    F3 outward;
    float ni_over_nt;
    if (dot(d_in, n) > t_min) { outward = -n; ni_over_nt = m.ref_idx; }
    else { outward = n; ni_over_nt = 1.0f / m.ref_idx; }
    F3 refracted;
    if (refract(d_in, outward, ni_over_nt, t_min, &refracted)) *out = refracted;
    else *out = reflected;
    return true;
}

// Material::scatter_d: material.rs:85-87, 144-151, 244-246, 334-359
__device__ __forceinline__ bool mat_scatter_d(const rt_material &m, F3 d_in, F3 n, float t_min, uint32_t pk, uint32_t nk, F3 *out) {
    switch (m.kind) {
    case RT_MAT_DIFFUSE: {
        Stream st(pk, nk, P_SCATTER);
        F3 d = unit_vec_on_sphere(st);
        if (dot(d, n) < 0.0f) d = -d;
        *out = d;
        return true;
    }
    case RT_MAT_METAL:
    case RT_MAT_DIELECTRIC: return mat_scatter_specular(m, d_in, n, t_min, out);
    default: {
        Stream st(pk, nk, P_SCATTER);
        float r0 = st.next_f32();
        float r1 = st.next_f32();
        float a = m.roughness * m.roughness;
        float a2 = a * a;
        float cos_theta = sqrtf((1.0f - r0) / ((a2 - 1.0f) * r0 + 1.0f));
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        float sphi, cphi;
        sincos_2pi(r1, &sphi, &cphi);
        float x = sin_theta * cphi;
        float y = cos_theta;
        float z = sin_theta * sphi;
        F3 bx, by, bz;
        get_local_coordinate(n, &bx, &by, &bz);
        F3 wm = normalize(x * bx + y * by + z * bz);
        *out = reflect(d_in, wm);
        return true;
    }
    }
}

// ------------------------------------------------------------------------------------------------ camera
__device__ __forceinline__ void get_ray(const rt_camera &c, float s, float t, F3 *o, F3 *d) { // camera.rs:29-34
    F3 origin = f3(c.origin[0], c.origin[1], c.origin[2]);
    F3 llc = f3(c.llc[0], c.llc[1], c.llc[2]);
    F3 h = f3(c.horizontal[0], c.horizontal[1], c.horizontal[2]);
    F3 v = f3(c.vertical[0], c.vertical[1], c.vertical[2]);
    *o = origin;
    *d = llc + s * h + t * v - origin;
}

__device__ __forceinline__ bool pixel_selected(const RenderConst &rc, int x, int y) {
    if (rc.tile_size <= 0) return true;
    int tiles_x = (rc.width + rc.tile_size - 1) / rc.tile_size;
    int t = (y / rc.tile_size) * tiles_x + (x / rc.tile_size);
    return t % rc.shard_count == rc.shard_rank;
}

// one camera sample (main.rs:35-39): sub-pixel offset from the pixel sampler, then Camera::get_ray
__device__ __forceinline__ bool camera_sample(const RenderConst &rc, int x, int y, int k, uint32_t pk, uint32_t nk, F3 *o, F3 *d) {
    Stream st(pk, nk, P_CAMERA);
    float a, b;
    if (!sampler_point(rc.pixel_sampler, 1.0f, rc.spp, rc.spp_edge, k, st, &a, &b)) return false;
    int i = x, j = rc.height - 1 - y;
    float u = ((float)i + a) / (float)rc.width;
    float v = ((float)j + b) / (float)rc.height;
    get_ray(rc.cam, u, v, o, d);
    return true;
}

} // namespace rtd

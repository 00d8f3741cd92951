// rt_api.cpp — the C ABI of include/rt_api.h: contexts, scene upload, the wavefront render loop.
// Host code, compiled by hipcc.  Citations are paths under the reference root (raytracer-rs).

#include "../../include/rt_api.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types and prototypes only: the library is loaded when rt_reduce_accum first needs it
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "rt_bvh.h"
#include "rt_bvh_device.h"
#include "rt_kernels.h"

using namespace rtd;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                              \
    do {                                                                                                           \
        hipError_t e_ = (expr);                                                                                    \
        if (e_ != hipSuccess) return fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)

template <class T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    hipError_t ensure(size_t count) {
        if (count <= n && p) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
        if (count == 0) return hipSuccess;
        hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
        if (e == hipSuccess) n = count;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

uint32_t env_u32_allow_zero(const char *name, uint32_t dflt) {
    const char *v = std::getenv(name);
    if (!v || !*v) return dflt;
    long long x = std::atoll(v);
    return x >= 0 ? (uint32_t)x : dflt;
}

uint32_t env_u32(const char *name, uint32_t dflt) {
    const char *v = std::getenv(name);
    if (!v || !*v) return dflt;
    long long x = std::atoll(v);
    return x > 0 ? (uint32_t)x : dflt;
}

} // namespace

struct rt_context {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // occlusion queries of the previous iteration run here, beside the next iteration's extend/shade
    hipEvent_t shadow_done = nullptr;
    // queue pool, sized on demand
    uint32_t capacity = 0, shadow_capacity = 0;
    DevBuf<float4> qa[2], qb[2], qc[2], ql[2], sa[2], sb[2];
    DevBuf<float2> hit, sc2[2];
    DevBuf<float> planes, accum_tmp;
    DevBuf<int32_t> blocks, prim_tmp;
    DevBuf<float> t_tmp;
    Ctl *ctl = nullptr;          // device
    unsigned char *scratch_ptr = nullptr; // this device's copy of the fixed array behind rt_render_static
    CtlHead *ctl_host = nullptr; // pinned copy of the head
    CtlHead *ctl_host_dev = nullptr; // the same memory as the device addresses it (k_plan writes the head there)
    std::vector<hipEvent_t> events;
};

struct rt_scene {
    rt_context *ctx = nullptr;
    DevBuf<float4> nodes, tris, tri_nrm, sph, loose, loose_boxes;
    DevBuf<uint4> cnodes;
    DevBuf<int2> sph_info, cyl_info;
    DevBuf<int32_t> tri_of_prim;
    DevBuf<float> cyl, ltris;
    DevBuf<rt_material> mats;
    DevBuf<DevLight> lights;
    DevScene dev{};
    bool has_microfacet = false, has_polygon_light = false;
    int n_prims = 0;
    int n_point_lights = 0, n_disk_lights = 0;
    rt_scene_info info{};
    // What the end of a render call has cost on this scene, per shader, in ms per bounce still to go: by the tail kernel
    // (one thread per path) and by the staged kernels at the population where the tail takes over; < 0 = not measured yet.
    double tail_ms_per_bounce[8] = {-1, -1, -1, -1, -1, -1, -1, -1}, staged_ms_per_bounce[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
};

// Device memory one render call works in: the two path-node queues, hit records, the two shadow-ray queues, the tiled
// colour planes, the list of selected 8x8 blocks and the loop state.  rt_render points it at the context's pool (grown
// on demand with hipMalloc); rt_render_static points it at a fixed array that exists from library load on.
struct RenderPool {
    float4 *qa[2], *qb[2], *qc[2], *ql[2], *sa[2], *sb[2];
    float2 *hit, *sc2[2];
    float *planes;
    int32_t *blocks;
    Ctl *ctl;
};

// ---- the fixed array behind rt_render_static ----------------------------------------------------------------------
// Refill level 2^16 path nodes: segments of 16384 nodes / shadow rays (8x their even share), frames of at most 2^21
// pixels after padding to 8x8 blocks (1920x1080 fits).
#define RT_STATIC_TARGET (1u << 16)
#define RT_STATIC_SEG_CAPACITY (8u * (RT_STATIC_TARGET / RT_NSEG))
#define RT_STATIC_CAPACITY ((size_t)RT_STATIC_SEG_CAPACITY * RT_NSEG)
#define RT_STATIC_MAX_PIXELS ((size_t)1 << 21)
#define RT_STATIC_ALIGN(x) (((size_t)(x) + 255u) & ~(size_t)255u)
#define RT_STATIC_OFF_CTL ((size_t)0)
#define RT_STATIC_OFF_BLOCKS (RT_STATIC_OFF_CTL + RT_STATIC_ALIGN(sizeof(Ctl)))
#define RT_STATIC_OFF_PLANES (RT_STATIC_OFF_BLOCKS + RT_STATIC_ALIGN((RT_STATIC_MAX_PIXELS / 64) * sizeof(int32_t)))
#define RT_STATIC_OFF_NODES (RT_STATIC_OFF_PLANES + RT_STATIC_ALIGN(3 * RT_STATIC_MAX_PIXELS * sizeof(float)))
#define RT_STATIC_OFF_HIT (RT_STATIC_OFF_NODES + 8 * RT_STATIC_CAPACITY * sizeof(float4))
#define RT_STATIC_OFF_SHADOW (RT_STATIC_OFF_HIT + RT_STATIC_CAPACITY * sizeof(float2))
#define RT_STATIC_SCRATCH_BYTES (RT_STATIC_OFF_SHADOW + RT_STATIC_CAPACITY * (4 * sizeof(float4) + 2 * sizeof(float2)))

// In host code this name is only a handle for hipGetSymbolAddress: every GPU holds its own copy at its own address,
// looked up once per device in rt_context_create and kept in g_scratch_ptr.  Host code never passes the name itself
// to a kernel, a memset or a copy.
static __device__ __align__(256) unsigned char scratch[RT_STATIC_SCRATCH_BYTES];

#define RT_MAX_DEVICES 64
static unsigned char *g_scratch_ptr[RT_MAX_DEVICES]; // per device; written under g_scratch_mutex
static std::mutex g_scratch_mutex;

namespace {

void drop_communicators_of(int device); // RCCL communicators that include the device (defined with rt_reduce_accum)

int set_device(rt_context *ctx) {
    HIP_TRY(hipSetDevice(ctx->device));
    return RT_OK;
}

// lib.rs:31-34 on the host
bool vec_eq3(const float *a, const float *b) {
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return dx * dx + dy * dy + dz * dz < 0.001f;
}
void tri_normal_area(const float *v, float *n, float *area) { // geometry.rs:16-26
    float e1[3] = {v[3] - v[0], v[4] - v[1], v[5] - v[2]}, e2[3] = {v[6] - v[0], v[7] - v[1], v[8] - v[2]};
    float c[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
    float len2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    float inv = 1.0f / std::sqrt(len2);
    n[0] = c[0] * inv; n[1] = c[1] * inv; n[2] = c[2] * inv;
    *area = 0.5f * std::sqrt(len2);
}

int check_params(const rt_params &p) {
    if (p.width <= 0 || p.height <= 0 || p.spp <= 0) return fail(RT_ERR_INVALID, "width, height and spp must be positive");
    if ((long long)p.width * p.height > (1ll << 24) || p.width > 65536) return fail(RT_ERR_INVALID, "at most 2^24 pixels per frame and 65536 per row");
    if (p.spp_begin < 0 || p.spp_end < p.spp_begin || p.spp_end > p.spp) return fail(RT_ERR_INVALID, "bad sample range");
    if (p.shader < RT_SHADER_NORMAL || p.shader > RT_SHADER_MIS) return fail(RT_ERR_INVALID, "bad shader");
    if (p.pixel_sampler < RT_SAMPLER_WHITE || p.pixel_sampler > RT_SAMPLER_BLUE) return fail(RT_ERR_INVALID, "unexpected integer"); // sampler.rs:19
    if (p.tile_size < 0 || (p.tile_size > 0 && (p.shard_count <= 0 || p.shard_rank < 0 || p.shard_rank >= p.shard_count)))
        return fail(RT_ERR_INVALID, "bad tile sharding");
    if (p.max_depth < 0 || p.max_depth > 62) return fail(RT_ERR_INVALID, "max_depth out of range");
    if (p.light_samples < 0) return fail(RT_ERR_INVALID, "light_samples must be >= 0");
    return RT_OK;
}

// combinations the reference stops on with assert!(false): material.rs:328-331,394-397, light.rs:292-301
int check_supported(const rt_scene &s, const rt_params &p) {
    if (p.shader == RT_SHADER_WHITTED && s.has_microfacet) return fail(RT_ERR_UNSUPPORTED, "whitted shader: Microfacet::scatter is not supported");
    if (p.shader == RT_SHADER_WHITTED && s.has_polygon_light) return fail(RT_ERR_UNSUPPORTED, "whitted shader: PolygonLight::visible is not supported");
    if (p.shader == RT_SHADER_PATH && s.has_microfacet) return fail(RT_ERR_UNSUPPORTED, "path shader: Microfacet::attenuation is not supported");
    return RT_OK;
}

RenderConst make_const(const rt_camera &cam, const rt_params &p) {
    RenderConst rc;
    std::memset(&rc, 0, sizeof(rc));
    rc.cam = cam;
    rc.width = p.width; rc.height = p.height; rc.spp = p.spp; rc.spp_begin = p.spp_begin; rc.spp_end = p.spp_end;
    rc.tile_size = p.tile_size; rc.shard_rank = p.shard_rank; rc.shard_count = p.shard_count;
    rc.pixel_sampler = p.pixel_sampler; rc.light_samples = p.light_samples; rc.max_depth = p.max_depth;
    rc.spp_edge = (int)std::sqrt((double)p.spp);           // sampler.rs:104
    rc.ns2_edge = (int)std::sqrt((double)p.light_samples);
    rc.t_min = p.t_min; rc.t_max = p.t_max; rc.seed = p.seed;
    rc.flags = p.flags & (RT_FLAG_RUSSIAN_ROULETTE | RT_FLAG_COSINE_DIFFUSE);
    rc.tiles_w = (uint32_t)((p.width + 7) / 8);
    rc.plane_stride = 64u * rc.tiles_w * (uint32_t)((p.height + 7) / 8);
    rc.w_magic = ((1ull << 40) + (unsigned long long)p.width - 1ull) / (unsigned long long)p.width;
    return rc;
}

bool pixel_selected_host(const rt_params &p, int x, int y) {
    if (p.tile_size <= 0) return true;
    int tiles_x = (p.width + p.tile_size - 1) / p.tile_size;
    int t = (y / p.tile_size) * tiles_x + (x / p.tile_size);
    return t % p.shard_count == p.shard_rank;
}

// hipMalloc failures of the queue pool are reported as RT_ERR_NOMEM; any other HIP error as RT_ERR_DEVICE
#define HIP_TRY_ALLOC(expr)                                                                                        \
    do {                                                                                                           \
        hipError_t e_ = (expr);                                                                                    \
        if (e_ == hipErrorOutOfMemory) return fail(RT_ERR_NOMEM, std::string(#expr) + ": " + hipGetErrorString(e_)); \
        if (e_ != hipSuccess) return fail(RT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)

int ensure_queues(rt_context *ctx, uint32_t capacity, uint32_t shadow_capacity, size_t n_pixels) {
    if (capacity > ctx->capacity) {
        // a failure part-way leaves some buffers freed: the pool counts as empty until every buffer has its new size
        ctx->capacity = 0;
        for (int k = 0; k < 2; k++) {
            HIP_TRY_ALLOC(ctx->qa[k].ensure(capacity));
            HIP_TRY_ALLOC(ctx->qb[k].ensure(capacity));
            HIP_TRY_ALLOC(ctx->qc[k].ensure(capacity));
            HIP_TRY_ALLOC(ctx->ql[k].ensure(capacity));
        }
        HIP_TRY_ALLOC(ctx->hit.ensure(capacity));
        ctx->capacity = capacity;
    }
    if (shadow_capacity > ctx->shadow_capacity) {
        ctx->shadow_capacity = 0;
        for (int k = 0; k < 2; k++) {
            HIP_TRY_ALLOC(ctx->sa[k].ensure(shadow_capacity));
            HIP_TRY_ALLOC(ctx->sb[k].ensure(shadow_capacity));
            HIP_TRY_ALLOC(ctx->sc2[k].ensure(shadow_capacity));
        }
        ctx->shadow_capacity = shadow_capacity;
    }
    HIP_TRY_ALLOC(ctx->planes.ensure(3 * n_pixels));
    return RT_OK;
}

struct StageTimer {
    rt_context *ctx;
    bool on;
    bool failed = false;
    size_t used = 0;
    struct Span { int stage; size_t e0; };
    std::vector<Span> spans;
    double ms[4] = {0, 0, 0, 0};
    hipEvent_t get() {
        if (used == ctx->events.size()) {
            hipEvent_t e = nullptr;
            if (hipEventCreate(&e) != hipSuccess) { on = false; failed = true; return nullptr; } // timing is dropped, the render goes on
            ctx->events.push_back(e);
        }
        return ctx->events[used++];
    }
    hipStream_t cur_stream = nullptr;
    void begin(int stage, hipStream_t stream = nullptr) {
        if (!on) return;
        cur_stream = stream ? stream : ctx->stream;
        const size_t e0 = used;
        hipEvent_t e = get();
        if (!e) return;
        spans.push_back({stage, e0});
        (void)hipEventRecord(e, cur_stream);
    }
    void end() {
        if (!on) return;
        hipEvent_t e = get();
        if (!e) { spans.pop_back(); return; }
        (void)hipEventRecord(e, cur_stream);
    }
    // call after a stream synchronisation
    void collect() {
        if (!on && !failed) return;
        if (failed) { spans.clear(); used = 0; for (double &m : ms) m = 0.0; return; }
        for (const Span &s : spans) {
            float t = 0.0f;
            hipError_t e = hipEventElapsedTime(&t, ctx->events[s.e0], ctx->events[s.e0 + 1]);
            if (e == hipErrorNotReady) { // (the host learns of the iteration's end from k_plan itself, perhaps before the runtime has looked at the events)
                (void)hipEventSynchronize(ctx->events[s.e0 + 1]);
                e = hipEventElapsedTime(&t, ctx->events[s.e0], ctx->events[s.e0 + 1]);
            }
            if (e == hipSuccess) ms[s.stage] += t;
        }
        spans.clear();
        used = 0;
    }
};

} // namespace

extern "C" {

const char *rt_last_error(void) { return g_err.c_str(); }

int32_t rt_bvh_build_host(const float *tri_v, int32_t n_tri, int32_t max_top, int32_t ordered, float *nodes_out, float *tris_out, int32_t *n_nodes,
                          int32_t *n_top, int32_t *n_orders) {
    if (n_tri < 0 || (n_tri > 0 && !tri_v) || max_top < 0) return fail(RT_ERR_INVALID, "rt_bvh_build_host: bad arguments");
    std::vector<int32_t> prim((size_t)n_tri), mat((size_t)n_tri, 0);
    for (int i = 0; i < n_tri; i++) prim[(size_t)i] = i;
    BvhBuild b;
    build_bvh(tri_v, prim.data(), mat.data(), n_tri, max_top, ordered != 0, (int)env_u32("RT_BVH_LEAF", 2), (int)env_u32("RT_BVH_ARITY", 8), &b);
    if (n_nodes) *n_nodes = b.n_nodes;
    if (n_top) *n_top = b.n_top;
    if (n_orders) *n_orders = b.n_orders;
    if (nodes_out && !b.nodes.empty()) std::memcpy(nodes_out, b.nodes.data(), b.nodes.size() * sizeof(float));
    if (tris_out && !b.tris.empty()) std::memcpy(tris_out, b.tris.data(), b.tris.size() * sizeof(float));
    return RT_OK;
}

int32_t rt_bvh_build_host_compact(const float *tri_v, int32_t n_tri, int32_t ordered, uint32_t *cnodes_out, float *q_params, int32_t *n_nodes,
                                  int32_t *n_orders) {
    if (n_tri < 0 || (n_tri > 0 && !tri_v)) return fail(RT_ERR_INVALID, "rt_bvh_build_host_compact: bad arguments");
    std::vector<int32_t> prim((size_t)n_tri), mat((size_t)n_tri, 0);
    for (int i = 0; i < n_tri; i++) prim[(size_t)i] = i;
    BvhBuild b;
    build_bvh(tri_v, prim.data(), mat.data(), n_tri, 0, ordered != 0, (int)env_u32("RT_BVH_LEAF", 2), (int)env_u32("RT_BVH_ARITY", 8), &b);
    if (n_nodes) *n_nodes = b.n_nodes;
    if (n_orders) *n_orders = b.n_orders;
    if (q_params) for (int a = 0; a < 3; a++) { q_params[a] = b.q_lo[a]; q_params[3 + a] = b.q_scale[a]; }
    if (cnodes_out && !b.cnodes.empty()) std::memcpy(cnodes_out, b.cnodes.data(), b.cnodes.size() * sizeof(uint32_t));
    return RT_OK;
}

void rt_params_default(rt_params *p, int32_t width, int32_t height, int32_t spp) {
    std::memset(p, 0, sizeof(*p));
    p->width = width; p->height = height; p->spp = spp;
    p->spp_begin = 0; p->spp_end = spp;
    p->tile_size = 0; p->shard_rank = 0; p->shard_count = 1;
    p->shader = RT_SHADER_MIS;
    p->pixel_sampler = RT_SAMPLER_BLUE; // main.rs:31
    p->light_samples = 9;               // NS2, lib.rs:24
    p->max_depth = 40;                  // shader.rs:94
    p->t_min = 0.001f; p->t_max = 10000000.0f; // lib.rs:17-18
    p->seed = 1;
    p->flags = 0;
}

int32_t rt_context_create(int32_t device_id, rt_context **out) {
    if (!out) return fail(RT_ERR_INVALID, "null argument");
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n) return fail(RT_ERR_INVALID, "no such device");
    HIP_TRY(hipSetDevice(device_id));
    rt_context *ctx = new rt_context();
    ctx->device = device_id;
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->shadow_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipMalloc((void **)&ctx->ctl, sizeof(Ctl));
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->ctl_host, sizeof(CtlHead), hipHostMallocMapped);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&ctx->ctl_host_dev, ctx->ctl_host, 0);
    if (e == hipSuccess && device_id < RT_MAX_DEVICES) {
        // address of this device's copy of the fixed array (the device is current: hipSetDevice above).  A failed
        // lookup only takes rt_render_static away (it reports the missing array itself); rt_render does not need it.
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[device_id]) {
            void *ptr = nullptr;
            if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)) == hipSuccess) g_scratch_ptr[device_id] = static_cast<unsigned char *>(ptr);
            else (void)hipGetLastError(); // clears the sticky error
        }
        ctx->scratch_ptr = g_scratch_ptr[device_id];
    }
    if (e != hipSuccess) {
        rt_context_destroy(ctx);
        return fail(RT_ERR_DEVICE, std::string("context creation: ") + hipGetErrorString(e));
    }
    *out = ctx;
    return RT_OK;
}

int32_t rt_context_destroy(rt_context *ctx) {
    if (!ctx) return RT_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    drop_communicators_of(ctx->device); // they were created on this context's stream; a later reduction makes new ones
    for (int k = 0; k < 2; k++) { ctx->qa[k].release(); ctx->qb[k].release(); ctx->qc[k].release(); ctx->ql[k].release(); }
    ctx->hit.release();
    for (int k = 0; k < 2; k++) { ctx->sa[k].release(); ctx->sb[k].release(); ctx->sc2[k].release(); }
    ctx->planes.release(); ctx->accum_tmp.release(); ctx->blocks.release(); ctx->prim_tmp.release(); ctx->t_tmp.release();
    for (hipEvent_t e : ctx->events) (void)hipEventDestroy(e);
    if (ctx->ctl) (void)hipFree(ctx->ctl);
    if (ctx->ctl_host) (void)hipHostFree(ctx->ctl_host);
    if (ctx->shadow_done) (void)hipEventDestroy(ctx->shadow_done);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RT_OK;
}

void *rt_context_stream(rt_context *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int32_t rt_scene_create(rt_context *ctx, const rt_scene_desc *d, rt_scene **out) {
    return rt_scene_create_ex(ctx, d, env_u32_allow_zero("RT_BVH_DEVICE", 0) != 0 ? RT_SCENE_BVH_ON_DEVICE : 0u, out);
}

int32_t rt_scene_get_info(const rt_scene *s, rt_scene_info *info) {
    if (!s || !info) return fail(RT_ERR_INVALID, "null argument");
    *info = s->info;
    return RT_OK;
}

int32_t rt_scene_create_ex(rt_context *ctx, const rt_scene_desc *d, uint32_t flags, rt_scene **out) {
    if (!ctx || !d || !out) return fail(RT_ERR_INVALID, "null argument");
    if (flags & ~(uint32_t)RT_SCENE_BVH_ON_DEVICE) return fail(RT_ERR_INVALID, "unknown scene flag");
    if (d->n_tri < 0 || d->n_sph < 0 || d->n_cyl < 0 || d->n_mat < 0 || d->n_light < 0 || d->n_light_tri < 0 || d->n_obj < 0)
        return fail(RT_ERR_INVALID, "negative count");
    if (d->n_tri >= (1 << 25)) return fail(RT_ERR_INVALID, "too many triangles");
    int rc = set_device(ctx);
    if (rc) return rc;

    // world order -> primitive ids (hitable.rs:35-56: objects in list order, a mesh expanded in face_list order)
    std::vector<int32_t> tri_prim(d->n_tri), sph_prim(d->n_sph), cyl_prim(d->n_cyl);
    int prim = 0, nt = 0, ns = 0, nc = 0;
    for (int i = 0; i < d->n_obj; i++) {
        const rt_object &ob = d->objects[i];
        switch (ob.kind) {
        case RT_OBJ_TRIANGLE:
            if (ob.first != nt || ob.count != 1 || nt + 1 > d->n_tri) return fail(RT_ERR_INVALID, "objects out of world order");
            tri_prim[nt++] = prim++;
            break;
        case RT_OBJ_MESH:
            if (ob.first != nt || ob.count < 0 || nt + ob.count > d->n_tri) return fail(RT_ERR_INVALID, "objects out of world order");
            for (int f = 0; f < ob.count; f++) tri_prim[nt++] = prim++;
            break;
        case RT_OBJ_SPHERE:
            if (ob.first != ns || ob.count != 1 || ns + 1 > d->n_sph) return fail(RT_ERR_INVALID, "objects out of world order");
            sph_prim[ns++] = prim++;
            break;
        case RT_OBJ_CYLINDER:
            if (ob.first != nc || ob.count != 1 || nc + 1 > d->n_cyl) return fail(RT_ERR_INVALID, "objects out of world order");
            cyl_prim[nc++] = prim++;
            break;
        default: return fail(RT_ERR_INVALID, "bad object kind");
        }
    }
    if (nt != d->n_tri || ns != d->n_sph || nc != d->n_cyl) return fail(RT_ERR_INVALID, "objects do not cover the primitive arrays");

    bool has_mf = false;
    auto mat_ok = [&](int m) {
        if (m < 0 || m >= d->n_mat) return false;
        has_mf |= d->mats[m].kind == RT_MAT_MICROFACET;
        return true;
    };
    for (int i = 0; i < d->n_mat; i++)
        if (d->mats[i].kind < RT_MAT_DIFFUSE || d->mats[i].kind > RT_MAT_MICROFACET) return fail(RT_ERR_INVALID, "bad material kind");
    for (int i = 0; i < d->n_tri; i++) if (!mat_ok(d->tri_mat[i])) return fail(RT_ERR_INVALID, "no material"); // shader.rs:46,81,106
    for (int i = 0; i < d->n_sph; i++) if (!mat_ok(d->sph_mat[i])) return fail(RT_ERR_INVALID, "no material");
    for (int i = 0; i < d->n_cyl; i++) if (!mat_ok(d->cyl_mat[i])) return fail(RT_ERR_INVALID, "no material");

    // lights: PolygonLight::new (light.rs:252-269)
    std::vector<DevLight> lights(d->n_light);
    bool has_poly = false;
    for (int i = 0; i < d->n_light; i++) {
        const rt_light &l = d->lights[i];
        DevLight &L = lights[i];
        std::memset(&L, 0, sizeof(L));
        L.kind = l.kind;
        L.ox = l.origin[0]; L.oy = l.origin[1]; L.oz = l.origin[2];
        L.radius = l.radius;
        L.sx = l.spectrum[0]; L.sy = l.spectrum[1]; L.sz = l.spectrum[2];
        L.tri_first = l.tri_first; L.tri_count = l.tri_count; L.sampler_kind = l.sampler_kind;
        if (l.kind == RT_LIGHT_POLYGON) {
            has_poly = true;
            if (l.tri_first < 0 || l.tri_count < 1 || l.tri_first + l.tri_count > d->n_light_tri) return fail(RT_ERR_INVALID, "polygon light: bad triangle range");
            float n0[3], a0;
            tri_normal_area(d->light_tri_v + 9 * (size_t)l.tri_first, n0, &a0);
            float area = 0.0f;
            for (int f = 0; f < l.tri_count; f++) {
                float n[3], a;
                tri_normal_area(d->light_tri_v + 9 * (size_t)(l.tri_first + f), n, &a);
                area += a;
                if (!vec_eq3(n0, n)) return fail(RT_ERR_INVALID, "the mesh of PolygonLight is not 2D!");
            }
            L.area = area;
            L.nx = n0[0]; L.ny = n0[1]; L.nz = n0[2];
            for (int a = 0; a < 3; a++) { L.bmin[a] = 1e30f; L.bmax[a] = -1e30f; }
            for (int v = 0; v < l.tri_count * 3; v++) {
                const float *q = d->light_tri_v + 9 * (size_t)l.tri_first + 3 * (size_t)v;
                for (int a = 0; a < 3; a++) { L.bmin[a] = std::min(L.bmin[a], q[a]); L.bmax[a] = std::max(L.bmax[a], q[a]); }
            }
            float ext = 0.0f, mag = 0.0f;
            for (int a = 0; a < 3; a++) {
                ext = std::max(ext, L.bmax[a] - L.bmin[a]);
                mag = std::max(mag, std::max(std::fabs(L.bmin[a]), std::fabs(L.bmax[a])));
            }
            const float pad = 1e-4f * ext + 1e-5f * mag + 1e-6f; // same padding as the BVH leaves
            for (int a = 0; a < 3; a++) { L.bmin[a] -= pad; L.bmax[a] += pad; }
        } else if (l.kind == RT_LIGHT_DISK) {
            if (l.sampler_kind < RT_SAMPLER_WHITE || l.sampler_kind > RT_SAMPLER_BLUE) return fail(RT_ERR_INVALID, "unexpected integer");
        } else if (l.kind != RT_LIGHT_POINT) return fail(RT_ERR_INVALID, "bad light kind");
    }

    rt_scene *s = new rt_scene();
    s->ctx = ctx;
    s->has_microfacet = has_mf;
    s->has_polygon_light = has_poly;
    s->n_prims = prim;
    auto cleanup_fail = [&](hipError_t e, const char *what) {
        std::string m = std::string(what) + ": " + hipGetErrorString(e);
        rt_scene_destroy(s);
        return fail(RT_ERR_DEVICE, m);
    };
#define UP(buf, src, count, T)                                                                                       \
    do {                                                                                                             \
        size_t c_ = (size_t)(count);                                                                                 \
        hipError_t e_ = (buf).ensure(c_ ? c_ : 1);                                                                   \
        if (e_ == hipSuccess && c_) e_ = hipMemcpy((buf).p, (src), c_ * sizeof(T), hipMemcpyHostToDevice);           \
        if (e_ != hipSuccess) return cleanup_fail(e_, #buf);                                                         \
    } while (0)

    // Boxes around the big meshes (at most RT_MAX_HEAVY, the largest first; further meshes are merged into the last box),
    // each the union of its triangles' boxes padded as the BVH pads them: a ray that misses all of them cannot hit a
    // triangle of a big mesh.  Such rays are kept apart in the queues, and when the triangles outside the big meshes
    // are few (RT_LOOSE_MAX, default 16; 0 = never) they are kept OUT of the tree: every ray scans them in a loop the
    // whole wave runs together, and only rays that touch a big mesh's box walk the tree, which then holds the big
    // meshes alone and is entered with the loose triangles' closest hit as its bound.
    std::vector<char> in_big((size_t)std::max(d->n_tri, 1), 0); // by position in the triangle array
    {
        struct MB { int count; float mn[3], mx[3]; };
        std::vector<MB> mbs;
        const bool classify = env_u32("RT_NO_HEAVY", 0) == 0;
        for (int i = 0; i < d->n_obj && classify; i++) {
            const rt_object &ob = d->objects[i];
            if (ob.kind != RT_OBJ_MESH || ob.count < 64) continue;
            MB mb{ob.count, {1e30f, 1e30f, 1e30f}, {-1e30f, -1e30f, -1e30f}};
            for (int f = 0; f < ob.count; f++) {
                const float *v = d->tri_v + 9 * (size_t)(ob.first + f);
                float mn[3], mx[3], ext = 0.0f, mag = 0.0f;
                for (int a = 0; a < 3; a++) {
                    mn[a] = std::min(std::min(v[a], v[3 + a]), v[6 + a]);
                    mx[a] = std::max(std::max(v[a], v[3 + a]), v[6 + a]);
                    ext = std::max(ext, mx[a] - mn[a]);
                    mag = std::max(mag, std::max(std::fabs(mn[a]), std::fabs(mx[a])));
                }
                const float pad = 1e-4f * ext + 1e-5f * mag + 1e-6f; // same padding as the BVH leaves
                for (int a = 0; a < 3; a++) { mb.mn[a] = std::min(mb.mn[a], mn[a] - pad); mb.mx[a] = std::max(mb.mx[a], mx[a] + pad); }
                in_big[(size_t)(ob.first + f)] = 1;
            }
            mbs.push_back(mb);
        }
        std::sort(mbs.begin(), mbs.end(), [](const MB &a, const MB &b) { return a.count > b.count; });
        s->dev.n_heavy = 0;
        for (size_t i = 0; i < mbs.size(); i++) {
            int k = (int)std::min<size_t>(i, RT_MAX_HEAVY - 1);
            if ((int)i < RT_MAX_HEAVY) {
                for (int a = 0; a < 3; a++) { s->dev.heavy_min[k][a] = mbs[i].mn[a]; s->dev.heavy_max[k][a] = mbs[i].mx[a]; }
                s->dev.n_heavy = k + 1;
            } else {
                for (int a = 0; a < 3; a++) {
                    s->dev.heavy_min[k][a] = std::min(s->dev.heavy_min[k][a], mbs[i].mn[a]);
                    s->dev.heavy_max[k][a] = std::max(s->dev.heavy_max[k][a], mbs[i].mx[a]);
                }
            }
        }
    }
    std::vector<int> loose_src; // positions in the triangle array, which is in primitive-id order
    for (int i = 0; i < d->n_tri; i++) if (!in_big[(size_t)i]) loose_src.push_back(i);
    const size_t loose_max = env_u32_allow_zero("RT_LOOSE_MAX", 16);
    const bool use_loose = loose_max > 0 && loose_src.size() <= loose_max;
    // the triangles the tree is built over: all of them, or the big meshes' when the others are scanned
    std::vector<float> sub_v;
    std::vector<int32_t> sub_prim, sub_mat;
    const float *bvh_v = d->tri_v;
    const int32_t *bvh_prim = tri_prim.data(), *bvh_mat = d->tri_mat;
    int n_bvh = d->n_tri;
    if (use_loose && !loose_src.empty()) {
        n_bvh = d->n_tri - (int)loose_src.size();
        sub_v.reserve(9 * (size_t)n_bvh); sub_prim.reserve((size_t)n_bvh); sub_mat.reserve((size_t)n_bvh);
        for (int i = 0; i < d->n_tri; i++) {
            if (!in_big[(size_t)i]) continue;
            sub_v.insert(sub_v.end(), d->tri_v + 9 * (size_t)i, d->tri_v + 9 * (size_t)i + 9);
            sub_prim.push_back(tri_prim[(size_t)i]);
            sub_mat.push_back(d->tri_mat[i]);
        }
        bvh_v = sub_v.data(); bvh_prim = sub_prim.data(); bvh_mat = sub_mat.data();
    }

    BvhBuild bvh;
    int n_cnodes = 0; // nodes per order of the 16-byte copy (the device build thins that copy only)
    const int max_top = (int)env_u32_allow_zero("RT_BVH_TOP", 32);
    const bool ordered = env_u32_allow_zero("RT_BVH_ORDERED", 1) != 0;
    const bool on_device = (flags & RT_SCENE_BVH_ON_DEVICE) != 0 && n_bvh > 0;
    const auto t_build = std::chrono::steady_clock::now();
    if (on_device) {
        // BVHTree::build_tree (mesh.rs:225-261) on the GPU: see rt_bvh_device.hip.  2n-1 nodes, one triangle per leaf,
        // no LDS-resident top (the node numbering is the builder's: inner nodes first, then leaves).
        bvh.n_nodes = 2 * n_bvh - 1;
        bvh.n_orders = ordered ? 8 : 1;
        bvh.n_top = 0;
    } else {
        build_bvh(bvh_v, bvh_prim, bvh_mat, n_bvh, max_top, ordered, (int)env_u32("RT_BVH_LEAF", 2), (int)env_u32("RT_BVH_ARITY", 8), &bvh);
    }
    // the traversal addresses nodes and triangles through 32-bit byte offsets of a buffer descriptor
    if ((unsigned long long)bvh.n_nodes * (unsigned long long)bvh.n_orders * 32ull >= (1ull << 32) || (unsigned long long)d->n_tri * 48ull >= (1ull << 32)) {
        rt_scene_destroy(s);
        return fail(RT_ERR_INVALID, "scene too large: node or triangle array reaches 4 GiB");
    }
    auto put_record = [&](float *dst, const float *v, int32_t prim_id, int32_t mat_id) { // v0 + id, E1 + material, E2 (geometry.rs:30-31)
        dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2];
        std::memcpy(&dst[3], &prim_id, 4);
        dst[4] = v[3] - v[0]; dst[5] = v[4] - v[1]; dst[6] = v[5] - v[2];
        std::memcpy(&dst[7], &mat_id, 4);
        dst[8] = v[6] - v[0]; dst[9] = v[7] - v[1]; dst[10] = v[8] - v[2];
        dst[11] = 0.0f;
    };
    if (on_device) {
        hipError_t e = s->nodes.ensure(2 * (size_t)bvh.n_nodes * (size_t)bvh.n_orders);
        if (e == hipSuccess) e = s->cnodes.ensure((size_t)bvh.n_nodes * (size_t)bvh.n_orders);
        if (e == hipSuccess) e = s->tris.ensure(3 * (size_t)d->n_tri);
        std::vector<uint32_t> order((size_t)n_bvh);
        float qp[6] = {0, 0, 0, 1, 1, 1};
        // thinning of the 16-byte copy (RT_BVH_DEV_WIDEN: per cent of the parent's surface from which a child node is dissolved,
        // 0 = never; RT_BVH_DEV_PAIRS=0: one triangle per leaf)
        const float widen = (float)env_u32_allow_zero("RT_BVH_DEV_WIDEN", 60) / 100.0f;
        const bool pair_leaves = env_u32_allow_zero("RT_BVH_DEV_PAIRS", 1) != 0;
        if (e == hipSuccess)
            e = build_bvh_device(ctx->stream, bvh_v, bvh_prim, bvh_mat, n_bvh, ordered, s->nodes.p, s->cnodes.p, s->tris.p, order.data(), qp, widen, pair_leaves,
                                 env_u32_allow_zero("RT_BVH_DEV_ROTATE", 1) != 0, &n_cnodes);
        for (int a = 0; a < 3; a++) { bvh.q_lo[a] = qp[a]; bvh.q_scale[a] = qp[3 + a]; }
        if (e != hipSuccess) return cleanup_fail(e, "BVH build on the device");
        // host copy of the triangle records, in the device's leaf order: the primitive -> record table below reads it
        bvh.tris.resize(12 * (size_t)n_bvh);
        for (int i = 0; i < n_bvh; i++) {
            const uint32_t t = order[(size_t)i];
            if (t >= (uint32_t)n_bvh) return cleanup_fail(hipErrorUnknown, "BVH build on the device returned a bad order");
            put_record(&bvh.tris[12 * (size_t)i], bvh_v + 9 * (size_t)t, bvh_prim[t], bvh_mat[t]);
        }
    }
    // the loose triangles' records follow the tree's, in primitive-id order; no leaf refers to them
    const size_t loose_first = bvh.tris.size() / 12;
    if (n_bvh < d->n_tri) {
        bvh.tris.resize(12 * (size_t)d->n_tri);
        for (size_t j = 0; j < loose_src.size(); j++) {
            const int i = loose_src[j];
            put_record(&bvh.tris[12 * (loose_first + j)], d->tri_v + 9 * (size_t)i, tri_prim[(size_t)i], d->tri_mat[i]);
        }
    }
    if (on_device) {
        if (n_bvh < d->n_tri) {
            hipError_t e = hipMemcpy(s->tris.p + 3 * loose_first, &bvh.tris[12 * loose_first], 48 * loose_src.size(), hipMemcpyHostToDevice);
            if (e != hipSuccess) return cleanup_fail(e, "loose triangle records");
        }
    } else {
        UP(s->nodes, bvh.nodes.data(), bvh.nodes.size() / 4, float4);
        UP(s->cnodes, bvh.cnodes.data(), bvh.cnodes.size() / 4, uint4);
        UP(s->tris, bvh.tris.data(), bvh.tris.size() / 4, float4);
    }
    for (int a = 0; a < 3; a++) { s->dev.q_lo[a] = bvh.q_lo[a]; s->dev.q_scale[a] = bvh.q_scale[a]; }
    s->dev.cnodes = s->cnodes.p;
    s->dev.n_cnodes = on_device ? n_cnodes : bvh.n_nodes;
    {
        hipError_t e = s->tri_nrm.ensure((size_t)std::max(d->n_tri, 1));
        if (e == hipSuccess) {
            launch_tri_normals(ctx->stream, s->tris.p, d->n_tri, s->tri_nrm.p);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return cleanup_fail(e, "triangle normals");
        s->dev.tri_nrm = s->tri_nrm.p;
    }
    s->info.bvh_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build).count();
    s->info.bvh_on_device = on_device ? 1 : 0;
    s->info.n_nodes = bvh.n_nodes; s->info.n_orders = bvh.n_orders; s->info.n_top = bvh.n_top; s->info.n_triangles = d->n_tri;
    s->info.n_loose = d->n_tri - n_bvh;
    UP(s->sph, d->sph, d->n_sph, float4);
    std::vector<int2> info(d->n_sph);
    for (int i = 0; i < d->n_sph; i++) info[i] = make_int2(d->sph_mat[i], sph_prim[i]);
    UP(s->sph_info, info.data(), d->n_sph, int2);
    UP(s->cyl, d->cyl, 5 * (size_t)d->n_cyl, float);
    info.resize(d->n_cyl);
    for (int i = 0; i < d->n_cyl; i++) info[i] = make_int2(d->cyl_mat[i], cyl_prim[i]);
    UP(s->cyl_info, info.data(), d->n_cyl, int2);
    UP(s->mats, d->mats, d->n_mat, rt_material);
    UP(s->lights, lights.data(), d->n_light, DevLight);
    UP(s->ltris, d->light_tri_v, 9 * (size_t)d->n_light_tri, float);

    s->dev.nodes = s->nodes.p; s->dev.tris = s->tris.p; s->dev.sph = s->sph.p; s->dev.sph_info = s->sph_info.p;
    s->dev.cyl = s->cyl.p; s->dev.cyl_info = s->cyl_info.p; s->dev.mats = s->mats.p; s->dev.lights = s->lights.p; s->dev.ltris = s->ltris.p;
    s->dev.n_nodes = bvh.n_nodes; s->dev.n_top = bvh.n_top; s->dev.n_orders = bvh.n_orders; s->dev.n_tri = d->n_tri; s->dev.n_sph = d->n_sph; s->dev.n_cyl = d->n_cyl;
    s->dev.n_light = d->n_light;
    s->dev.n_mat = d->n_mat; s->dev.n_light_tri = d->n_light_tri;
    {
        // the loose list: the records appended above, each followed by its index into the triangle array
        std::vector<std::pair<int, int>> by_prim; // (prim id, index into tris)
        if (use_loose)
            for (size_t j = 0; j < loose_src.size(); j++) by_prim.emplace_back(tri_prim[(size_t)loose_src[j]], (int)(loose_first + j));
        std::vector<float> loose;
        s->dev.n_loose = -1;
        if (use_loose) {
            s->dev.n_loose = (int32_t)by_prim.size();
            // Scale of everything a ray's end points can be: rays start on surfaces and end on lights.
            double scale = 0.0;
            auto grow = [&](double v) { scale = std::max(scale, std::fabs(v)); };
            for (size_t i = 0; i < 9 * (size_t)d->n_tri; i++) grow(d->tri_v[i]);
            for (int i = 0; i < d->n_sph; i++) for (int a = 0; a < 3; a++) grow(std::fabs(d->sph[4 * i + a]) + std::fabs(d->sph[4 * i + 3]));
            for (int i = 0; i < d->n_cyl; i++) { grow(std::fabs(d->cyl[5 * i]) + std::fabs(d->cyl[5 * i + 4])); grow(std::fabs(d->cyl[5 * i + 1]) + std::fabs(d->cyl[5 * i + 4])); grow(d->cyl[5 * i + 2]); grow(d->cyl[5 * i + 3]); }
            for (int i = 0; i < d->n_light; i++) for (int a = 0; a < 3; a++) grow(std::fabs(d->lights[i].origin[a]) + std::fabs(d->lights[i].radius));
            for (size_t i = 0; i < 9 * (size_t)d->n_light_tri; i++) grow(d->light_tri_v[i]);
            std::vector<float> planes; // behind the triangle records, padded to a multiple of four (loose_traverse loads four at a time)
            for (const auto &pi : by_prim) {
                const float *rec = &bvh.tris[12 * (size_t)pi.second];
                loose.insert(loose.end(), rec, rec + 12);
                std::memcpy(&loose[loose.size() - 1], &pi.second, sizeof(int32_t));
                // The triangle's plane, scaled so that "farther than 1" means "safely off the plane" — an
                // occlusion query whose segment starts behind it or ends in front of it by that margin cannot hit the
                // front face, whatever the rounding of the exact test (loose_traverse).  Margin: 2e-5 of the scene's
                // scale, more for slivers (the exact test's error grows with |E1||E2| / |E1 x E2|).
                const double e1[3] = {rec[4], rec[5], rec[6]}, e2[3] = {rec[8], rec[9], rec[10]};
                const double nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
                const double nl = std::sqrt(nx * nx + ny * ny + nz * nz);
                const double l1 = std::sqrt(e1[0] * e1[0] + e1[1] * e1[1] + e1[2] * e1[2]), l2 = std::sqrt(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
                float plane[4] = {0.0f, 0.0f, 0.0f, 0.0f}; // degenerate triangle: never "safely off" — the exact test decides
                if (nl > 0.0 && scale > 0.0 && std::isfinite(nl) && std::isfinite(scale)) {
                    const double sliver = l1 * l2 / nl;
                    const double margin = 2e-5 * scale * std::max(1.0, sliver / 10.0); // the exact test's own error: about 1.5e-6 * scale
                    const double k = 1.0 / (nl * margin);
                    plane[0] = (float)(nx * k); plane[1] = (float)(ny * k); plane[2] = (float)(nz * k);
                    plane[3] = (float)((nx * rec[0] + ny * rec[1] + nz * rec[2]) * k);
                }
                planes.insert(planes.end(), plane, plane + 4);
            }
            const float nothing_in_front[4] = {0.0f, 0.0f, 0.0f, 2.0f}; // signed distance -2 for every point: always skipped
            while ((planes.size() / 4) % 4 != 0) planes.insert(planes.end(), nothing_in_front, nothing_in_front + 4);
            loose.insert(loose.end(), planes.begin(), planes.end());
        }
        UP(s->loose, loose.data(), loose.size() / 4, float4);
        s->dev.loose = s->loose.p;
        // boxes of the loose triangles for the closest-hit scan, padded like the leaves of the tree; triangles with the same
        // box (the two halves of a quad) share one (RT_LOOSE_BOXES=0: every triangle is tested exactly, one after the other)
        std::vector<float> lboxes;
        if (use_loose && by_prim.size() <= 32 && env_u32_allow_zero("RT_LOOSE_BOXES", 1) != 0) {
            std::vector<std::array<float, 6>> boxes;
            std::vector<uint32_t> masks;
            for (size_t j = 0; j < by_prim.size(); j++) {
                const float *rec = &bvh.tris[12 * (size_t)by_prim[j].second];
                const float v[3][3] = {{rec[0], rec[1], rec[2]}, {rec[0] + rec[4], rec[1] + rec[5], rec[2] + rec[6]}, {rec[0] + rec[8], rec[1] + rec[9], rec[2] + rec[10]}};
                std::array<float, 6> bx;
                float ext = 0.0f, mag = 0.0f;
                for (int a = 0; a < 3; a++) {
                    bx[(size_t)a] = std::min(std::min(v[0][a], v[1][a]), v[2][a]);
                    bx[(size_t)a + 3] = std::max(std::max(v[0][a], v[1][a]), v[2][a]);
                    ext = std::max(ext, bx[(size_t)a + 3] - bx[(size_t)a]);
                    mag = std::max(mag, std::max(std::fabs(bx[(size_t)a]), std::fabs(bx[(size_t)a + 3])));
                }
                // the record holds v1 - v0 and v2 - v0, so v1 and v2 above carry a rounding error: a wider pad than the tree's
                const float pad = 2e-4f * ext + 2e-5f * mag + 2e-6f;
                for (int a = 0; a < 3; a++) { bx[(size_t)a] -= pad; bx[(size_t)a + 3] += pad; }
                size_t g = 0;
                for (; g < boxes.size(); g++) {
                    bool same = true;
                    for (size_t a = 0; a < 6; a++) same = same && std::fabs(boxes[g][a] - bx[a]) <= pad;
                    if (same) break;
                }
                if (g == boxes.size()) { boxes.push_back(bx); masks.push_back(0u); }
                else for (size_t a = 0; a < 3; a++) { boxes[g][a] = std::min(boxes[g][a], bx[a]); boxes[g][a + 3] = std::max(boxes[g][a + 3], bx[a + 3]); }
                masks[g] |= 1u << j;
            }
            for (size_t g = 0; g < boxes.size(); g++) {
                float mbits;
                std::memcpy(&mbits, &masks[g], 4);
                const float rec[8] = {boxes[g][0], boxes[g][1], boxes[g][2], mbits, boxes[g][3], boxes[g][4], boxes[g][5], 0.0f};
                lboxes.insert(lboxes.end(), rec, rec + 8);
            }
        }
        UP(s->loose_boxes, lboxes.data(), lboxes.size() / 4, float4);
        s->dev.loose_boxes = s->loose_boxes.p;
        s->dev.n_loose_boxes = (int32_t)(lboxes.size() / 8);
        // primitive id -> triangle record, for the walk that merges hits by (t, primitive id)
        std::vector<int32_t> tri_of_prim((size_t)std::max(prim, 1), -1);
        for (int i = 0; i < d->n_tri; i++) {
            int32_t pid;
            std::memcpy(&pid, &bvh.tris[12 * (size_t)i + 3], sizeof(pid));
            tri_of_prim[(size_t)pid] = i;
        }
        UP(s->tri_of_prim, tri_of_prim.data(), tri_of_prim.size(), int32_t);
        s->dev.tri_of_prim = s->tri_of_prim.p;
        s->dev.deferred = env_u32_allow_zero("RT_DEFERRED", 1) != 0 ? 1 : 0;
        s->dev.walk_span = (int32_t)std::min<uint32_t>(64u, env_u32("RT_WALK_SPAN", 8));
        s->dev.walk_refill = (int32_t)std::min<uint32_t>(64u, env_u32("RT_WALK_REFILL", 48));
        s->dev.walk_pairs = env_u32_allow_zero("RT_WALK_PAIRS", 1) != 0 ? 1 : 0;
        s->dev.walk_batch = (int32_t)std::min<uint32_t>(64u, env_u32("RT_WALK_BATCH", 64));
        s->dev.walk_batch_any = (int32_t)std::min<uint32_t>(64u, env_u32("RT_WALK_BATCH_ANY", 64));
        s->dev.inline_shadow = (int32_t)env_u32_allow_zero("RT_INLINE_SHADOW", 1); // 0 never, 1 when it pays (see the render loop), 2 always
    }
#undef UP
    for (const DevLight &L : lights) { s->n_disk_lights += L.kind == RT_LIGHT_DISK; s->n_point_lights += L.kind == RT_LIGHT_POINT; }
    *out = s;
    return RT_OK;
}

int32_t rt_scene_destroy(rt_scene *s) {
    if (!s) return RT_OK;
    (void)hipSetDevice(s->ctx->device);
    (void)hipStreamSynchronize(s->ctx->stream);
    s->loose_boxes.release();
    s->nodes.release(); s->cnodes.release(); s->tris.release(); s->sph.release(); s->sph_info.release(); s->cyl.release(); s->cyl_info.release();
    s->tri_nrm.release(); s->mats.release(); s->lights.release(); s->ltris.release(); s->loose.release(); s->tri_of_prim.release();
    delete s;
    return RT_OK;
}

// The render loop shared by rt_render and rt_render_static.  `pool` = the device memory to work in (already large
// enough for the given queue geometry), `accum_dev` = device sum buffer the planes are added into at the end.
// Allocates no device memory unless RT_FLAG_TIMING asks for events.
static int32_t render_in_pool(rt_scene *s, const rt_camera *cam, const rt_params *p, const RenderPool &pool, uint32_t seg_capacity,
                              uint32_t shadow_seg_capacity, uint32_t target, float *accum_dev, rt_stats *stats_out,
                              std::chrono::steady_clock::time_point t0, bool fixed_pool) {
    rt_context *ctx = s->ctx;
    // an error inside the loop leaves with both streams idle: the occlusion kernel on the second stream works in the pool
    // (for rt_render_static: the array every context of the GPU shares) and the next call starts by clearing it
    auto loop_fail = [&](int code, const std::string &msg) {
        (void)hipStreamSynchronize(ctx->stream2);
        (void)hipStreamSynchronize(ctx->stream);
        return fail(code, msg);
    };
    const char *overflow_msg = fixed_pool ? "path queue overflow: the fixed queues of rt_render_static are too small for this frame, use rt_render"
                                          : "path queue overflow: raise RT_QUEUE_TARGET";
    const size_t n_pixels = (size_t)p->width * p->height;
    const RenderConst rcst = make_const(*cam, *p);
    const int n_samples = p->spp_end - p->spp_begin;
    const size_t plane_stride = rcst.plane_stride; // the kernels index the colour planes by tile

    // 8x8 pixel blocks to render: all of them, or those touching this shard's tiles
    const uint32_t blocks_w = (uint32_t)((p->width + 7) / 8), blocks_h = (uint32_t)((p->height + 7) / 8);
    uint32_t n_blocks = blocks_w * blocks_h;
    const int32_t *blocks_dev = nullptr;
    if (p->tile_size > 0) {
        std::vector<int32_t> sel;
        for (uint32_t by = 0; by < blocks_h; by++)
            for (uint32_t bx = 0; bx < blocks_w; bx++) {
                bool any = false;
                for (int yy = 0; yy < 8 && !any; yy++)
                    for (int xx = 0; xx < 8 && !any; xx++) {
                        int x = (int)bx * 8 + xx, y = (int)by * 8 + yy;
                        if (x < p->width && y < p->height && pixel_selected_host(*p, x, y)) any = true;
                    }
                if (any) sel.push_back((int32_t)(by * blocks_w + bx));
            }
        n_blocks = (uint32_t)sel.size();
        if (!sel.empty()) HIP_TRY(hipMemcpyAsync(pool.blocks, sel.data(), sel.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream)); // sel goes out of scope
        blocks_dev = pool.blocks;
    }
    const unsigned long long total = (unsigned long long)n_blocks * 64ull * (unsigned long long)std::max(0, n_samples);

    LaunchCfg cfg{ctx->stream, seg_capacity, shadow_seg_capacity, target};
    HIP_TRY(hipMemsetAsync(pool.planes, 0, 3 * plane_stride * sizeof(float), ctx->stream));
    rt_stats st;
    std::memset(&st, 0, sizeof(st));
    StageTimer timer{ctx, (p->flags & RT_FLAG_TIMING) != 0};
    const bool count_work = (p->flags & RT_FLAG_COUNT_WORK) != 0;

    if (p->shader == RT_SHADER_NORMAL) {
        timer.begin(1);
        launch_normal_shader(cfg, s->dev, rcst, pool.planes, (uint32_t)n_pixels);
        timer.end();
        st.launches_shade = 1;
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        timer.collect();
    } else if (total > 0) {
        HIP_TRY(hipMemsetAsync(pool.ctl, 0, sizeof(Ctl), ctx->stream));
        std::memset(ctx->ctl_host, 0, sizeof(CtlHead));
        ctx->ctl_host->total = total;
        ctx->ctl_host->per_sample = n_blocks * 64u;
        ctx->ctl_host->n_samples = (uint32_t)n_samples;
        // samples of one 8x8 block generated together; bounded so that a chunk's candidate count stays below 2^30
        uint32_t chunk_samples = env_u32("RT_TILE_SAMPLES", 16);
        while (chunk_samples > 1 && (unsigned long long)n_blocks * 64ull * chunk_samples >= (1ull << 30)) chunk_samples /= 2;
        ctx->ctl_host->chunk_samples = chunk_samples;
        // Path nodes carry the light gathered so far and add it to the pixel once, at the path's end (RT_CARRY=0: every
        // contribution is added where it arises).  Not for the Whitted shader: all its light arrives through queued shadow rays.
        // It pays where most occlusion queries are answered by the shading kernel (their contributions then never
        // become atomic adds) and costs 16 bytes of traffic per node where they are not, so it follows the same rule as
        // the in-place queries, decided once: the first iteration's children carry, the second iteration sees how the
        // rays split into light and heavy and either goes on or adds what the nodes brought and stops carrying.
        const uint32_t carry_mode = p->shader == RT_SHADER_WHITTED || s->dev.n_loose < 0 ? 0u : env_u32_allow_zero("RT_CARRY", 1); // 0 never, 1 by the rule, 2 always
        bool carry_in = false, carry_out = carry_mode != 0;
        int carry_decided = carry_mode == 1 ? 0 : 1;
        HIP_TRY(hipMemcpyAsync(&pool.ctl->h, ctx->ctl_host, sizeof(CtlHead), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        int cur = 0, scur = 0;
        // The occlusion queries an iteration's shading queues are traced during the NEXT iteration, on a second stream
        // beside that iteration's ray generation, extend and shade (which fill the other shadow buffer).
        const bool overlap = env_u32_allow_zero("RT_OVERLAP", 1) != 0;
        hipStream_t shadow_stream = overlap ? ctx->stream2 : ctx->stream;
        bool shadow_pending = false;
        const bool plan_poll = ctx->ctl_host_dev != nullptr && env_u32_allow_zero("RT_PLAN_POLL", 1) != 0; // 0: copy + stream synchronisation
        uint32_t plan_seq = 0; // runs of k_plan so far (the head was uploaded with seq = 0)
        // The main stream's ray generation and extend kernels are submitted before the occlusion kernel of the second
        // stream: they are the critical path (the occlusion kernel only has to be through before the next k_plan), and
        // whichever kernel is submitted first gets the idle GPU.  Measured 1-3 % per frame on the Utah, Miku and composite
        // frames against the other order (RT_SHADOW_LATE=0).
        const bool shadow_late = overlap && env_u32_allow_zero("RT_SHADOW_LATE", 1) != 0;
        const bool dump_iter = std::getenv("RT_DUMP_ITER") != nullptr;
        // paths left when the staged iterations hand over to k_tail (RT_TAIL, 0 = never); not for the Whitted shader
        // (several shadow rays per node) nor for work-counting calls
        const uint32_t tail_limit = (p->shader == RT_SHADER_WHITTED || count_work) ? 0u : env_u32_allow_zero("RT_TAIL", 1u << 18);
        // Whether the tail kernel pays depends on the scene: a thread that walks big meshes alone does it at a few lanes
        // per wave, and where such paths dominate the staged kernels finish the call sooner (Miku at 128 spp: 270 ms with the
        // tail, 254 without; Cornell at 16 spp: 10.1 with, 13.4 without).  So it is measured: the first call on a scene uses
        // the tail and notes what it cost per bounce still to go, and what the last staged iteration before it cost; later
        // calls use the tail only where it was the cheaper one (RT_TAIL_ADAPT=0: always).  Results do not depend on it.
        const bool tail_adapt = env_u32_allow_zero("RT_TAIL_ADAPT", 1) != 0;
        const int tail_key = p->shader & 7;
        bool tail_pays = true;
        if (tail_adapt && s->tail_ms_per_bounce[tail_key] >= 0.0 && s->staged_ms_per_bounce[tail_key] >= 0.0)
            tail_pays = s->tail_ms_per_bounce[tail_key] < s->staged_ms_per_bounce[tail_key];
        auto t_readback = std::chrono::steady_clock::now(); // wall clock of the previous read-back
        double last_staged_ms = -1.0;                        // duration of the last staged iteration without new camera rays
        int drain_iters = 0;                                 // iterations since the last camera ray was added
        int tail_bounces_left = 0;                           // > 0: the previous iteration was the tail kernel's first launch
        bool prev_staged_drain = false;
        for (;;) {
            if (shadow_pending && overlap) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->shadow_done, 0)); // k_plan rewrites the counts k_shadow reads
            shadow_pending = false;
            if (plan_poll) {
                // k_plan writes the head into the pinned host copy itself, the sequence number last: no copy command, no
                // wake-up through the runtime.  The stream is looked at now and then so that a failed launch ends the wait.
                launch_plan(cfg, pool.ctl, ctx->ctl_host_dev);
                plan_seq++;
                for (uint64_t spins = 1;; spins++) {
                    if (__atomic_load_n(&ctx->ctl_host->seq, __ATOMIC_ACQUIRE) == plan_seq) break;
                    if ((spins & 0xFFFu) == 0u) {
                        const hipError_t q = hipStreamQuery(ctx->stream);
                        if (q == hipErrorNotReady) continue;
                        if (q == hipSuccess && __atomic_load_n(&ctx->ctl_host->seq, __ATOMIC_ACQUIRE) == plan_seq) break;
                        return loop_fail(RT_ERR_DEVICE, std::string("k_plan did not report: ") + hipGetErrorString(q == hipSuccess ? hipErrorUnknown : q));
                    }
                    __builtin_ia32_pause();
                }
            } else {
                launch_plan(cfg, pool.ctl);
                HIP_TRY(hipMemcpyAsync(ctx->ctl_host, &pool.ctl->h, sizeof(CtlHead), hipMemcpyDeviceToHost, ctx->stream));
                HIP_TRY(hipStreamSynchronize(ctx->stream));
            }
            timer.collect();
            const CtlHead c = *ctx->ctl_host;
            if (c.overflow) return loop_fail(RT_ERR_NOMEM, overflow_msg);
            {
                const auto now = std::chrono::steady_clock::now();
                const double ms = std::chrono::duration<double, std::milli>(now - t_readback).count();
                t_readback = now;
                if (prev_staged_drain) last_staged_ms = ms;
                if (tail_bounces_left > 0) { // the iteration that just ended was the tail kernel
                    s->tail_ms_per_bounce[tail_key] = ms / tail_bounces_left;
                    if (last_staged_ms >= 0.0) s->staged_ms_per_bounce[tail_key] = last_staged_ms;
                    tail_bounces_left = 0;
                }
                prev_staged_drain = false;
            }
            if (dump_iter) // development aid: the population and the wall clock at every iteration
                std::fprintf(stderr, "[rt] %8.3f ms: %u nodes carried in, %u camera rays to add, shadow rays <= %u + %u per segment\n",
                             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), c.n_alive, c.n_new, c.max_sh_light,
                             c.max_sh_heavy);
            // occlusion queries queued by the previous iteration's shading, now with exact per-class counts
            bool shadow_launched = false;
            auto launch_pending_shadow = [&]() -> int {
                shadow_launched = true;
                if (!(c.max_sh_light || c.max_sh_heavy)) return RT_OK;
                ShadowQueue filled{pool.sa[scur ^ 1], pool.sb[scur ^ 1], pool.sc2[scur ^ 1]};
                LaunchCfg scfg = cfg;
                scfg.stream = shadow_stream;
                timer.begin(2, shadow_stream);
                launch_shadow(scfg, s->dev, rcst, pool.ctl, filled, pool.planes, (uint32_t)plane_stride, std::min(c.max_sh_light, shadow_seg_capacity),
                              std::min(c.max_sh_heavy, shadow_seg_capacity), count_work);
                timer.end();
                if (overlap) HIP_TRY(hipEventRecord(ctx->shadow_done, shadow_stream));
                shadow_pending = true;
                st.launches_shadow++;
                return RT_OK;
            };
            if (!shadow_late || c.done) {
                const int src = launch_pending_shadow();
                if (src) return loop_fail(src, rt_last_error());
            }
            if (c.done) break;
            NodeQueue in{pool.qa[cur], pool.qb[cur], pool.qc[cur], pool.ql[cur]};
            NodeQueue out{pool.qa[cur ^ 1], pool.qb[cur ^ 1], pool.qc[cur ^ 1], pool.ql[cur ^ 1]};
            ShadowQueue sh{pool.sa[scur], pool.sb[scur], pool.sc2[scur]};
            if (c.n_new) {
                timer.begin(3);
                launch_raygen(cfg, s->dev, rcst, pool.ctl, in, blocks_dev, blocks_w, c.n_new);
                timer.end();
                st.launches_raygen++;
            }
            // most nodes a segment of each class can hold after ray generation: what it carried in + its share of the
            // raygen workgroups (workgroup b feeds segment b % RT_NSEG_HALF of either class)
            const uint32_t new_blocks = (c.n_new + RT_RAYGEN_PER_WG - 1u) / RT_RAYGEN_PER_WG;
            const uint32_t fresh = ((new_blocks + RT_NSEG_HALF - 1) / RT_NSEG_HALF) * RT_RAYGEN_PER_WG;
            const uint32_t light_threads = std::min(seg_capacity, c.max_in_light + fresh);
            const uint32_t heavy_threads = s->dev.n_heavy ? std::min(seg_capacity, c.max_in_heavy + fresh) : c.max_in_heavy;
            const bool draining = c.n_new == 0 && c.cursor == c.total;
            if (draining) drain_iters++;
            if (tail_limit && tail_pays && draining && c.n_alive <= tail_limit) {
                if (!shadow_launched) {
                    const int src = launch_pending_shadow();
                    if (src) return loop_fail(src, rt_last_error());
                }
                // the last few paths: one launch follows each of them to its end (see k_tail)
                if (last_staged_ms >= 0.0 && tail_bounces_left == 0 && s->tail_ms_per_bounce[tail_key] < 0.0)
                    tail_bounces_left = std::max(1, p->max_depth + 1 - drain_iters); // (an upper bound: the youngest path left was born drain_iters ago)
                timer.begin(0);
                DevScene dev_tail = s->dev;
                if (carry_in) st.nodes_carrying += c.n_alive;
                dev_tail.carry_in = carry_in ? 1 : 0;
                dev_tail.carry_out = carry_in ? 1 : 0; // (a second child that overflows the thread's stack goes to the next queue)
                carry_out = carry_in;
                launch_tail(cfg, p->shader, dev_tail, rcst, pool.ctl, in, out, pool.planes, (uint32_t)plane_stride, light_threads, heavy_threads);
                timer.end();
                st.launches_extend++;
                hipError_t le = hipGetLastError();
                if (le != hipSuccess) return loop_fail(RT_ERR_DEVICE, std::string("kernel launch: ") + hipGetErrorString(le));
                cur ^= 1;
                scur ^= 1;
                continue;
            }
            prev_staged_drain = draining && c.n_alive <= 4u * std::max(tail_limit, 1u);
            timer.begin(0);
            launch_extend(cfg, s->dev, rcst, pool.ctl, in, pool.hit, light_threads, heavy_threads, count_work);
            timer.end();
            st.launches_extend++;
            if (!shadow_launched) {
                const int src = launch_pending_shadow();
                if (src) return loop_fail(src, rt_last_error());
            }
            // Light occlusion queries answered by the shading kernel itself: measured -5 % per frame when few rays are heavy
            // (Utah teapot: 5 %), +1 to +2.5 % when none are (the occlusion kernel then overlaps with everything on the
            // second stream) or when many are (Miku, composite).  The choice only moves work: results are the same.
            DevScene dev_now = s->dev;
            const bool few_heavy = c.max_in_heavy > 0 && 4u * c.max_in_heavy < c.max_in_light;
            if (!carry_decided && c.n_alive > 0) { // the first iteration that carries nodes in: they are the first children
                carry_out = s->dev.inline_shadow >= 2 || (s->dev.inline_shadow == 1 && few_heavy);
                carry_decided = 1;
            }
            dev_now.carry_in = carry_in ? 1 : 0;
            dev_now.carry_out = carry_out ? 1 : 0;
            if (carry_in) st.nodes_carrying += c.n_alive;
            dev_now.inline_shadow = s->dev.inline_shadow >= 2 || (s->dev.inline_shadow == 1 && c.max_in_heavy > 0 && 4u * c.max_in_heavy < c.max_in_light) ? 1 : 0;
            timer.begin(1);
            launch_shade(cfg, p->shader, dev_now, rcst, pool.ctl, in, out, pool.hit, sh, pool.planes, (uint32_t)plane_stride, light_threads, heavy_threads);
            timer.end();
            st.launches_shade++;
            carry_in = carry_out; // the queue just written is the one read next
            hipError_t le = hipGetLastError();
            if (le != hipSuccess) return loop_fail(RT_ERR_DEVICE, std::string("kernel launch: ") + hipGetErrorString(le));
            cur ^= 1;
            scur ^= 1;
        }
        if (overlap) HIP_TRY(hipStreamSynchronize(ctx->stream2));
        // the last occlusion launch and the tail kernel add to the counters after the loop's last read-back
        HIP_TRY(hipMemcpyAsync(ctx->ctl_host, &pool.ctl->h, sizeof(CtlHead), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        timer.collect();
        const CtlHead c = *ctx->ctl_host;
        if (c.overflow) return fail(RT_ERR_NOMEM, overflow_msg);
        st.camera_samples = c.cam_samples;
        st.rays_closest = c.rays_closest;
        st.rays_shadow = c.rays_shadow;
        st.rays_shadow_inline = c.rays_shadow_inline;
        st.nodes_visited = c.nodes + c.nodes_sh;
        st.tris_tested = c.tris + c.tris_sh + c.scanned;
        st.nodes_visited_shadow = c.nodes_sh;
        st.tris_tested_shadow = c.tris_sh;
        st.tris_scanned = c.scanned;
        if ((p->flags & RT_FLAG_COUNT_WORK) && getenv("RT_DUMP_HIST")) { // development aid: distribution of closest-hit walk lengths
            std::fprintf(stderr, "[rt] closest-hit walks: bucket b = visited nodes in [2^(b-1), 2^b)\n[rt] rays :");
            for (int b = 0; b < 16; b++) std::fprintf(stderr, " %llu", c.hist_ray[b]);
            std::fprintf(stderr, "\n[rt] waves:");
            for (int b = 0; b < 16; b++) std::fprintf(stderr, " %llu", c.hist_wave[b]);
            std::fprintf(stderr, "\n[rt] sum of per-wave longest walk %llu (x64 lanes = %llu lane-steps paid for, %llu used by closest-hit and shadow rays)\n",
                         c.wave_steps, c.wave_steps * 64ull, c.nodes);
            std::fprintf(stderr, "[rt] closest-hit triangle-loop iterations run by waves %llu (x64 = %llu lane-tests paid for; %llu tests used by both kinds)\n",
                         c.wave_tri_iters, c.wave_tri_iters * 64ull, c.tris);
        }
        st.iterations = c.iterations;
    }

    // add the planes into the caller's interleaved sum buffer
    launch_planes_to_rgb(cfg, rcst, pool.planes, accum_dev, (uint32_t)n_pixels);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(RT_ERR_DEVICE, std::string("render: ") + hipGetErrorString(le));

    st.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    st.ms_extend = timer.ms[0]; st.ms_shade = timer.ms[1]; st.ms_shadow = timer.ms[2]; st.ms_raygen = timer.ms[3];
    if (p->shader == RT_SHADER_NORMAL) {
        st.camera_samples = 0; // not tracked by the single-kernel normal shader
    }
    if (stats_out) *stats_out = st;
    return RT_OK;
}

// shadow rays one shaded node can emit: 1 (next-event estimation), or every light sample of LightList::visible for the
// Whitted shader (light.rs:368-385, 150-181)
static uint32_t shadow_rays_per_node(const rt_scene *s, const rt_params *p) {
    if (p->shader != RT_SHADER_WHITTED) return 1;
    return (uint32_t)std::max(1, s->n_point_lights + s->n_disk_lights * std::max(1, p->light_samples));
}

int32_t rt_render(rt_scene *s, const rt_camera *cam, const rt_params *p, float *accum_rgb, int32_t accum_is_device, rt_stats *stats) {
    if (!s || !cam || !p || !accum_rgb) return fail(RT_ERR_INVALID, "null argument");
    int rc = check_params(*p);
    if (rc) return rc;
    rc = check_supported(*s, *p);
    if (rc) return rc;
    rt_context *ctx = s->ctx;
    rc = set_device(ctx);
    if (rc) return rc;
    auto t0 = std::chrono::steady_clock::now();

    const size_t n_pixels = (size_t)p->width * p->height;
    const size_t plane_stride = 64u * (size_t)((p->width + 7) / 8) * (size_t)((p->height + 7) / 8);
    const unsigned long long total = (unsigned long long)plane_stride * (unsigned long long)std::max(0, p->spp_end - p->spp_begin);

    // queue geometry: RT_NSEG segments per queue, sized from the refill level `target`.  The default keeps 32 Mi rays in
    // flight (measured: 8 Mi -> 32 Mi is 3-8 % faster: fewer iterations, and the few long walks of a launch overlap with
    // more short ones; 64 Mi gives another 2-3 % but doubles the footprint to about 100 GB);
    // that is about 50 GB of queues, so without an explicit RT_QUEUE_TARGET the level is halved until the queues fit
    // into half of the free device memory.
    const char *target_env = std::getenv("RT_QUEUE_TARGET");
    const bool target_given = target_env && *target_env;
    uint32_t target = env_u32("RT_QUEUE_TARGET", 1u << 25);
    const uint32_t per_node = shadow_rays_per_node(s, p);
    if (per_node > 1) target = std::max<uint32_t>(1u << 16, target / 8);
    if (total < target) target = (uint32_t)total;
    uint32_t seg_capacity = 0, shadow_seg_capacity = 0;
    for (;;) {
        target = std::max<uint32_t>((target + 255u) & ~255u, 1u << 14);
        // a segment holds 8x its even share of the refill level: bursts of branching (Microfacet spawns two rays per hit)
        // and the uneven split between light and heavy segments must fit; running out of room is reported, never silent
        seg_capacity = 8u * (target / RT_NSEG);
        shadow_seg_capacity = (per_node > 1 ? 4u * (target / RT_NSEG) * per_node : seg_capacity);
        const size_t need = (size_t)seg_capacity * RT_NSEG * (2 * 64 + 8) + (size_t)shadow_seg_capacity * RT_NSEG * (2 * 40);
        const bool grows = (size_t)seg_capacity * RT_NSEG > ctx->capacity || (size_t)shadow_seg_capacity * RT_NSEG > ctx->shadow_capacity;
        if (target_given || !grows || target <= (1u << 20)) break;
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) break;
        if (const uint32_t pretend_mb = env_u32("RT_FREE_MEM_MB", 0)) free_b = (size_t)pretend_mb << 20; // for tests of this fallback
        const size_t held = ctx->capacity * (size_t)(2 * 64 + 8) + ctx->shadow_capacity * (size_t)(2 * 40); // freed and re-allocated on growth
        if (need <= (free_b + held) / 2) break;
        target /= 2;
    }
    rc = ensure_queues(ctx, seg_capacity * RT_NSEG, shadow_seg_capacity * RT_NSEG, plane_stride);
    if (rc) return rc;
    if (p->tile_size > 0) HIP_TRY_ALLOC(ctx->blocks.ensure(std::max<size_t>(1, plane_stride / 64)));
    float *accum_dev = accum_rgb;
    if (!accum_is_device) {
        HIP_TRY_ALLOC(ctx->accum_tmp.ensure(3 * n_pixels));
        HIP_TRY(hipMemcpyAsync(ctx->accum_tmp.p, accum_rgb, 3 * n_pixels * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        accum_dev = ctx->accum_tmp.p;
    }
    RenderPool pool;
    for (int k = 0; k < 2; k++) {
        pool.qa[k] = ctx->qa[k].p; pool.qb[k] = ctx->qb[k].p; pool.qc[k] = ctx->qc[k].p; pool.ql[k] = ctx->ql[k].p;
        pool.sa[k] = ctx->sa[k].p; pool.sb[k] = ctx->sb[k].p; pool.sc2[k] = ctx->sc2[k].p;
    }
    pool.hit = ctx->hit.p; pool.planes = ctx->planes.p; pool.blocks = ctx->blocks.p; pool.ctl = ctx->ctl;
    rt_stats st;
    rc = render_in_pool(s, cam, p, pool, seg_capacity, shadow_seg_capacity, target, accum_dev, &st, t0, false);
    if (rc) return rc;
    if (!accum_is_device) {
        HIP_TRY(hipMemcpyAsync(accum_rgb, ctx->accum_tmp.p, 3 * n_pixels * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        st.ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    st.device_bytes = (uint64_t)ctx->capacity * (2 * 64 + 8) + (uint64_t)ctx->shadow_capacity * (2 * 40) + (uint64_t)ctx->planes.n * sizeof(float) +
                      (uint64_t)ctx->accum_tmp.n * sizeof(float) + (uint64_t)ctx->blocks.n * sizeof(int32_t) + sizeof(Ctl);
    if (stats) *stats = st;
    return RT_OK;
}

// rt_render for callers that cannot allocate: device sum buffer only, everything else lives in this device's copy of
// the fixed array (ctx->scratch_ptr).  No hipMalloc / hipFree / event creation on this path.  The array is shared by
// every call on the GPU and keeps the last call's contents: the loop state and the planes are cleared at the start
// of every call (render_in_pool), queue slots are written before they are read.  Calls that overlap on one GPU race.
int32_t rt_render_static(rt_scene *s, const rt_camera *cam, const rt_params *p, float *accum_rgb_dev, rt_stats *stats) {
    if (!s || !cam || !p || !accum_rgb_dev) return fail(RT_ERR_INVALID, "null argument");
    int rc = check_params(*p);
    if (rc) return rc;
    rc = check_supported(*s, *p);
    if (rc) return rc;
    rt_context *ctx = s->ctx;
    if (!ctx->scratch_ptr) return fail(RT_ERR_DEVICE, "rt_render_static: the fixed array was not found on this device");
    if (p->flags & RT_FLAG_TIMING) return fail(RT_ERR_INVALID, "rt_render_static: RT_FLAG_TIMING creates events; not available on the no-allocation path");
    const size_t plane_stride = 64u * (size_t)((p->width + 7) / 8) * (size_t)((p->height + 7) / 8);
    if (plane_stride > RT_STATIC_MAX_PIXELS)
        return fail(RT_ERR_INVALID, "rt_render_static: frame too large: at most 2^21 pixels after padding width and height to multiples of 8 (use rt_render)");
    // the refill level the fixed segments can take: a shaded node of the Whitted shader emits per_node shadow rays
    const uint32_t per_node = shadow_rays_per_node(s, p);
    uint32_t target = RT_STATIC_TARGET;
    if (per_node > 1) target = std::min<uint32_t>(target, ((RT_STATIC_SEG_CAPACITY / (4u * per_node)) * RT_NSEG) & ~255u);
    if (target < 2048u) return fail(RT_ERR_INVALID, "rt_render_static: too many shadow rays per surface point for the fixed queues (use rt_render)");
    rc = set_device(ctx);
    if (rc) return rc;
    auto t0 = std::chrono::steady_clock::now();
    unsigned char *base = ctx->scratch_ptr;
    RenderPool pool;
    pool.ctl = reinterpret_cast<Ctl *>(base + RT_STATIC_OFF_CTL);
    pool.blocks = reinterpret_cast<int32_t *>(base + RT_STATIC_OFF_BLOCKS);
    pool.planes = reinterpret_cast<float *>(base + RT_STATIC_OFF_PLANES);
    float4 *nodes = reinterpret_cast<float4 *>(base + RT_STATIC_OFF_NODES);
    for (int k = 0; k < 2; k++) {
        pool.qa[k] = nodes + (size_t)(4 * k + 0) * RT_STATIC_CAPACITY;
        pool.qb[k] = nodes + (size_t)(4 * k + 1) * RT_STATIC_CAPACITY;
        pool.qc[k] = nodes + (size_t)(4 * k + 2) * RT_STATIC_CAPACITY;
        pool.ql[k] = nodes + (size_t)(4 * k + 3) * RT_STATIC_CAPACITY;
    }
    pool.hit = reinterpret_cast<float2 *>(base + RT_STATIC_OFF_HIT);
    float4 *sh4 = reinterpret_cast<float4 *>(base + RT_STATIC_OFF_SHADOW);
    float2 *sh2 = reinterpret_cast<float2 *>(sh4 + 4 * RT_STATIC_CAPACITY);
    for (int k = 0; k < 2; k++) {
        pool.sa[k] = sh4 + (size_t)(2 * k + 0) * RT_STATIC_CAPACITY;
        pool.sb[k] = sh4 + (size_t)(2 * k + 1) * RT_STATIC_CAPACITY;
        pool.sc2[k] = sh2 + (size_t)k * RT_STATIC_CAPACITY;
    }
    rt_stats st;
    rc = render_in_pool(s, cam, p, pool, RT_STATIC_SEG_CAPACITY, RT_STATIC_SEG_CAPACITY, target, accum_rgb_dev, &st, t0, true);
    if (rc) return rc;
    st.device_bytes = RT_STATIC_SCRATCH_BYTES;
    if (stats) *stats = st;
    return RT_OK;
}

int32_t rt_primary_hits(rt_scene *s, const rt_camera *cam, const rt_params *p, int32_t *prim_id, float *t) {
    if (!s || !cam || !p || !prim_id || !t) return fail(RT_ERR_INVALID, "null argument");
    int rc = check_params(*p);
    if (rc) return rc;
    rt_context *ctx = s->ctx;
    rc = set_device(ctx);
    if (rc) return rc;
    const size_t n = (size_t)p->width * p->height;
    HIP_TRY(ctx->prim_tmp.ensure(n));
    HIP_TRY(ctx->t_tmp.ensure(n));
    LaunchCfg cfg{ctx->stream, 0, 0, 0};
    launch_primary_hits(cfg, s->dev, make_const(*cam, *p), ctx->prim_tmp.p, ctx->t_tmp.p);
    HIP_TRY(hipMemcpyAsync(prim_id, ctx->prim_tmp.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(t, ctx->t_tmp.p, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

static int32_t cast_i32(float v) { // Rust `as i32`
    if (v != v) return 0;
    if (v >= 2147483648.0f) return 2147483647;
    if (v <= -2147483648.0f) return (int32_t)(-2147483647 - 1);
    return (int32_t)v;
}

int32_t rt_resolve(const float *accum_rgb, int64_t n_pixels, int32_t spp, int32_t *out_i32, uint8_t *rgb8) {
    if (!accum_rgb || n_pixels < 0 || spp <= 0) return fail(RT_ERR_INVALID, "bad argument");
    for (int64_t i = 0; i < n_pixels * 3; i++) {
        float c = accum_rgb[i] / (float)spp; // main.rs:45
        c = std::sqrt(c);                    // main.rs:46
        int32_t v = cast_i32(255.99f * c);   // main.rs:48-50
        if (out_i32) out_i32[i] = v;
        if (rgb8) rgb8[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
    return RT_OK;
}

int32_t rt_resolve_device(rt_context *ctx, const float *accum_rgb_dev, int64_t n_pixels, int32_t spp, uint8_t *rgb8_dev) {
    if (!ctx || !accum_rgb_dev || !rgb8_dev || n_pixels < 0 || spp <= 0) return fail(RT_ERR_INVALID, "bad argument");
    int rc = set_device(ctx);
    if (rc) return rc;
    launch_resolve(ctx->stream, accum_rgb_dev, n_pixels, spp, rgb8_dev);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipGetLastError());
    return RT_OK;
}


// ---- per-GPU accumulation buffers and their reduction ---------------------------------------------------------------------
int32_t rt_accum_create(rt_context *ctx, int32_t width, int32_t height, float **out) {
    if (!ctx || !out || width <= 0 || height <= 0) return fail(RT_ERR_INVALID, "rt_accum_create: bad argument");
    int rc = set_device(ctx);
    if (rc) return rc;
    const size_t bytes = (size_t)width * (size_t)height * 3u * sizeof(float);
    float *p = nullptr;
    HIP_TRY_ALLOC(hipMalloc((void **)&p, bytes));
    hipError_t e = hipMemsetAsync(p, 0, bytes, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        (void)hipFree(p);
        return fail(RT_ERR_DEVICE, std::string("rt_accum_create: ") + hipGetErrorString(e));
    }
    *out = p;
    return RT_OK;
}

int32_t rt_accum_destroy(rt_context *ctx, float *accum_dev) {
    if (!ctx) return fail(RT_ERR_INVALID, "null argument");
    if (!accum_dev) return RT_OK;
    int rc = set_device(ctx);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipFree(accum_dev));
    return RT_OK;
}

int32_t rt_accum_download(rt_context *ctx, const float *accum_dev, float *host_out, int64_t n_floats) {
    if (!ctx || !accum_dev || !host_out || n_floats < 0) return fail(RT_ERR_INVALID, "rt_accum_download: bad argument");
    int rc = set_device(ctx);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(host_out, accum_dev, (size_t)n_floats * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

} // extern "C"

namespace {

// RCCL, resolved at the first reduction over more than one device.  A process that already holds an RCCL (PyTorch's
// bundled copy has the same soname) gets that one.
struct Rccl {
    void *lib = nullptr;
    decltype(&ncclCommInitAll) comm_init_all = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclReduce) reduce = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    std::string error;
    bool tried = false;
};
Rccl g_rccl;
std::mutex g_rccl_mutex;
std::map<std::vector<int>, std::vector<ncclComm_t>> g_comms; // by device list; kept until the process ends

bool load_rccl() { // g_rccl_mutex held
    if (g_rccl.tried) return g_rccl.lib != nullptr;
    g_rccl.tried = true;
    std::vector<std::string> names;
    if (const char *v = std::getenv("RT_RCCL_LIB")) names.push_back(v);
    names.insert(names.end(), {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"});
    for (const std::string &n : names) {
        g_rccl.lib = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.lib) break;
        if (const char *e = dlerror()) g_rccl.error = e;
    }
    if (!g_rccl.lib) return false;
    auto sym = [&](const char *name) {
        void *p = dlsym(g_rccl.lib, name);
        if (!p) g_rccl.error = std::string("missing symbol ") + name;
        return p;
    };
    g_rccl.comm_init_all = reinterpret_cast<decltype(g_rccl.comm_init_all)>(sym("ncclCommInitAll"));
    g_rccl.comm_destroy = reinterpret_cast<decltype(g_rccl.comm_destroy)>(sym("ncclCommDestroy"));
    g_rccl.reduce = reinterpret_cast<decltype(g_rccl.reduce)>(sym("ncclReduce"));
    g_rccl.group_start = reinterpret_cast<decltype(g_rccl.group_start)>(sym("ncclGroupStart"));
    g_rccl.group_end = reinterpret_cast<decltype(g_rccl.group_end)>(sym("ncclGroupEnd"));
    g_rccl.error_string = reinterpret_cast<decltype(g_rccl.error_string)>(sym("ncclGetErrorString"));
    if (!g_rccl.comm_init_all || !g_rccl.comm_destroy || !g_rccl.reduce || !g_rccl.group_start || !g_rccl.group_end || !g_rccl.error_string) {
        dlclose(g_rccl.lib);
        g_rccl.lib = nullptr;
        return false;
    }
    return true;
}

void drop_communicators_of(int device) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    for (auto it = g_comms.begin(); it != g_comms.end();) {
        if (std::find(it->first.begin(), it->first.end(), device) == it->first.end()) { ++it; continue; }
        for (size_t k = 0; k < it->second.size(); k++) {
            (void)hipSetDevice(it->first[k]);
            (void)g_rccl.comm_destroy(it->second[k]);
        }
        it = g_comms.erase(it);
    }
    (void)hipSetDevice(device);
}

} // namespace

extern "C" {

int32_t rt_reduce_accum(rt_context *const *ctxs, int32_t n, float *const *accum_dev, int64_t n_floats, int32_t root) {
    if (!ctxs || !accum_dev || n < 1 || root < 0 || root >= n || n_floats < 0) return fail(RT_ERR_INVALID, "rt_reduce_accum: bad argument");
    for (int i = 0; i < n; i++) {
        if (!ctxs[i] || !accum_dev[i]) return fail(RT_ERR_INVALID, "rt_reduce_accum: null context or buffer");
        for (int j = 0; j < i; j++)
            if (accum_dev[j] == accum_dev[i]) return fail(RT_ERR_INVALID, "rt_reduce_accum: the same buffer twice");
    }
    // 1. buffers that share a device are added into that device's representative (the root's buffer on the root's device)
    std::vector<int> devices, rep; // distinct devices in order of first appearance, and the index of their representative
    for (int i = 0; i < n; i++) {
        size_t k = 0;
        while (k < devices.size() && devices[k] != ctxs[i]->device) k++;
        if (k == devices.size()) { devices.push_back(ctxs[i]->device); rep.push_back(i); }
        if (i == root) rep[k] = root;
    }
    for (int i = 0; i < n; i++) {
        size_t k = 0;
        while (devices[k] != ctxs[i]->device) k++;
        if (rep[k] == i) continue;
        rt_context *rc_ctx = ctxs[rep[k]];
        int rc = set_device(rc_ctx);
        if (rc) return rc;
        launch_accum_add(rc_ctx->stream, accum_dev[rep[k]], accum_dev[i], n_floats);
        HIP_TRY(hipGetLastError());
    }
    for (size_t k = 0; k < devices.size(); k++) {
        int rc = set_device(ctxs[rep[k]]);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(ctxs[rep[k]]->stream));
    }
    // 2. one RCCL reduce over the distinct devices (a single device goes through RCCL only on request: RT_REDUCE_RCCL_ALWAYS=1)
    if (devices.size() == 1 && env_u32_allow_zero("RT_REDUCE_RCCL_ALWAYS", 0) == 0) return RT_OK;
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (!load_rccl()) return fail(RT_ERR_UNSUPPORTED, "rt_reduce_accum: RCCL is not available: " + g_rccl.error);
    auto nccl_fail = [&](const char *what, ncclResult_t r) { return fail(RT_ERR_DEVICE, std::string(what) + ": " + g_rccl.error_string(r)); };
    auto it = g_comms.find(devices);
    if (it == g_comms.end()) {
        std::vector<ncclComm_t> comms(devices.size());
        ncclResult_t r = g_rccl.comm_init_all(comms.data(), (int)devices.size(), devices.data());
        if (r != ncclSuccess) return nccl_fail("ncclCommInitAll", r);
        it = g_comms.emplace(devices, comms).first;
    }
    size_t root_k = 0;
    while (devices[root_k] != ctxs[root]->device) root_k++;
    ncclResult_t r = g_rccl.group_start();
    if (r != ncclSuccess) return nccl_fail("ncclGroupStart", r);
    for (size_t k = 0; k < devices.size(); k++) {
        int rc = set_device(ctxs[rep[k]]);
        if (rc) { (void)g_rccl.group_end(); return rc; }
        float *buf = accum_dev[rep[k]];
        r = g_rccl.reduce(buf, buf, (size_t)n_floats, ncclFloat, ncclSum, (int)root_k, it->second[k], ctxs[rep[k]]->stream);
        if (r != ncclSuccess) { (void)g_rccl.group_end(); return nccl_fail("ncclReduce", r); }
    }
    r = g_rccl.group_end();
    if (r != ncclSuccess) return nccl_fail("ncclGroupEnd", r);
    for (size_t k = 0; k < devices.size(); k++) {
        int rc = set_device(ctxs[rep[k]]);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(ctxs[rep[k]]->stream));
    }
    return RT_OK;
}

} // extern "C"

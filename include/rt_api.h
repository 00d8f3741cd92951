/*
 * rt_api.h — C ABI of the MI355X path-tracing backend.
 *
 * The reference (raytracer-rs) has no FFI of its own; the narrowest seam in it is
 * the per-sample shader call inside the pixel loop (src/bin/main.rs:28-52), fed by
 * `Scene { cam, world }` (src/scene.rs:14-17).  This header replaces that loop at
 * image granularity: the host flattens its Scene/World into the plain arrays below,
 * hands them over once (rt_scene_create) and then asks for whole images
 * (rt_render / rt_primary_hits).  Everything here is POD: plain pointers and sizes,
 * no C++ types, no exceptions across the boundary.  Every function returns an
 * int32 status (RT_OK == 0); rt_last_error() gives the message of the last
 * failure on the calling thread.
 */
#ifndef RT_API_H
#define RT_API_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------ */
enum {
    RT_OK = 0,
    RT_ERR_INVALID = 1,     /* bad argument / malformed scene (reference: assert!/io::Error) */
    RT_ERR_UNSUPPORTED = 2, /* combination the reference rejects with assert!(false), e.g.
                               Whitted shader + Microfacet (material.rs:328-331) or + PolygonLight (light.rs:292-301) */
    RT_ERR_DEVICE = 3,      /* HIP runtime failure */
    RT_ERR_NOMEM = 4
};

/* ---- materials: src/material.rs:31-34,117-120,182-185,305-311 ----------- */
enum { RT_MAT_DIFFUSE = 0, RT_MAT_METAL = 1, RT_MAT_DIELECTRIC = 2, RT_MAT_MICROFACET = 3 };

typedef struct rt_material {
    int32_t kind;
    float albedo[3]; /* Diffuse/Metal: albedo.  Microfacet: attenuation.  Dielectric: unused */
    float f0[3];     /* Microfacet only */
    float roughness; /* Microfacet only */
    float metallic;  /* Microfacet only */
    float ref_idx;   /* Dielectric only */
} rt_material;

/* ---- lights: src/light.rs:54-57,106-112,242-247 ------------------------- */
enum { RT_LIGHT_POINT = 0, RT_LIGHT_DISK = 1, RT_LIGHT_POLYGON = 2 };

/* sampler kinds, numbering of SamplerKind::from_int (src/sampler.rs:12-22) */
enum { RT_SAMPLER_WHITE = 1, RT_SAMPLER_UNIFORM = 2, RT_SAMPLER_JITTERED = 3, RT_SAMPLER_BLUE = 4 };

typedef struct rt_light {
    int32_t kind;
    float origin[3];      /* Point, Disk */
    float radius;         /* Disk (horizontal disk, normal +-y) */
    float spectrum[3];
    int32_t tri_first;    /* Polygon: range in rt_scene_desc.light_tri_v */
    int32_t tri_count;
    int32_t sampler_kind; /* Disk: RT_SAMPLER_* used to place samples on the disk */
} rt_light;

/* ---- world objects: src/hitable.rs:25-27, src/geometry.rs:9-12,76-80,119-126, src/mesh.rs:17-20 */
enum { RT_OBJ_TRIANGLE = 0, RT_OBJ_SPHERE = 1, RT_OBJ_CYLINDER = 2, RT_OBJ_MESH = 3 };
/* mesh flavours of the reference: NaiveMesh / BoxMesh / FastMesh (src/mesh.rs:62,98,99).  All three are defined
 * here as "closest front-facing triangle of the face list" (the brute-force result); the field is informational. */
enum { RT_ACCEL_NAIVE = 0, RT_ACCEL_BOX = 1, RT_ACCEL_BVH = 2 };

typedef struct rt_object {
    int32_t kind;  /* RT_OBJ_* */
    int32_t first; /* first element in the class array (tri / sph / cyl); a MESH is a run of triangles */
    int32_t count; /* 1 for TRIANGLE/SPHERE/CYLINDER, #faces for MESH */
    int32_t accel; /* RT_ACCEL_* (MESH only) */
} rt_object;

/*
 * Flattened scene.  All pointers are borrowed host memory, copied during rt_scene_create.
 * `objects` lists the world in the order of the reference's HitableList (src/hitable.rs:35-56); that order
 * defines the primitive id reported by rt_primary_hits: objects are numbered in list order, a mesh contributing
 * one id per face in face_list order.  The class arrays must hold their elements in that same world order.
 */
typedef struct rt_scene_desc {
    int32_t n_tri;
    const float *tri_v;     /* 9 floats per triangle: v0 v1 v2 */
    const int32_t *tri_mat; /* material index per triangle */
    int32_t n_sph;
    const float *sph;       /* 4 floats per sphere: cx cy cz r */
    const int32_t *sph_mat;
    int32_t n_cyl;
    const float *cyl;       /* 5 floats per cylinder: center_x center_z y_min y_max radius */
    const int32_t *cyl_mat;
    int32_t n_mat;
    const rt_material *mats;
    int32_t n_light;
    const rt_light *lights; /* order of LightList.list (src/light.rs:350-352): first hit in list order wins */
    int32_t n_light_tri;
    const float *light_tri_v; /* 9 floats per light triangle */
    int32_t n_obj;
    const rt_object *objects;
} rt_scene_desc;

/* Camera after Camera::new (src/camera.rs:13-27); get_ray(s,t) = { origin, llc + s*horizontal + t*vertical - origin } */
typedef struct rt_camera {
    float origin[3];
    float llc[3];
    float horizontal[3];
    float vertical[3];
} rt_camera;

/* shaders: src/shader.rs:7,22,58,93 */
enum { RT_SHADER_NORMAL = 0, RT_SHADER_WHITTED = 1, RT_SHADER_PATH = 2, RT_SHADER_MIS = 3 };

enum {
    RT_FLAG_TIMING = 1, /* time every kernel launch with HIP events on the render stream and fill rt_stats.ms_* */
    RT_FLAG_COUNT_WORK = 2, /* count BVH node visits and primitive tests (slower; fills rt_stats.nodes/tris) */
    /* Options the reference does not have; off by default, results then follow the reference.  MIS and path shaders only
     * (the Whitted and normal shaders ignore them). */
    RT_FLAG_RUSSIAN_ROULETTE = 4, /* a scattered ray at depth >= RT_RR_MIN_DEPTH survives with probability q = largest
                                     component of the material's attenuation, clamped to [0.05, 1], and is weighted 1/q:
                                     same expectation as src/shader.rs:93-115, shorter paths */
    RT_FLAG_COSINE_DIFFUSE = 8    /* Diffuse surfaces scatter with density cos/pi (direction = normal + point on the unit
                                     sphere) and the cosine enters the scattered ray's weight, instead of the uniform
                                     hemisphere with nominal density 1/pi of src/material.rs:85-103,144-151: changes the
                                     expectation (towards the rendering equation's) */
};

#define RT_RR_MIN_DEPTH 3

typedef struct rt_params {
    int32_t width, height;      /* NX, NY (src/lib.rs:21-22) */
    int32_t spp;                /* NS (src/lib.rs:23): size of the per-pixel sample set and the divisor in rt_resolve */
    int32_t spp_begin, spp_end; /* sample indices [begin,end) rendered by this call (spp-range sharding) */
    int32_t tile_size;          /* 0: whole frame.  >0: only square tiles t with t % shard_count == shard_rank,
                                   tiles numbered row-major from the top-left */
    int32_t shard_rank, shard_count;
    int32_t shader;             /* RT_SHADER_* */
    int32_t pixel_sampler;      /* RT_SAMPLER_* for the sub-pixel offsets (src/bin/main.rs:31-34) */
    int32_t light_samples;      /* NS2 (src/lib.rs:24) */
    int32_t max_depth;          /* 40 (src/shader.rs:23,59,94): depth > max_depth returns black */
    float t_min, t_max;         /* T_MIN, T_MAX (src/lib.rs:17-18) */
    uint32_t seed;
    uint32_t flags;             /* RT_FLAG_* */
} rt_params;

typedef struct rt_stats {
    uint64_t camera_samples;
    uint64_t rays_closest;   /* closest-hit queries: primary + scattered */
    uint64_t rays_shadow;    /* shadow (occlusion) queries */
    uint64_t nodes_visited;  /* RT_FLAG_COUNT_WORK: tree nodes visited, closest-hit and occlusion queries together */
    uint64_t tris_tested;    /* RT_FLAG_COUNT_WORK: exact triangle tests (tree leaves and the scanned list), both kinds of query */
    uint64_t iterations;     /* wavefront iterations (bounce steps) */
    uint64_t launches_extend, launches_shade, launches_shadow, launches_raygen;
    double ms_total;         /* wall time of the call, device work complete */
    double ms_extend, ms_shade, ms_shadow, ms_raygen; /* RT_FLAG_TIMING: summed HIP-event durations */
    uint64_t device_bytes;   /* device memory the context holds for queues and planes after this call */
    uint64_t rays_shadow_inline; /* of rays_shadow: answered by the shading kernel itself, never written to the shadow queue */
    uint64_t nodes_carrying;     /* of rays_closest: path nodes that brought the light gathered so far along (16 more bytes read,
                                    and written for their successor), to be added to the pixel once at the path's end */
    /* RT_FLAG_COUNT_WORK, the split the per-kernel rooflines need: */
    uint64_t nodes_visited_shadow; /* of nodes_visited: by occlusion queries */
    uint64_t tris_tested_shadow;   /* of tris_tested: tree-leaf triangles tested by occlusion queries */
    uint64_t tris_scanned;         /* of tris_tested: tests of the triangles kept outside the tree (rt_scene_info.n_loose),
                                      which a wave reads through uniform loads, not one address per lane */
} rt_stats;

typedef struct rt_context rt_context;
typedef struct rt_scene rt_scene;

/* Message of the last failing call on this thread ("" if none). */
const char *rt_last_error(void);

/* One context per device (one process per GPU).  Creates the render stream and the queue pool. */
int32_t rt_context_create(int32_t device_id, rt_context **out);
int32_t rt_context_destroy(rt_context *ctx);

/* Replaces the construction of `World` in Scene::* (src/scene.rs:170-351): validates the description
 * (PolygonLight planarity, light.rs:252-269), builds the BVH over all triangles, uploads everything. */
int32_t rt_scene_create(rt_context *ctx, const rt_scene_desc *desc, rt_scene **out);
int32_t rt_scene_destroy(rt_scene *scene);

/*
 * rt_scene_create with flags.  RT_SCENE_BVH_ON_DEVICE: the acceleration structure is built on the GPU (Morton codes,
 * radix sort, one thread per inner node, bottom-up refit) instead of on the host — the counterpart of
 * BVHTree::build_tree (src/mesh.rs:225-261), which the reference re-runs after every Mesh::transform (src/mesh.rs:42).
 * The tree differs from the host builder's (one triangle per leaf, 2n-1 nodes, no LDS-resident top); hits and images
 * do not.  rt_scene_create is rt_scene_create_ex with flags 0 (or RT_SCENE_BVH_ON_DEVICE when RT_BVH_DEVICE=1 is set
 * in the environment).
 */
enum { RT_SCENE_BVH_ON_DEVICE = 1 };
int32_t rt_scene_create_ex(rt_context *ctx, const rt_scene_desc *desc, uint32_t flags, rt_scene **out);

typedef struct rt_scene_info {
    int32_t n_triangles;   /* all triangles of the world */
    int32_t n_nodes;       /* per visiting order; the tree holds n_triangles - n_loose triangles */
    int32_t n_orders;      /* 8 = one link set per direction octant, 1 = a single order */
    int32_t n_top;         /* nodes kept in LDS by the traversal kernels */
    int32_t bvh_on_device; /* built by the GPU builder */
    int32_t n_loose;       /* triangles kept out of the tree: when the triangles outside the big meshes (>= 64 faces) are few
                              (at most 16), every ray scans them in list order, as hit_list does (src/hitable.rs:44-50), and
                              only rays that touch a big mesh's box walk the tree; 0 when the tree holds everything */
    double bvh_build_ms;   /* wall time of the build, uploads / downloads included */
} rt_scene_info;
int32_t rt_scene_get_info(const rt_scene *scene, rt_scene_info *info);

/*
 * Replaces the pixel x sample loop of src/bin/main.rs:28-44: adds the radiance of samples [spp_begin,spp_end) of
 * every selected pixel into accum_rgb (width*height*3 floats, row 0 = top image row, i.e. j = height-1 of the
 * reference loop).  accum_rgb is a device pointer if accum_is_device != 0, else host memory (read, added to,
 * written back).  The buffer is NOT cleared: callers accumulate across calls.  stats may be NULL.
 */
int32_t rt_render(rt_scene *scene, const rt_camera *cam, const rt_params *params, float *accum_rgb,
                  int32_t accum_is_device, rt_stats *stats);

/*
 * No counterpart in the reference.  rt_render for callers that cannot allocate device memory: while another stream of
 * the process is capturing a graph, tight serving loops, harnesses that forbid hipMalloc (the call itself synchronises
 * its streams and reads loop counters back, so it cannot be captured).  Same arguments and results as rt_render with
 * accum_is_device != 0 (accum_rgb_dev must be a device pointer), but the path queues, hit records, shadow queues,
 * colour planes and loop counters live in a fixed array in device memory (about 139 MB per GPU, held from library
 * load until the process ends) instead of the context's pool, and the call creates no events: it calls no hipMalloc,
 * hipFree or other allocating routine.  Limits, rejected with RT_ERR_INVALID: ceil(width/8)*ceil(height/8)*64 <= 2^21
 * pixels (1920x1080 fits), RT_FLAG_TIMING not set, and for the Whitted shader at most 64 shadow rays per surface
 * point.  The refill level is fixed at 2^16 path nodes, so large frames take more iterations than with rt_render.
 * Every call on a GPU uses that GPU's one array: calls that overlap on one GPU (two threads, two contexts, two
 * processes do NOT share it; two contexts of one process do) race.
 */
int32_t rt_render_static(rt_scene *scene, const rt_camera *cam, const rt_params *params, float *accum_rgb_dev, rt_stats *stats);

/* Parity AOV: primitive id (-1 = miss) and t of world.objects.hit(primary ray, t_min, t_max) for the sample
 * spp_begin of every pixel (src/shader.rs:99 with depth 0).  Host output buffers of width*height entries. */
int32_t rt_primary_hits(rt_scene *scene, const rt_camera *cam, const rt_params *params, int32_t *prim_id, float *t);

/* src/bin/main.rs:45-50: c/NS, sqrt, (255.99*c) as i32 (Rust cast: saturating, NaN -> 0).  out_i32 receives the
 * raw integers (3 per pixel; values above 255 are possible exactly as in the reference); rgb8 (may be NULL)
 * receives them clamped to 0..255.  Host pointers. */
int32_t rt_resolve(const float *accum_rgb, int64_t n_pixels, int32_t spp, int32_t *out_i32, uint8_t *rgb8);

/* Device-side resolve of a device accumulation buffer into device rgb8 (same arithmetic), on the render stream. */
int32_t rt_resolve_device(rt_context *ctx, const float *accum_rgb_dev, int64_t n_pixels, int32_t spp, uint8_t *rgb8_dev);

/*
 * Per-GPU accumulation buffers and their reduction: the multi-GPU form of the pixel x sample loop (src/bin/main.rs:28-52
 * has one `col` per pixel on one thread; here every GPU sums its share of the samples or tiles of a frame and the sums
 * meet once).  One context per device, all in ONE process; each context is driven by its own host thread while it
 * renders (rt_render blocks), any thread may call the reduction afterwards.
 *
 * rt_accum_create: a zeroed device buffer of width*height*3 floats on the context's device, to be passed to rt_render with
 * accum_is_device != 0.  rt_accum_download: copies n_floats floats to host memory.  rt_accum_destroy frees it.
 */
int32_t rt_accum_create(rt_context *ctx, int32_t width, int32_t height, float **accum_dev_out);
int32_t rt_accum_destroy(rt_context *ctx, float *accum_dev);
int32_t rt_accum_download(rt_context *ctx, const float *accum_dev, float *host_out, int64_t n_floats);

/*
 * accum_dev[root] += sum of accum_dev[i], i != root (n_floats floats each; accum_dev[i] lives on ctxs[i]'s device; the
 * other buffers are left as they are).  Buffers on different devices are summed with one RCCL reduce (ncclReduce,
 * sum, over xGMI; communicators from ncclCommInitAll are created at the first call for a set of devices and kept until
 * one of those devices' contexts is destroyed);
 * buffers that share a device (logical shards on one GPU) are added by a kernel first.  RCCL is loaded when first
 * needed (librccl.so.1; RT_RCCL_LIB overrides the name): RT_ERR_UNSUPPORTED if it is missing.  Tile shards touch
 * disjoint pixels, so their sum is exact; sample-range shards differ from one GPU's result by float summation order.
 * All work of the contexts must be complete (rt_render has returned); the call returns with the sum complete.
 */
int32_t rt_reduce_accum(rt_context *const *ctxs, int32_t n, float *const *accum_dev, int64_t n_floats, int32_t root);

/* The HIP stream (hipStream_t) all work of this context is enqueued on, for callers that time with events. */
void *rt_context_stream(rt_context *ctx);

/* Fills p with the reference defaults: NS2 = 9, max_depth = 40, T_MIN = 1e-3, T_MAX = 1e7, MIS shader, blue-noise
 * (= white, see sampler.rs:243-250) pixel sampler, whole frame, all samples. */
void rt_params_default(rt_params *p, int32_t width, int32_t height, int32_t spp);

/* Host-only diagnostic (no GPU needed): builds the acceleration structure rt_scene_create builds for the given
 * triangles (9 floats each; primitive id = position in the list) and copies it out.  Takes the place of
 * BVHTree::build_tree (src/mesh.rs:225-261).  Layout: `ordered` != 0 gives 8 node arrays, one per sign octant of the
 * ray direction (bit a of the array index set = component a negative), else 1; each array has *n_nodes nodes of 8
 * floats: xmin, xmax, ymin, ymax, zmin, zmax, link, miss (the last two are int32 bit patterns); with 8 arrays the pair
 * of an axis is stored as (max, min) in the arrays of the octants whose direction is negative along that axis, i.e.
 * always (plane the octant's rays enter through, plane they leave through).  link >= 0: node to
 * continue with when the box is hit; link < 0: leaf with triangles [first, first + count), ~link = first * 8 + count - 1.
 * miss: node to continue with when the box is missed or the leaf is done, -1 = walk finished.  Triangles are written
 * in leaf order, 12 floats each: v0.xyz, primitive id (int32 bits), v1 - v0, material (int32 bits, 0 here), v2 - v0, 0.
 * nodes_out / tris_out may be NULL (sizes only); they must hold n_orders * n_nodes * 8 and n_tri * 12 floats. */
int32_t rt_bvh_build_host(const float *tri_v, int32_t n_tri, int32_t max_top, int32_t ordered, float *nodes_out, float *tris_out,
                          int32_t *n_nodes, int32_t *n_top, int32_t *n_orders);

/* Host-only diagnostic like rt_bvh_build_host, for the second node format rt_scene_create uploads: the same tree in
 * 16-byte nodes, *n_orders arrays of *n_nodes nodes, each array in the depth-first order of its octant's visiting
 * order (so the first child of an inner node is the next node).  Four uint32 per node: x, y, z = entry plane |
 * exit plane << 16 for the rays of that octant, quantised to 16 bits over the root box and rounded outwards
 * (plane = q_params[a] + q * q_params[3 + a] along axis a); w = inner node: index to continue with when the box is
 * missed (*n_nodes = walk finished), leaf: 0x80000000 | first_triangle * 8 + count - 1 (triangle order: that of
 * rt_bvh_build_host's tris_out).  cnodes_out may be NULL (sizes only). */
int32_t rt_bvh_build_host_compact(const float *tri_v, int32_t n_tri, int32_t ordered, uint32_t *cnodes_out, float *q_params, int32_t *n_nodes,
                                  int32_t *n_orders);

#ifdef __cplusplus
}
#endif
#endif /* RT_API_H */

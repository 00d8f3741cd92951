// rt_kernels.h — host-visible launchers of the wavefront stages (implemented in rt_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device.h"

namespace rtd {

// Path-node queue (SoA, 48 B per node, 64 B when the node carries its path's light):
//   a = (o.x, o.y, o.z, d.x)   b = (d.y, d.z, w.x, w.y)   c = (w.z, b_pdf, meta, node_key)
// w = path weight so far times the BSDF value of the last bounce; the divisor (light pdf + b_pdf) of that bounce is
// applied when the node is shaded, once the extend stage knows which light the ray meets (LightList::pdf).
// meta = pixel (24 bits) | depth << 24 | root << 30.
// l = (L.x, L.y, L.z, -): light already gathered along the path (emission met, next-event samples whose occlusion
// query the shading kernel answered itself), carried to the path's end and added to the pixel once — when
// DevScene::carry_out / carry_in say so; otherwise l is neither written nor read and every contribution is added where it arises.
struct NodeQueue {
    float4 *a, *b, *c, *l;
};

// Shadow-ray queue (40 B per ray): a = (o.xyz, d.x), b = (d.yz, c.x, c.y), c = (c.z, pixel); c = contribution added
// to the pixel when nothing lies between the surface point and the light sample.
struct ShadowQueue {
    float4 *a, *b;
    float2 *c;
};

// The queues are cut into NSEG equal segments, each with its own fill counter on its own 128-byte line: appends are
// one returning atomic per workgroup, and a single counter word would serialise them (about 90 per microsecond).
// Segments [0, RT_NSEG/2) hold "light" rays, [RT_NSEG/2, RT_NSEG) "heavy" ones (see DevScene::n_heavy); a producing
// workgroup b appends to segment (b % (RT_NSEG/2)) of each half; consumers run a (blocks, NSEG) grid, heavy half first.
#define RT_NSEG 32
// camera candidates one workgroup / one thread of the ray generation kernel takes (256 threads)
#define RT_RAYGEN_PER_THREAD 4
#define RT_RAYGEN_PER_WG (256 * RT_RAYGEN_PER_THREAD)
#define RT_NSEG_HALF (RT_NSEG / 2)

struct SegCounter {
    uint32_t v;
    uint32_t pad[31];
};

// Device-resident loop state, written by k_plan and the appending stages; the host reads back the head only.
struct CtlHead {
    uint32_t n_alive;   // nodes carried into this iteration (before ray generation)
    uint32_t n_new;     // camera candidates to generate this iteration
    uint32_t max_in_light, max_in_heavy; // largest per-segment node count of each class before ray generation
    uint32_t max_sh_light, max_sh_heavy; // largest per-segment shadow-ray count queued by the previous iteration
    // Camera candidates are numbered tile-major in chunks of `chunk_samples` samples: chunk -> 8x8 pixel block -> sample
    // inside the chunk -> pixel inside the block.  The rays of one block are generated together and stay together in
    // the queues, so the atomic adds of a wave fall into few cache lines.
    uint32_t first_sample;    // first sample (relative to spp_begin) of the chunk the first candidate of this iteration lies in
    uint32_t first_rem;       // ... and the candidate's position inside that chunk
    uint32_t per_sample;      // camera candidates per sample = 64 * number of 8x8 pixel blocks
    uint32_t n_samples;       // samples of this render call (spp_end - spp_begin)
    uint32_t chunk_samples;   // samples per chunk (the last chunk may hold fewer)
    uint32_t iterations;
    uint32_t done;      // nothing left to do
    uint32_t overflow;  // a queue segment ran out of room (the render call fails)
    uint32_t seq;       // runs of k_plan in this render call; in the host's copy it is written last (see k_plan)
    uint32_t pad0;
    unsigned long long cursor; // next camera candidate not yet generated
    unsigned long long total;  // all camera candidates of this render call
    unsigned long long rays_closest, rays_shadow, cam_samples, nodes, tris;
    unsigned long long rays_shadow_inline; // of rays_shadow: answered inside k_shade
    // work-counting launches: nodes / tris = tree nodes visited and leaf triangles tested by closest-hit rays, nodes_sh /
    // tris_sh by occlusion queries, scanned = exact tests of the list of triangles kept outside the tree (both kinds)
    unsigned long long nodes_sh, tris_sh, scanned;
    // work-counting launches only: closest-hit rays / waves by floor(log2(visited nodes)) + 1 (bucket 0 = none), and the
    // sum over waves of the largest count among their lanes (the walk length the wave actually pays for)
    unsigned long long hist_ray[16], hist_wave[16], wave_steps, wave_tri_iters;
};

struct Ctl {
    CtlHead h;
    SegCounter seg_in[RT_NSEG];     // nodes per segment of the current queue
    SegCounter seg_out[RT_NSEG];    // nodes appended per segment of the next queue
    SegCounter seg_shadow[RT_NSEG]; // shadow rays per segment, being filled by k_shade
    uint32_t seg_shadow_cur[RT_NSEG]; // ... of the previous iteration, being consumed by k_shadow
    SegCounter seg_inline[RT_NSEG_HALF]; // occlusion queries k_shade answered itself (never queued), per producing segment
};

struct LaunchCfg {
    hipStream_t stream;
    uint32_t seg_capacity;        // nodes per queue segment
    uint32_t shadow_seg_capacity; // shadow rays per segment
    uint32_t target;              // refill level: camera rays are generated while fewer nodes than this are alive
};

// `host_head` (may be null): pinned host memory the kernel copies the head into when it is through, the sequence number
// last — the host can wait for that number instead of a copy and a stream synchronisation.
void launch_plan(const LaunchCfg &cfg, Ctl *ctl, CtlHead *host_head = nullptr);
// Consumer grids are one-dimensional: RT_NSEG_HALF * heavy_blocks workgroups for the heavy segments first (they run
// longest), then RT_NSEG_HALF * light_blocks for the light ones; *_threads = most elements in any one segment of the class.
void launch_raygen(const LaunchCfg &cfg, const DevScene &sc, const RenderConst &rc, Ctl *ctl, NodeQueue in, const int32_t *blocks, uint32_t blocks_w,
                   uint32_t n_threads);
void launch_extend(const LaunchCfg &cfg, const DevScene &sc, const RenderConst &rc, Ctl *ctl, NodeQueue in, float2 *hit, uint32_t light_threads,
                   uint32_t heavy_threads, bool count);
void launch_shade(const LaunchCfg &cfg, int shader, const DevScene &sc, const RenderConst &rc, Ctl *ctl, NodeQueue in, NodeQueue out,
                  const float2 *hit, ShadowQueue sh, float *planes, uint32_t n_pixels, uint32_t light_threads, uint32_t heavy_threads);
void launch_shadow(const LaunchCfg &cfg, const DevScene &sc, const RenderConst &rc, Ctl *ctl, ShadowQueue sh, float *planes, uint32_t n_pixels,
                   uint32_t light_threads, uint32_t heavy_threads, bool count);
// the end of a call: every node of the current queue is followed to the end of its path by one thread (MIS and path shaders)
void launch_tail(const LaunchCfg &cfg, int shader, const DevScene &sc, const RenderConst &rc, Ctl *ctl, NodeQueue in, NodeQueue out, float *planes,
                 uint32_t n_pixels, uint32_t light_threads, uint32_t heavy_threads);
void launch_normal_shader(const LaunchCfg &cfg, const DevScene &sc, const RenderConst &rc, float *planes, uint32_t n_pixels);
void launch_primary_hits(const LaunchCfg &cfg, const DevScene &sc, const RenderConst &rc, int32_t *prim, float *t);
void launch_planes_to_rgb(const LaunchCfg &cfg, const RenderConst &rc, const float *planes, float *accum_rgb, uint32_t n_pixels);
void launch_tri_normals(hipStream_t stream, const float4 *tris, int n, float4 *out);
void launch_accum_add(hipStream_t stream, float *dst, const float *src, int64_t n);
void launch_resolve(hipStream_t stream, const float *accum_rgb, int64_t n_pixels, int32_t spp, uint8_t *rgb8);

} // namespace rtd

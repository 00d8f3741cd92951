// rt_kernels.h — host-visible launchers of the wavefront stages (implemented in rt_kernels.hip).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device.h"

namespace rtd {

// Path-node queue (SoA, 48 B per node):
//   a = (o.x, o.y, o.z, d.x)   b = (d.y, d.z, w.x, w.y)   c = (w.z, b_pdf, meta, node_key)
// w = path weight so far times the BSDF value of the last bounce; the divisor (light pdf + b_pdf) of that bounce is
// applied when the node is shaded, once the extend stage knows which light the ray meets (LightList::pdf).
// meta = pixel (24 bits) | depth << 24 | root << 30.
struct NodeQueue {
    float4 *a, *b, *c;
};

// Shadow-ray queue (40 B per ray): a = (o.xyz, d.x), b = (d.yz, c.x, c.y), c = (c.z, pixel); c = contribution added
// to the pixel when nothing lies between the surface point and the light sample.
struct ShadowQueue {
    float4 *a, *b;
    float2 *c;
};

// Device-resident loop state, written by k_plan and the appending stages; the host only reads it back.
struct Ctl {
    uint32_t n_in;     // nodes in the current queue (after ray generation)
    uint32_t n_out;    // nodes appended to the next queue so far
    uint32_t n_shadow; // shadow rays queued this iteration
    uint32_t n_new;    // camera candidates to generate this iteration
    unsigned long long first;  // first camera candidate of this iteration
    unsigned long long cursor; // next camera candidate not yet generated
    unsigned long long total;  // all camera candidates of this render call
    unsigned long long rays_closest, rays_shadow, cam_samples, nodes, tris;
    uint32_t iterations;
    uint32_t done;     // nothing left to do
    uint32_t overflow; // a queue ran out of room (the render call fails)
    uint32_t n_proc;   // nodes of the current queue that are extended and shaded this iteration; the rest is carried over
};

struct LaunchCfg {
    hipStream_t stream;
    uint32_t capacity;        // nodes per queue
    uint32_t shadow_capacity; // shadow rays
    uint32_t target;          // refill level: camera rays are generated while fewer nodes than this are alive
    uint32_t max_process;     // most nodes one iteration may shade (bounds the appends of that iteration)
};

void launch_plan(const LaunchCfg &cfg, Ctl *ctl);
void launch_raygen(const LaunchCfg &cfg, const RenderConst &rc, Ctl *ctl, NodeQueue in, const int32_t *blocks, uint32_t n_blocks,
                   uint32_t blocks_w, uint32_t n_threads);
void launch_extend(const LaunchCfg &cfg, const DevScene &sc, const RenderConst &rc, Ctl *ctl, NodeQueue in, float4 *hit, uint32_t n_threads,
                   bool count);
void launch_shade(const LaunchCfg &cfg, int shader, const DevScene &sc, const RenderConst &rc, Ctl *ctl, NodeQueue in, NodeQueue out,
                  const float4 *hit, ShadowQueue sh, float *planes, uint32_t n_pixels, uint32_t n_threads);
void launch_shadow(const LaunchCfg &cfg, const DevScene &sc, const RenderConst &rc, Ctl *ctl, ShadowQueue sh, float *planes, uint32_t n_pixels,
                   uint32_t n_threads, bool count);
void launch_normal_shader(const LaunchCfg &cfg, const DevScene &sc, const RenderConst &rc, float *planes, uint32_t n_pixels);
void launch_primary_hits(const LaunchCfg &cfg, const DevScene &sc, const RenderConst &rc, int32_t *prim, float *t);
void launch_planes_to_rgb(const LaunchCfg &cfg, const float *planes, float *accum_rgb, uint32_t n_pixels);
void launch_resolve(hipStream_t stream, const float *accum_rgb, int64_t n_pixels, int32_t spp, uint8_t *rgb8);

} // namespace rtd

//! Declarations of `include/rt_api.h` (the MI355X path-tracing backend) for Rust callers.
//! Source-only: written without a Rust toolchain at hand, never compiled.  Field order, widths and enum values mirror
//! the C header; `tests/test_abi_symbols.py` checks the same layout through ctypes against the C compiler's sizes.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_void};

pub const RT_OK: i32 = 0;
pub const RT_ERR_INVALID: i32 = 1;
pub const RT_ERR_UNSUPPORTED: i32 = 2;
pub const RT_ERR_DEVICE: i32 = 3;
pub const RT_ERR_NOMEM: i32 = 4;


#[repr(C)] #[derive(Clone, Copy)] pub struct rt_material { pub kind: i32, pub albedo: [f32; 3], pub f0: [f32; 3],
    pub roughness: f32, pub metallic: f32, pub ref_idx: f32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct rt_light { pub kind: i32, pub origin: [f32; 3], pub radius: f32,
    pub spectrum: [f32; 3], pub tri_first: i32, pub tri_count: i32, pub sampler_kind: i32 }
#[repr(C)] #[derive(Clone, Copy)] pub struct rt_object { pub kind: i32, pub first: i32, pub count: i32, pub accel: i32 }
#[repr(C)] pub struct rt_scene_desc {
    pub n_tri: i32, pub tri_v: *const f32, pub tri_mat: *const i32,
    pub n_sph: i32, pub sph: *const f32, pub sph_mat: *const i32,
    pub n_cyl: i32, pub cyl: *const f32, pub cyl_mat: *const i32,
    pub n_mat: i32, pub mats: *const rt_material,
    pub n_light: i32, pub lights: *const rt_light,
    pub n_light_tri: i32, pub light_tri_v: *const f32,
    pub n_obj: i32, pub objects: *const rt_object }
#[repr(C)] #[derive(Clone, Copy)] pub struct rt_camera { pub origin: [f32; 3], pub llc: [f32; 3],
    pub horizontal: [f32; 3], pub vertical: [f32; 3] }
#[repr(C)] #[derive(Clone, Copy)] pub struct rt_params { pub width: i32, pub height: i32, pub spp: i32,
    pub spp_begin: i32, pub spp_end: i32, pub tile_size: i32, pub shard_rank: i32, pub shard_count: i32,
    pub shader: i32, pub pixel_sampler: i32, pub light_samples: i32, pub max_depth: i32,
    pub t_min: f32, pub t_max: f32, pub seed: u32, pub flags: u32 }
#[repr(C)] #[derive(Clone, Copy, Default)] pub struct rt_stats { pub camera_samples: u64, pub rays_closest: u64,
    pub rays_shadow: u64, pub nodes_visited: u64, pub tris_tested: u64, pub iterations: u64,
    pub launches_extend: u64, pub launches_shade: u64, pub launches_shadow: u64, pub launches_raygen: u64,
    pub ms_total: f64, pub ms_extend: f64, pub ms_shade: f64, pub ms_shadow: f64, pub ms_raygen: f64,
    pub device_bytes: u64, pub rays_shadow_inline: u64, pub nodes_carrying: u64,
    pub nodes_visited_shadow: u64, pub tris_tested_shadow: u64, pub tris_scanned: u64 }
#[repr(C)] #[derive(Clone, Copy, Default)] pub struct rt_scene_info { pub n_triangles: i32, pub n_nodes: i32, pub n_orders: i32,
    pub n_top: i32, pub bvh_on_device: i32, pub n_loose: i32, pub bvh_build_ms: f64 }
pub const RT_SCENE_BVH_ON_DEVICE: u32 = 1;
pub const RT_FLAG_TIMING: u32 = 1;
pub const RT_FLAG_COUNT_WORK: u32 = 2;
pub const RT_FLAG_RUSSIAN_ROULETTE: u32 = 4;
pub const RT_FLAG_COSINE_DIFFUSE: u32 = 8;
pub const RT_RR_MIN_DEPTH: i32 = 3;
pub enum rt_context {}
pub enum rt_scene {}

extern "C" {
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_context_create(device_id: i32, out: *mut *mut rt_context) -> i32;
    pub fn rt_context_destroy(ctx: *mut rt_context) -> i32;
    pub fn rt_scene_create(ctx: *mut rt_context, desc: *const rt_scene_desc, out: *mut *mut rt_scene) -> i32;
    pub fn rt_scene_destroy(scene: *mut rt_scene) -> i32;
    pub fn rt_scene_create_ex(ctx: *mut rt_context, desc: *const rt_scene_desc, flags: u32, out: *mut *mut rt_scene) -> i32;
    pub fn rt_scene_get_info(scene: *const rt_scene, info: *mut rt_scene_info) -> i32;
    pub fn rt_render(scene: *mut rt_scene, cam: *const rt_camera, params: *const rt_params, accum_rgb: *mut f32,
                     accum_is_device: i32, stats: *mut rt_stats) -> i32;
    pub fn rt_render_static(scene: *mut rt_scene, cam: *const rt_camera, params: *const rt_params, accum_rgb_dev: *mut f32,
                            stats: *mut rt_stats) -> i32;   // no counterpart in the reference: for callers that cannot allocate
    pub fn rt_primary_hits(scene: *mut rt_scene, cam: *const rt_camera, params: *const rt_params,
                           prim_id: *mut i32, t: *mut f32) -> i32;
    pub fn rt_resolve(accum_rgb: *const f32, n_pixels: i64, spp: i32, out_i32: *mut i32, rgb8: *mut u8) -> i32;
    pub fn rt_resolve_device(ctx: *mut rt_context, accum_dev: *const f32, n_pixels: i64, spp: i32, rgb8_dev: *mut u8) -> i32;
    pub fn rt_accum_create(ctx: *mut rt_context, width: i32, height: i32, accum_dev_out: *mut *mut f32) -> i32;
    pub fn rt_accum_destroy(ctx: *mut rt_context, accum_dev: *mut f32) -> i32;
    pub fn rt_accum_download(ctx: *mut rt_context, accum_dev: *const f32, host_out: *mut f32, n_floats: i64) -> i32;
    pub fn rt_reduce_accum(ctxs: *const *mut rt_context, n: i32, accum_dev: *const *mut f32, n_floats: i64, root: i32) -> i32;   // RCCL sum over the GPUs of this process
    pub fn rt_context_stream(ctx: *mut rt_context) -> *mut c_void;
    pub fn rt_params_default(p: *mut rt_params, width: i32, height: i32, spp: i32);
    pub fn rt_bvh_build_host(tri_v: *const f32, n_tri: i32, max_top: i32, ordered: i32, nodes_out: *mut f32, tris_out: *mut f32,
                             n_nodes: *mut i32, n_top: *mut i32, n_orders: *mut i32) -> i32;   // host-only diagnostic
    pub fn rt_bvh_build_host_compact(tri_v: *const f32, n_tri: i32, ordered: i32, cnodes_out: *mut u32, q_params: *mut f32,
                                     n_nodes: *mut i32, n_orders: *mut i32) -> i32;   // host-only diagnostic
}

//! FFI declarations of librt_hip.so (include/rt_hip.h).  New module `src/hip.rs` of the reference crate.
#![allow(non_camel_case_types, dead_code)]
use std::ffi::CStr;
use std::os::raw::{c_char, c_double, c_int, c_void};

#[repr(C)] pub struct RtBuilder { _private: [u8; 0] }
#[repr(C)] pub struct RtScene { _private: [u8; 0] }
pub type RtTex = i32;
pub type RtMat = i32;
pub type RtObj = i32;

#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtCamera {                      // arguments of Camera::new, cameras.rs:38-48
    pub look_from: [f64; 3], pub look_at: [f64; 3], pub v_up: [f64; 3],
    pub vertical_fov_deg: f64, pub aspect_ratio: f64, pub aperture: f64, pub focus_dist: f64,
    pub shutter_open: f64, pub shutter_close: f64,
}
#[repr(C)]
#[derive(Clone, Copy)]
pub struct RtRenderParams {
    pub width: i32, pub height: i32, pub spp: i32, pub sample_offset: i32, pub max_depth: i32,
    pub background: [f32; 3], pub seed: u64, pub paths_per_pass: i32, pub flags: i32,
    pub tile: [i32; 4], pub stripe: [i32; 3], pub reserved_: i32,   // the pixels rendered; all zero = the whole image
}
#[repr(C)]
pub struct RtStats {
    pub rays: u64, pub paths: u64, pub rays_per_bounce: [u64; 64], pub seconds: f64,
    pub kernel_ms_extend: f64, pub extend_launches: u64, pub passes: i32, pub work: [u64; 16],
    pub kernel_ms_trace: f64, pub trace_launches: u64,
    pub multi_gpus: i32, pub multi_comm_ranks: i32, pub multi_tiles: i32, pub reserved_: i32,
    pub rays_settled: u64,
}

extern "C" {
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_device_count(count: *mut c_int) -> c_int;
    pub fn rtb_create(out: *mut *mut RtBuilder) -> c_int;
    pub fn rtb_destroy(b: *mut RtBuilder);
    pub fn rtb_object_count(b: *const RtBuilder) -> c_int;
    pub fn rtb_solid_color(b: *mut RtBuilder, r: c_double, g: c_double, bl: c_double) -> RtTex;
    pub fn rtb_checker(b: *mut RtBuilder, even: RtTex, odd: RtTex) -> RtTex;
    pub fn rtb_noise(b: *mut RtBuilder, scale: c_double, ranvec_768: *const f64, perm_768: *const i32) -> RtTex;
    pub fn rtb_lambertian(b: *mut RtBuilder, albedo: RtTex) -> RtMat;
    pub fn rtb_metal(b: *mut RtBuilder, r: c_double, g: c_double, bl: c_double, fuzz: c_double) -> RtMat;
    pub fn rtb_dielectric(b: *mut RtBuilder, ir: c_double) -> RtMat;
    pub fn rtb_diffuse_light(b: *mut RtBuilder, emit: RtTex) -> RtMat;
    pub fn rtb_isotropic(b: *mut RtBuilder, albedo: RtTex) -> RtMat;
    pub fn rtb_sphere(b: *mut RtBuilder, center: *const f64, radius: c_double, m: RtMat) -> RtObj;
    pub fn rtb_moving_sphere(b: *mut RtBuilder, c0: *const f64, c1: *const f64, t0: c_double, t1: c_double,
                             radius: c_double, m: RtMat) -> RtObj;
    pub fn rtb_rectangle(b: *mut RtBuilder, m: RtMat, x0: c_double, x1: c_double, y0: c_double, y1: c_double,
                         k: c_double, rtype: c_int) -> RtObj;
    pub fn rtb_box(b: *mut RtBuilder, box_min: *const f64, box_max: *const f64, m: RtMat) -> RtObj;
    pub fn rtb_list(b: *mut RtBuilder, children: *const RtObj, n: c_int) -> RtObj;
    pub fn rtb_bvh_node(b: *mut RtBuilder, list: RtObj, time0: c_double, time1: c_double) -> RtObj;
    pub fn rtb_translate(b: *mut RtBuilder, obj: RtObj, offset: *const f64) -> RtObj;
    pub fn rtb_rotate_y(b: *mut RtBuilder, obj: RtObj, angle_deg: c_double) -> RtObj;
    pub fn rtb_constant_medium(b: *mut RtBuilder, boundary: RtObj, density: c_double, albedo: RtTex) -> RtObj;
    pub fn rt_scene_create(b: *const RtBuilder, world: RtObj, device: c_int, out: *mut *mut RtScene) -> c_int;
    pub fn rt_scene_destroy(s: *mut RtScene);
    pub fn rt_render(s: *mut RtScene, cam: *const RtCamera, p: *const RtRenderParams, accum_rgb: *mut f32,
                     accum_sq: *mut f32, stats: *mut RtStats) -> c_int;
    pub fn rt_render_device(s: *mut RtScene, cam: *const RtCamera, p: *const RtRenderParams, d_accum_rgb: *mut f32,
                            d_accum_sq: *mut f32, hip_stream: *mut c_void, stats: *mut RtStats) -> c_int;
    pub fn rt_trace_primary(s: *mut RtScene, cam: *const RtCamera, p: *const RtRenderParams, sample: i32,
                            object_id: *mut i32, t: *mut f32, normal: *mut f32, front_face: *mut i32, uv: *mut f32) -> c_int;
    pub fn rt_tonemap_u8(accum_rgb: *const f32, n_pixels: i32, spp: i32, rgb: *mut u8) -> c_int;
    // several GPUs inside one process (src/main.rs:585-589 lifted across GPUs); see include/rt_hip.h
    pub fn rt_render_multi(scenes: *const *mut RtScene, n_scenes: i32, cam: *const RtCamera, p: *const RtRenderParams, accum_rgb: *mut f32,
                           accum_sq: *mut f32, stats: *mut RtStats) -> c_int;
    pub fn rt_shard_samples(total_spp: i32, n_shards: i32, shard: i32, first: *mut i32, count: *mut i32) -> c_int;
    pub fn rt_shard_tiles(p: *mut RtRenderParams, n_shards: i32, shard: i32, n_pixels: *mut i64) -> c_int;
    pub fn rt_multi_shutdown() -> c_int;
}

/// Negative status -> Err(message of rt_last_error()).  Handles (RtTex / RtMat / RtObj) follow the same rule.
pub fn check(rc: c_int) -> Result<c_int, String> {
    if rc >= 0 { return Ok(rc); }
    let msg = unsafe { CStr::from_ptr(rt_last_error()) }.to_string_lossy().into_owned();
    Err(format!("librt_hip: status {rc}: {msg}"))
}

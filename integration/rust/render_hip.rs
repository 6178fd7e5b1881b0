//! Replacement of the pixel / sample loop of main() (src/main.rs:567-599): flatten, upload, render, tonemap.
use crate::flatten::Flatten;
use crate::hip::{self, check, RtCamera, RtRenderParams};
use crate::{cameras::CameraConfig, utils::{get_pixel, SerdeVector}, Color, HittableList, Vec3};
use image::RgbImage;

fn arr(v: &SerdeVector) -> [f64; 3] { let v: Vec3 = v.clone().into(); [v[0], v[1], v[2]] }   // utils.rs:93-102

pub struct GpuOptions { pub device: i32, pub seed: u64, pub background: [f32; 3] }

pub fn render_on_gpu(world: &HittableList, cam: &CameraConfig, width: u32, height: u32, spp: u32, max_depth: u32,
                     opt: &GpuOptions) -> Result<RgbImage, String> {
    unsafe {
        let mut b = std::ptr::null_mut();
        check(hip::rtb_create(&mut b))?;
        let world_h = check(world.flatten(b));                      // main.rs:520-531: the list BvhNode::new would wrap
        let mut scene = std::ptr::null_mut();
        let rc = world_h.and_then(|w| check(hip::rt_scene_create(b, w, opt.device, &mut scene)));
        hip::rtb_destroy(b);                                        // the scene keeps its own copy
        rc?;
        let camera = RtCamera {
            look_from: arr(&cam.look_from), look_at: arr(&cam.look_at), v_up: arr(&cam.v_up),
            vertical_fov_deg: cam.vertical_fov_deg, aspect_ratio: cam.aspect_ratio, aperture: cam.aperture,
            focus_dist: cam.focus_distance, shutter_open: 0.0, shutter_close: 1.0,          // main.rs:544-545
        };
        let p = RtRenderParams { width: width as i32, height: height as i32, spp: spp as i32, sample_offset: 0,
                                 max_depth: max_depth as i32, background: opt.background, seed: opt.seed,
                                 paths_per_pass: 0, flags: 0, tile: [0; 4], stripe: [0; 3], reserved_: 0 };
        let mut accum = vec![0f32; (width * height * 3) as usize];
        let rc = check(hip::rt_render(scene, &camera, &p, accum.as_mut_ptr(), std::ptr::null_mut(), std::ptr::null_mut()));
        hip::rt_scene_destroy(scene);
        rc?;
        let mut img = RgbImage::new(width, height);
        for (i, px) in accum.chunks_exact(3).enumerate() {          // row 0 of accum is the top row (main.rs:592-597)
            let c = Color::new(px[0] as f64, px[1] as f64, px[2] as f64);
            img.put_pixel(i as u32 % width, i as u32 / width, get_pixel(&c, spp as usize));
        }
        Ok(img)
    }
}

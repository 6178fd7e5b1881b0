//! `Flatten`: re-creates a scene object inside an RtBuilder and returns its handle.
//! Every impl block below goes into the module that owns the type, because it reads private fields.
use crate::hip::{self, RtBuilder};

pub trait Flatten {
    /// Handle of this object in `b` (RtObj, RtMat or RtTex depending on the trait it accompanies); negative = error.
    fn flatten(&self, b: *mut RtBuilder) -> i32;
}
// objects.rs:16   pub trait Hittable: Flatten { ... }
// materials.rs:14 pub trait Scatterable: Flatten + DynClone { ... }
// textures.rs:12  pub trait Textured: Flatten + DynClone { ... }

fn p3(v: &crate::Vec3) -> [f64; 3] { [v[0], v[1], v[2]] }

// ---------------------------------------------------------------- textures.rs
impl Flatten for SolidColor {            // textures.rs:19
    fn flatten(&self, b: *mut RtBuilder) -> i32 {
        unsafe { hip::rtb_solid_color(b, self.color_value[0], self.color_value[1], self.color_value[2]) }
    }
}
impl Flatten for Checker {               // textures.rs:35
    fn flatten(&self, b: *mut RtBuilder) -> i32 {
        let (even, odd) = (self.even.flatten(b), self.odd.flatten(b));
        unsafe { hip::rtb_checker(b, even, odd) }
    }
}
impl Flatten for Noise {                 // textures.rs:73, Perlin tables :99-109
    fn flatten(&self, b: *mut RtBuilder) -> i32 {
        let p = &self.noise;
        let ranvec: Vec<f64> = p.ranvec.iter().flat_map(|v| [v[0], v[1], v[2]]).collect();      // 256 x 3
        let perm: Vec<i32> = p.perm_x.iter().chain(&p.perm_y).chain(&p.perm_z).copied().collect(); // 3 x 256
        unsafe { hip::rtb_noise(b, self.scale, ranvec.as_ptr(), perm.as_ptr()) }
    }
}

// ---------------------------------------------------------------- materials.rs
impl Flatten for Lambertian {            // materials.rs:55
    fn flatten(&self, b: *mut RtBuilder) -> i32 { let t = self.albedo.flatten(b); unsafe { hip::rtb_lambertian(b, t) } }
}
impl Flatten for Metal {                 // materials.rs:109
    fn flatten(&self, b: *mut RtBuilder) -> i32 {
        unsafe { hip::rtb_metal(b, self.albedo[0], self.albedo[1], self.albedo[2], self.fuzz) }
    }
}
impl Flatten for Dielectric {            // materials.rs:159
    fn flatten(&self, b: *mut RtBuilder) -> i32 { unsafe { hip::rtb_dielectric(b, self.ir) } }
}
impl Flatten for DiffuseLight {          // materials.rs:220
    fn flatten(&self, b: *mut RtBuilder) -> i32 { let t = self.emit.flatten(b); unsafe { hip::rtb_diffuse_light(b, t) } }
}
impl Flatten for Isotropic {             // materials.rs:253
    fn flatten(&self, b: *mut RtBuilder) -> i32 { let t = self.albedo.flatten(b); unsafe { hip::rtb_isotropic(b, t) } }
}

// ---------------------------------------------------------------- objects.rs
impl Flatten for Sphere {                // objects.rs:155
    fn flatten(&self, b: *mut RtBuilder) -> i32 {
        let m = self.material.flatten(b);
        unsafe { hip::rtb_sphere(b, p3(&self.center).as_ptr(), self.radius, m) }
    }
}
impl Flatten for MovingSphere {          // objects.rs:257
    fn flatten(&self, b: *mut RtBuilder) -> i32 {
        let m = self.material.flatten(b);
        unsafe {
            hip::rtb_moving_sphere(b, p3(&self.initial_center).as_ptr(), p3(&self.final_center).as_ptr(),
                                   self.initial_time, self.final_time, self.radius, m)
        }
    }
}
impl Flatten for Rectangle {             // objects.rs:346; RectangleType :339-343 -> 0 = Xy, 1 = Xz, 2 = Yz
    fn flatten(&self, b: *mut RtBuilder) -> i32 {
        let m = self.material.flatten(b);
        let t = match self.rtype { RectangleType::Xy => 0, RectangleType::Xz => 1, RectangleType::Yz => 2 };
        unsafe { hip::rtb_rectangle(b, m, self.x0, self.x1, self.y0, self.y1, self.k, t) }
    }
}
impl Flatten for BoxObj {                // objects.rs:433; the six sides share one material (objects.rs:445-500)
    fn flatten(&self, b: *mut RtBuilder) -> i32 {
        // BoxObj keeps only its sides; store the material passed to BoxObj::new in a new field `material`
        let m = self.material.flatten(b);
        unsafe { hip::rtb_box(b, p3(&self.box_min).as_ptr(), p3(&self.box_max).as_ptr(), m) }
    }
}
impl Flatten for HittableList {          // objects.rs:25
    fn flatten(&self, b: *mut RtBuilder) -> i32 {
        let kids: Vec<i32> = self.0.iter().map(|o| o.flatten(b)).collect();
        if let Some(bad) = kids.iter().find(|h| **h < 0) { return *bad; }
        unsafe { hip::rtb_list(b, kids.as_ptr(), kids.len() as i32) }
    }
}

// ---------------------------------------------------------------- bvh.rs
impl Flatten for BvhNode {               // bvh.rs:82: a BVH is only a spatial index over its leaves; the library rebuilds its own
    fn flatten(&self, b: *mut RtBuilder) -> i32 {
        let mut kids = vec![self.left.flatten(b)];
        if let Some(r) = &self.right { kids.push(r.flatten(b)); }
        unsafe { hip::rtb_list(b, kids.as_ptr(), kids.len() as i32) }
    }
}

// ---------------------------------------------------------------- transrot.rs
impl Flatten for Translate {             // transrot.rs:10
    fn flatten(&self, b: *mut RtBuilder) -> i32 {
        let o = self.obj.flatten(b);
        unsafe { hip::rtb_translate(b, o, p3(&self.offset).as_ptr()) }
    }
}
impl Flatten for RotateY {               // transrot.rs:42: keeps sin and cos of the angle given to RotateY::new (:53-56)
    fn flatten(&self, b: *mut RtBuilder) -> i32 {
        let o = self.obj.flatten(b);
        unsafe { hip::rtb_rotate_y(b, o, self.sin_theta.atan2(self.cos_theta).to_degrees()) }
    }
}

// ---------------------------------------------------------------- mediums.rs
impl Flatten for ConstantMedium {        // mediums.rs:14: phase_function = Isotropic::new(albedo) (:27), neg_inv_density = -1/density (:28)
    fn flatten(&self, b: *mut RtBuilder) -> i32 {
        let boundary = self.boundary.flatten(b);
        // the albedo texture is inside the Isotropic; keep a clone of it in a new field `albedo` at construction
        let tex = self.albedo.flatten(b);
        unsafe { hip::rtb_constant_medium(b, boundary, -1.0 / self.neg_inv_density, tex) }
    }
}

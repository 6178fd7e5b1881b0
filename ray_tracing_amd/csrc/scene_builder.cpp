// scene_builder.cpp -- Builder methods (one per reference constructor), the canned scenes of
// main.rs:13-451 and the thin-lens camera frame of cameras.rs:38-79.
#include "scene_builder.h"
#include <cmath>
#include <cstdlib>

namespace rt {

static const double kPi = 3.14159265358979323846;

// ------------------------------------------------------------------ textures
Texture Builder::solid_color(Color c) {
    TexNode t = {}; t.type = RT_TEX_SOLID; t.rgb[0] = c.x; t.rgb[1] = c.y; t.rgb[2] = c.z; t.even = t.odd = t.perlin = -1;
    textures.push_back(t); return (int)textures.size() - 1;
}
Texture Builder::checker(Texture even, Texture odd) {
    if (!valid_tex(even) || !valid_tex(odd)) return fail("checker: bad texture handle");
    TexNode t = {}; t.type = RT_TEX_CHECKER; t.even = even; t.odd = odd; t.perlin = -1;
    textures.push_back(t); return (int)textures.size() - 1;
}
Texture Builder::checker_from_solid_colors(Color c1, Color c2) {
    Texture e = solid_color(c1), o = solid_color(c2);
    return checker(e, o);
}
Texture Builder::noise_from_table(double scale, const PerlinTable& tab) {
    perlins.push_back(tab);
    TexNode t = {}; t.type = RT_TEX_NOISE; t.scale = scale; t.even = t.odd = -1; t.perlin = (int)perlins.size() - 1;
    textures.push_back(t); return (int)textures.size() - 1;
}
Texture Builder::noise(double scale, SceneRng& rng) {
    PerlinTable tab;
    for (int i = 0; i < 256; ++i)               // textures.rs:112-115: 256 x gen_random(3, -1, 1), NOT normalised
        for (int c = 0; c < 3; ++c) tab.ranvec[i][c] = rng.range(-1.0, 1.0);
    for (int k = 0; k < 3; ++k) {               // textures.rs:175-192: identity then Fisher-Yates from the top
        for (int i = 0; i < 256; ++i) tab.perm[k][i] = i;
        for (int i = 255; i >= 1; --i) { uint32_t j = rng.upto((uint32_t)i); int32_t tmp = tab.perm[k][i]; tab.perm[k][i] = tab.perm[k][j]; tab.perm[k][j] = tmp; }
    }
    return noise_from_table(scale, tab);
}
Texture Builder::noise_clone(Texture t) {
    if (!valid_tex(t) || textures[t].type != RT_TEX_NOISE) return fail("noise_clone: not a Noise texture");
    TexNode c = textures[t]; textures.push_back(c); return (int)textures.size() - 1;
}

// ------------------------------------------------------------------ materials
static MatNode mat_node(int type, int tex) { MatNode m = {}; m.type = type; m.tex = tex; m.ir = 1.0; return m; }
Material Builder::lambertian_from_texture(Texture t) {
    if (!valid_tex(t)) return fail("lambertian: bad texture handle");
    materials.push_back(mat_node(RT_MAT_LAMBERTIAN, t)); return (int)materials.size() - 1;
}
Material Builder::lambertian(Color a) { return lambertian_from_texture(solid_color(a)); }
Material Builder::metal(Color a, double fuzz) {
    MatNode m = mat_node(RT_MAT_METAL, -1); m.albedo[0] = a.x; m.albedo[1] = a.y; m.albedo[2] = a.z; m.fuzz = fuzz;
    materials.push_back(m); return (int)materials.size() - 1;
}
Material Builder::dielectric(double ir) {
    MatNode m = mat_node(RT_MAT_DIELECTRIC, -1); m.ir = ir; materials.push_back(m); return (int)materials.size() - 1;
}
Material Builder::diffuse_light(Texture e) {
    if (!valid_tex(e)) return fail("diffuse_light: bad texture handle");
    materials.push_back(mat_node(RT_MAT_LIGHT, e)); return (int)materials.size() - 1;
}
Material Builder::diffuse_light_from_color(Color c) { return diffuse_light(solid_color(c)); }
Material Builder::isotropic(Texture a) {
    if (!valid_tex(a)) return fail("isotropic: bad texture handle");
    materials.push_back(mat_node(RT_MAT_ISOTROPIC, a)); return (int)materials.size() - 1;
}

// ------------------------------------------------------------------ objects
static ObjNode obj_node(int type) { ObjNode o; o.type = type; o.mat = -1; o.id = -1; o.rect_type = 0; for (double& v : o.p) v = 0; return o; }

HittableObj Builder::sphere(Point c, double r, Material m) {
    if (!valid_mat(m)) return fail("sphere: bad material handle");
    ObjNode o = obj_node(OBJ_SPHERE); o.mat = m; o.id = next_id++;
    o.p[0] = c.x; o.p[1] = c.y; o.p[2] = c.z; o.p[3] = r;
    objects.push_back(o); return (int)objects.size() - 1;
}
HittableObj Builder::moving_sphere(Point c0, Point c1, double t0, double t1, double r, Material m) {
    if (!valid_mat(m)) return fail("moving_sphere: bad material handle");
    ObjNode o = obj_node(OBJ_MSPHERE); o.mat = m; o.id = next_id++;
    o.p[0] = c0.x; o.p[1] = c0.y; o.p[2] = c0.z; o.p[3] = c1.x; o.p[4] = c1.y; o.p[5] = c1.z; o.p[6] = t0; o.p[7] = t1; o.p[8] = r;
    objects.push_back(o); return (int)objects.size() - 1;
}
HittableObj Builder::rectangle(Material m, double x0, double x1, double y0, double y1, double k, RectangleType t) {
    if (!valid_mat(m)) return fail("rectangle: bad material handle");
    ObjNode o = obj_node(OBJ_RECT); o.mat = m; o.id = next_id++; o.rect_type = (int)t;
    o.p[0] = x0; o.p[1] = x1; o.p[2] = y0; o.p[3] = y1; o.p[4] = k;
    objects.push_back(o); return (int)objects.size() - 1;
}
HittableObj Builder::box(Point mn, Point mx, Material m) {  // objects.rs:442-507, side order kept
    if (!valid_mat(m)) return fail("box: bad material handle");
    int s[6];
    s[0] = rectangle(m, mn.x, mx.x, mn.y, mx.y, mx.z, Xy);
    s[1] = rectangle(m, mn.x, mx.x, mn.y, mx.y, mn.z, Xy);
    s[2] = rectangle(m, mn.x, mx.x, mn.z, mx.z, mx.y, Xz);
    s[3] = rectangle(m, mn.x, mx.x, mn.z, mx.z, mn.y, Xz);
    s[4] = rectangle(m, mn.y, mx.y, mn.z, mx.z, mx.x, Yz);
    s[5] = rectangle(m, mn.y, mx.y, mn.z, mx.z, mn.x, Yz);
    ObjNode o = obj_node(OBJ_BOX); o.mat = m; o.id = objects[s[0]].id;
    o.p[0] = mn.x; o.p[1] = mn.y; o.p[2] = mn.z; o.p[3] = mx.x; o.p[4] = mx.y; o.p[5] = mx.z;
    o.children.assign(s, s + 6);
    objects.push_back(o); return (int)objects.size() - 1;
}
HittableObj Builder::list(const std::vector<HittableObj>& ch) {
    for (int c : ch) if (!valid_obj(c)) return fail("list: bad object handle");
    ObjNode o = obj_node(OBJ_LIST); o.children = ch;
    objects.push_back(o); return (int)objects.size() - 1;
}
HittableObj Builder::bvh_node(HittableObj l, double t0, double t1) {
    // bvh.rs:100-121: an empty list recurses forever in the reference; here it is an error.
    if (!valid_obj(l) || objects[l].type != OBJ_LIST) return fail("bvh_node: needs a HittableList");
    if (objects[l].children.empty()) return fail("bvh_node: empty HittableList");
    ObjNode o = obj_node(OBJ_BVH); o.children.push_back(l); o.p[0] = t0; o.p[1] = t1;
    objects.push_back(o); return (int)objects.size() - 1;
}
HittableObj Builder::translate(HittableObj obj, Vec3 off) {
    if (!valid_obj(obj)) return fail("translate: bad object handle");
    ObjNode o = obj_node(OBJ_TRANSLATE); o.children.push_back(obj); o.p[0] = off.x; o.p[1] = off.y; o.p[2] = off.z;
    objects.push_back(o); return (int)objects.size() - 1;
}
HittableObj Builder::rotate_y(HittableObj obj, double deg) {
    if (!valid_obj(obj)) return fail("rotate_y: bad object handle");
    ObjNode o = obj_node(OBJ_ROTATE_Y); o.children.push_back(obj);
    double rad = deg * (kPi / 180.0);  // transrot.rs:54-56
    o.p[0] = deg; o.p[1] = std::sin(rad); o.p[2] = std::cos(rad);
    objects.push_back(o); return (int)objects.size() - 1;
}
HittableObj Builder::constant_medium(HittableObj boundary, double density, Texture albedo) {
    if (!valid_obj(boundary)) return fail("constant_medium: bad boundary handle");
    Material phase = isotropic(albedo);  // mediums.rs:26
    if (phase < 0) return -1;
    ObjNode o = obj_node(OBJ_MEDIUM); o.children.push_back(boundary); o.mat = phase; o.id = next_id++; o.p[0] = density;
    objects.push_back(o); return (int)objects.size() - 1;
}
HittableObj Builder::constant_medium_from_color(HittableObj boundary, double density, Color c) {
    return constant_medium(boundary, density, solid_color(c));
}

// ------------------------------------------------------------------ canned scenes
HittableObj random_scene(Builder& b, SceneRng& rng) {  // main.rs:13-79
    std::vector<HittableObj> w;
    Texture ck = b.checker_from_solid_colors(vec3(0.2, 0.3, 0.1), vec3(0.9, 0.9, 0.9));
    w.push_back(b.sphere(vec3(0, -1000, 0), 1000, b.lambertian_from_texture(ck)));
    for (int a = -11; a < 11; ++a)
        for (int c = -11; c < 11; ++c) {
            double choose = rng.gen();
            double cx = a + 0.9 * rng.gen();
            double cz = c + 0.9 * rng.gen();
            Point center = vec3(cx, 0.2, cz);
            double dx = cx - 4.0, dz = cz;
            if (std::sqrt(dx * dx + dz * dz) > 0.9) {
                if (choose < 0.8) {  // diffuse, moving
                    double a1[3], a2[3];
                    for (double& v : a1) v = rng.gen();
                    for (double& v : a2) v = rng.gen();
                    Material m = b.lambertian(vec3(a1[0] * a2[0], a1[1] * a2[1], a1[2] * a2[2]));
                    Point c2 = vec3(cx, 0.2 + rng.range(0.0, 0.5), cz);
                    w.push_back(b.moving_sphere(center, c2, 0.0, 1.0, 0.2, m));
                } else if (choose < 0.95) {  // metal
                    double r = rng.range(0.5, 1.0), g = rng.range(0.5, 1.0), bl = rng.range(0.5, 1.0);
                    double fuzz = rng.range(0.0, 0.5);
                    w.push_back(b.sphere(center, 0.2, b.metal(vec3(r, g, bl), fuzz)));
                } else w.push_back(b.sphere(center, 0.2, b.dielectric(1.5)));
            }
        }
    w.push_back(b.sphere(vec3(0, 1, 0), 1.0, b.dielectric(1.5)));
    w.push_back(b.sphere(vec3(-4, 1, 0), 1.0, b.lambertian(vec3(0.4, 0.2, 0.1))));
    w.push_back(b.sphere(vec3(4, 1, 0), 1.0, b.metal(vec3(0.7, 0.6, 0.5), 0.0)));
    return b.list(w);
}

// Stress scene for deep-BVH traversal (not in the reference): n small spheres over a 2000 x 2000 field above a huge
// ground sphere, materials drawn 80 / 15 / 5 % Lambertian / metal / glass with the parameter rules of main.rs:43-66.
HittableObj synthetic_scene(Builder& b, int n, SceneRng& rng) {
    std::vector<HittableObj> w;
    w.reserve((size_t)n + 1);
    w.push_back(b.sphere(vec3(0, -100000, 0), 100000, b.lambertian(vec3(0.5, 0.5, 0.5))));
    for (int i = 0; i < n; ++i) {
        double choose = rng.gen();
        double cx = rng.range(-1000.0, 1000.0), cy = rng.range(0.0, 200.0), cz = rng.range(-1000.0, 1000.0);
        double r = rng.range(0.2, 2.0);
        Material m;
        if (choose < 0.8) {
            double a1[3], a2[3];
            for (double& v : a1) v = rng.gen();
            for (double& v : a2) v = rng.gen();
            m = b.lambertian(vec3(a1[0] * a2[0], a1[1] * a2[1], a1[2] * a2[2]));
        } else if (choose < 0.95) {
            double cr = rng.range(0.5, 1.0), cg = rng.range(0.5, 1.0), cb = rng.range(0.5, 1.0);
            double fuzz = rng.range(0.0, 0.5);
            m = b.metal(vec3(cr, cg, cb), fuzz);
        } else m = b.dielectric(1.5);
        w.push_back(b.sphere(vec3(cx, cy, cz), r, m));
    }
    return b.list(w);
}

HittableObj two_spheres(Builder& b) {  // main.rs:81-100
    Texture ck = b.checker_from_solid_colors(vec3(0.2, 0.3, 0.1), vec3(0.9, 0.9, 0.9));
    std::vector<HittableObj> w;
    w.push_back(b.sphere(vec3(0, -10, 0), 10, b.lambertian_from_texture(ck)));
    w.push_back(b.sphere(vec3(0, 10, 0), 10, b.lambertian_from_texture(ck)));
    return b.list(w);
}

static void perlin_pair(Builder& b, SceneRng& rng, std::vector<HittableObj>& w) {  // main.rs:105-115, 123-133
    Texture pertext = b.noise(4.0, rng);
    Texture clone = b.noise_clone(pertext);
    w.push_back(b.sphere(vec3(0, -1000, 0), 1000, b.lambertian_from_texture(clone)));
    w.push_back(b.sphere(vec3(0, 2, 0), 2, b.lambertian_from_texture(pertext)));
}
HittableObj two_perlin_spheres(Builder& b, SceneRng& rng) {  // main.rs:102-118
    std::vector<HittableObj> w; perlin_pair(b, rng, w); return b.list(w);
}
HittableObj simple_light(Builder& b, SceneRng& rng) {  // main.rs:120-147
    std::vector<HittableObj> w; perlin_pair(b, rng, w);
    w.push_back(b.rectangle(b.diffuse_light_from_color(vec3(4, 4, 4)), 3, 5, 1, 3, -2, Xy));
    return b.list(w);
}

static Material cornell_walls(Builder& b, std::vector<HittableObj>& w, Color light_c, double lx0, double lx1, double lz0, double lz1) {
    Material red = b.lambertian(vec3(0.65, 0.05, 0.05));
    Material white = b.lambertian(vec3(0.73, 0.73, 0.73));
    Material green = b.lambertian(vec3(0.12, 0.45, 0.15));
    Material light = b.diffuse_light_from_color(light_c);
    w.push_back(b.rectangle(green, 0, 555, 0, 555, 555, Yz));
    w.push_back(b.rectangle(red, 0, 555, 0, 555, 0, Yz));
    w.push_back(b.rectangle(light, lx0, lx1, lz0, lz1, 554, Xz));
    w.push_back(b.rectangle(white, 0, 555, 0, 555, 0, Xz));
    w.push_back(b.rectangle(white, 0, 555, 0, 555, 555, Xz));
    w.push_back(b.rectangle(white, 0, 555, 0, 555, 555, Xy));
    return white;
}
HittableObj cornell_box(Builder& b) {  // main.rs:149-239
    std::vector<HittableObj> w;
    Material white = cornell_walls(b, w, vec3(15, 15, 15), 213, 343, 227, 332);
    HittableObj box1 = b.box(vec3(0, 0, 0), vec3(165, 330, 165), white);
    w.push_back(b.translate(b.rotate_y(box1, 15.0), vec3(265, 0, 295)));
    HittableObj box2 = b.box(vec3(0, 0, 0), vec3(165, 165, 165), white);
    w.push_back(b.translate(b.rotate_y(box2, -18.0), vec3(130, 0, 65)));
    return b.list(w);
}
HittableObj cornell_smoke(Builder& b) {  // main.rs:241-330
    std::vector<HittableObj> w;
    Material white = cornell_walls(b, w, vec3(7, 7, 7), 113, 443, 127, 432);
    HittableObj box1 = b.translate(b.rotate_y(b.box(vec3(0, 0, 0), vec3(165, 330, 165), white), 15.0), vec3(265, 0, 295));
    HittableObj box2 = b.translate(b.rotate_y(b.box(vec3(0, 0, 0), vec3(165, 165, 165), white), -18.0), vec3(130, 0, 65));
    w.push_back(b.constant_medium_from_color(box1, 0.01, vec3(0, 0, 0)));
    w.push_back(b.constant_medium_from_color(box2, 0.01, vec3(1, 1, 1)));
    return b.list(w);
}
HittableObj final_scene(Builder& b, SceneRng& rng) {  // main.rs:332-451
    std::vector<HittableObj> boxes1;
    Material ground = b.lambertian(vec3(0.48, 0.83, 0.53));
    for (int i = 0; i < 20; ++i)
        for (int j = 0; j < 20; ++j) {
            double w = 100.0, x0 = -1000.0 + i * w, z0 = -1000.0 + j * w, y0 = 0.0;
            double x1 = x0 + w, y1 = rng.range(1.0, 101.0), z1 = z0 + w;
            boxes1.push_back(b.box(vec3(x0, y0, z0), vec3(x1, y1, z1), ground));
        }
    std::vector<HittableObj> o;
    o.push_back(b.bvh_node(b.list(boxes1), 0.0, 1.0));
    o.push_back(b.rectangle(b.diffuse_light_from_color(vec3(7, 7, 7)), 123, 423, 147, 412, 554, Xz));
    o.push_back(b.moving_sphere(vec3(400, 400, 200), vec3(430, 400, 200), 0.0, 1.0, 50.0, b.lambertian(vec3(0.7, 0.3, 0.1))));
    o.push_back(b.sphere(vec3(260, 150, 45), 50.0, b.dielectric(1.5)));
    o.push_back(b.sphere(vec3(0, 150, 145), 50.0, b.metal(vec3(0.8, 0.8, 0.9), 1.0)));
    o.push_back(b.sphere(vec3(360, 150, 145), 70.0, b.dielectric(1.5)));                        // boundary.clone()
    HittableObj boundary = b.sphere(vec3(360, 150, 145), 70.0, b.dielectric(1.5));
    o.push_back(b.constant_medium_from_color(boundary, 0.2, vec3(0.2, 0.4, 0.9)));
    HittableObj fog = b.sphere(vec3(0, 0, 0), 5000.0, b.dielectric(1.5));
    o.push_back(b.constant_medium_from_color(fog, 0.0001, vec3(1, 1, 1)));
    o.push_back(b.sphere(vec3(400, 200, 400), 100.0, b.lambertian(vec3(0, 1, 0))));
    o.push_back(b.sphere(vec3(220, 280, 300), 80.0, b.lambertian_from_texture(b.noise(0.1, rng))));
    std::vector<HittableObj> boxes2;
    Material white = b.lambertian(vec3(0.73, 0.73, 0.73));
    for (int i = 0; i < 1000; ++i) {
        double x = rng.range(0.0, 165.0), y = rng.range(0.0, 165.0), z = rng.range(0.0, 165.0);
        boxes2.push_back(b.sphere(vec3(x, y, z), 10.0, white));
    }
    HittableObj cluster = b.bvh_node(b.list(boxes2), 0.0, 1.0);
    o.push_back(b.translate(b.rotate_y(cluster, 15.0), vec3(-100, 270, 395)));
    return b.list(o);
}

HittableObj canned_scene(Builder& b, const std::string& type, uint64_t seed) {  // main.rs:520-528
    SceneRng rng(seed);
    if (type == "Random") return random_scene(b, rng);
    if (type == "TwoSpheres") return two_spheres(b);
    if (type == "TwoPerlinSpheres") return two_perlin_spheres(b, rng);
    if (type == "SimpleLight") return simple_light(b, rng);
    if (type == "Cornell") return cornell_box(b);
    if (type == "CornellSmoke") return cornell_smoke(b);
    if (type == "FinalScene") return final_scene(b, rng);
    if (type.compare(0, 10, "Synthetic:") == 0) {
        long n = std::strtol(type.c_str() + 10, nullptr, 10);
        if (n < 1 || n > 50000000) { b.error = "Synthetic:<n> needs 1 <= n <= 50000000"; return -1; }
        return synthetic_scene(b, (int)n, rng);
    }
    b.error = "unknown scene type '" + type + "'";
    return -1;
}

HittableObj scene_from_config(Builder& b, const Config& cfg, uint64_t seed) {
    if (cfg.scene_type != "List") return canned_scene(b, cfg.scene_type, seed);
    SceneRng rng(seed);                      // objects.rs:56-63, 170-176; materials.rs:45-51, and the typed items of config.cpp
    return list_scene_from_yaml(b, cfg.source, rng);
}

// ------------------------------------------------------------------ camera (cameras.rs:38-79)
static Vec3 sub(Vec3 a, Vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
static Vec3 scale(double s, Vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }
static Vec3 crossp(Vec3 a, Vec3 b) { return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static Vec3 unit(Vec3 a) { double n = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); return vec3(a.x / n, a.y / n, a.z / n); }

void make_camera(const CameraDesc& d, RtCameraDev& c) {
    double theta = d.vertical_fov_deg * (kPi / 180.0);
    double h = std::tan(theta / 2.0);
    double vh = 2.0 * h, vw = d.aspect_ratio * vh;
    Vec3 w = unit(sub(d.look_from, d.look_at));
    Vec3 u = unit(crossp(d.v_up, w));
    Vec3 v = crossp(w, u);
    Vec3 horizontal = scale(d.focus_dist * vw, u), vertical = scale(d.focus_dist * vh, v);
    // dir0 = lower_left_corner - origin = -horizontal/2 - vertical/2 - focus_dist*w   (cameras.rs:63, :95)
    Vec3 dir0 = vec3(-horizontal.x / 2 - vertical.x / 2 - d.focus_dist * w.x,
                     -horizontal.y / 2 - vertical.y / 2 - d.focus_dist * w.y,
                     -horizontal.z / 2 - vertical.z / 2 - d.focus_dist * w.z);
    const Vec3* src[6] = {&d.look_from, &dir0, &horizontal, &vertical, &u, &v};
    float* dst[6] = {c.origin, c.dir0, c.horizontal, c.vertical, c.u, c.v};
    for (int i = 0; i < 6; ++i) { dst[i][0] = (float)src[i]->x; dst[i][1] = (float)src[i]->y; dst[i][2] = (float)src[i]->z; }
    c.lens_radius = (float)(d.aperture / 2.0);
    c.shutter_open = (float)d.shutter_open; c.shutter_close = (float)d.shutter_close;
}

int Config::image_height() const { return (int)std::llround((double)image_width / camera.aspect_ratio); }

}  // namespace rt

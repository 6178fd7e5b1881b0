// scene_builder.h -- host-side scene graph with the constructor API of the reference's prelude
// (lib.rs:23-35).  A Builder owns every texture, material and object as a small tagged node; the
// objects form the same tree the Rust code builds out of Box<dyn Hittable>.  flatten() (flatten.cpp)
// turns a tree into the flat arrays of rt_types.h.  Nothing here touches the GPU.
#pragma once
#include <stdint.h>
#include <string>
#include <vector>
#include "rt_types.h"

namespace rt {

struct Vec3 { double x, y, z; };
typedef Vec3 Point;
typedef Vec3 Color;
inline Vec3 vec3(double x, double y, double z) { Vec3 v = {x, y, z}; return v; }

// Scene-construction RNG: SplitMix64 in place of the reference's thread_rng (OS-seeded), so that
// scenes are reproducible.  gen() = rng.gen::<f64>(), range() = gen_range(a..b), upto() = gen_range(0..=i).
struct SceneRng {
    uint64_t s;
    explicit SceneRng(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double gen() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    double range(double a, double b) { return a + (b - a) * gen(); }
    uint32_t upto(uint32_t i) { return (uint32_t)(next() % (uint64_t)(i + 1)); }
};

enum ObjType { OBJ_SPHERE, OBJ_MSPHERE, OBJ_RECT, OBJ_BOX, OBJ_LIST, OBJ_BVH, OBJ_TRANSLATE, OBJ_ROTATE_Y, OBJ_MEDIUM };
enum RectangleType { Xy = 0, Xz = 1, Yz = 2 };  // objects.rs:339-343

struct PerlinTable { double ranvec[256][3]; int32_t perm[3][256]; };  // textures.rs:99-109
struct TexNode { int type; double rgb[3]; double scale; int even, odd, perlin; };
struct MatNode { int type; int tex; double albedo[3]; double fuzz, ir; };
struct ObjNode {
    int type;
    int mat;              // prims: material; medium: its Isotropic phase function
    int id;               // creation-order id of a prim / medium (a box owns the 6 ids of its sides)
    double p[9];          // sphere: c, r | msphere: c0, c1, t0, t1, r | rect: x0,x1,y0,y1,k | box: min,max
                          // translate: offset | rotate_y: deg, sin, cos | medium: density
    int rect_type;
    std::vector<int> children;
};

typedef int Texture;      // handles: index into the Builder's tables, negative = error
typedef int Material;
typedef int HittableObj;

class Builder {
public:
    std::vector<TexNode> textures;
    std::vector<PerlinTable> perlins;
    std::vector<MatNode> materials;
    std::vector<ObjNode> objects;
    int next_id = 0;
    std::string error;

    // textures.rs
    Texture solid_color(Color c);                                   // SolidColor::new            :24
    Texture checker(Texture even, Texture odd);                     // Checker::new               :40
    Texture checker_from_solid_colors(Color c1, Color c2);          // Checker::from_solid_colors :45
    Texture noise(double scale, SceneRng& rng);                     // Noise::new (draws a table) :80, :110-128
    Texture noise_from_table(double scale, const PerlinTable& t);   // a Noise whose table is given
    Texture noise_clone(Texture t);                                 // pertext.clone(): same table values
    // materials.rs
    Material lambertian(Color albedo);                              // Lambertian::new            :60
    Material lambertian_from_texture(Texture t);                    // Lambertian::from_texture   :67
    Material metal(Color albedo, double fuzz);                      // Metal::new                 :116
    Material dielectric(double ir);                                 // Dielectric::new            :164
    Material diffuse_light(Texture emit);                           // DiffuseLight::new          :230
    Material diffuse_light_from_color(Color c);                     // DiffuseLight::from_color   :236
    Material isotropic(Texture albedo);                             // Isotropic::new             :263
    // objects.rs / transrot.rs / mediums.rs / bvh.rs
    HittableObj sphere(Point center, double radius, Material m);    // Sphere::new                :161
    HittableObj moving_sphere(Point c0, Point c1, double t0, double t1, double radius, Material m);  // MovingSphere::new :266
    HittableObj rectangle(Material m, double x0, double x1, double y0, double y1, double k, RectangleType t);  // Rectangle::new :363
    HittableObj box(Point mn, Point mx, Material m);                // BoxObj::new                :442
    HittableObj list(const std::vector<HittableObj>& children);     // HittableList               :25
    HittableObj bvh_node(HittableObj list, double time0, double time1);  // BvhNode::new          bvh.rs:89
    HittableObj translate(HittableObj obj, Vec3 offset);            // Translate::new             transrot.rs:15
    HittableObj rotate_y(HittableObj obj, double angle_deg);        // RotateY::new               transrot.rs:53
    HittableObj constant_medium(HittableObj boundary, double density, Texture albedo);  // ConstantMedium::new mediums.rs:23
    HittableObj constant_medium_from_color(HittableObj boundary, double density, Color c);  // mediums.rs:33

    bool valid_tex(int t) const { return t >= 0 && t < (int)textures.size(); }
    bool valid_mat(int m) const { return m >= 0 && m < (int)materials.size(); }
    bool valid_obj(int o) const { return o >= 0 && o < (int)objects.size(); }
private:
    int fail(const char* msg) { error = msg; return -1; }
};

// Canned scenes of main.rs:13-451; each returns the HittableList main.rs:520-529 would build.
HittableObj random_scene(Builder& b, SceneRng& rng);
HittableObj two_spheres(Builder& b);
HittableObj two_perlin_spheres(Builder& b, SceneRng& rng);
HittableObj simple_light(Builder& b, SceneRng& rng);
HittableObj cornell_box(Builder& b);
HittableObj cornell_smoke(Builder& b);
HittableObj final_scene(Builder& b, SceneRng& rng);
HittableObj synthetic_scene(Builder& b, int n, SceneRng& rng);   // "Synthetic:<n>": n random spheres (traversal stress, not in the reference)
HittableObj canned_scene(Builder& b, const std::string& type, uint64_t scene_seed);  // -1: unknown type

// ---- Config (main.rs:453-494, cameras.rs:10-18, objects.rs:104-107,248-253, materials.rs:34-40)
struct CameraConfig { Vec3 look_from, look_at, v_up; double vertical_fov_deg, aspect_ratio, aperture, focus_distance; };
struct Config {
    int image_width = 400, samples_per_pixel = 500, max_depth = 50;   // main.rs:468-480
    CameraConfig camera;
    double shutter_open = 0.0, shutter_close = 1.0;   // main.rs:544-545 hard-codes 0 and 1; optional camera keys here
    Vec3 background = {0.0, 0.0, 0.0};                // main.rs:534 hard-codes black; optional top-level key here
    std::string scene_type;                 // Random | TwoSpheres | ... | List
    std::string source;                     // the YAML text (a List scene is built from it)
    int n_list_objects = 0;                 // top-level items of a List scene
    int image_height() const;               // main.rs:515-516
};
// Parses the YAML subset the reference's configs use.  Returns false and sets err on failure
// (the reference panics with "Could not open config file." / "Could not read values.").
bool load_config(const std::string& path, Config& out, std::string& err);
bool parse_config(const std::string& text, Config& out, std::string& err);
HittableObj scene_from_config(Builder& b, const Config& cfg, uint64_t scene_seed);
// The `objects` of a List scene (the reference's spheres-only form and the typed items described in config.cpp).
HittableObj list_scene_from_yaml(Builder& b, const std::string& yaml_text, SceneRng& rng);

// ---- Camera (cameras.rs:38-79), evaluated in f64 on the host
struct CameraDesc {
    Vec3 look_from, look_at, v_up;
    double vertical_fov_deg, aspect_ratio, aperture, focus_dist, shutter_open, shutter_close;
};
void make_camera(const CameraDesc& d, RtCameraDev& out);

// ---- Flattened scene (host arrays in the layout of rt_types.h)
struct FlatScene {
    std::vector<rt_f4> nodes, spheres, mspheres, rects, boxes, instances, ops, media, materials, textures, perlin_vec;
    std::vector<rt_i2> sphere_meta, msphere_meta, rect_meta;
    std::vector<uint8_t> perlin_perm;
    std::vector<rt_f4> tlas_items;   // deferred top level (flatten(..., defer_top)): 2 records per item, see lbvh.h
    std::vector<rt_f4> wnodes;       // the 4-wide quantised form of `nodes` (build_wide_bvh)
    std::vector<int32_t> winst_roots;
    std::vector<int32_t> wsrc;       // per wide node and child slot: the binary node << 1 | which child its box came from, -1 = absent
    int wroot = -1, wmax_stack = 0;
    int root = -1, max_stack = 0, tlas_depth = 0, blas_depth = 0, n_ids = 0;
    double world_box[6] = {0, 0, 0, 0, 0, 0};
    RtSceneView view() const;  // host pointers
};
// Flattens the tree under `world` (what main.rs:531 wraps in BvhNode::new).  Returns false and
// sets err for trees the flat form cannot express (e.g. an object without a bounding box).
// defer_top: the top-level BVH is not built; out.tlas_items holds its leaves (unless there are fewer than two, in which
// case the normal builder runs) and out.root stays -1 until lbvh_finish_host or the device builder appends the nodes.
bool flatten(const Builder& b, HittableObj world, FlatScene& out, std::string& err, bool defer_top = false);
// Collapses the finished binary tree (nodes, root, instances) into the 4-wide quantised tree (wide_bvh.cpp).
void build_wide_bvh(FlatScene& f);
// Moves spheres / moving spheres of a flattened scene (by object id, in creation order) and recomputes the boxes of its trees
// (refit.h) on the host.  A primitive that bounds a constant medium cannot be moved (the medium record holds a copy of it).
void build_prim_index(const FlatScene& f, std::vector<int32_t> of_id[2]);
bool set_prims_host(FlatScene& f, const std::vector<int32_t> of_id[2], int kind, const int32_t* ids, int n, const double* values,
                    std::vector<int32_t>& touched, std::string& err);
bool refit_host(FlatScene& f, std::string& err);
// Builds the deferred top level as a linear BVH on the host, with the per-element steps of lbvh.h in plain loops.
bool lbvh_finish_host(FlatScene& f, std::string& err);
// Scene box of the deferred items -> Morton frame.
struct LbvhFrameHost { float lo[3], inv_ext[3]; };
LbvhFrameHost lbvh_frame(const FlatScene& f);

}  // namespace rt

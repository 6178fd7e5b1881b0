// capi.cpp -- the host-only half of include/rt_hip.h: builder entry points (one per reference
// constructor), config loading, host flatten, tonemap and PPM output.  No HIP calls here.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cctype>
#include <string>
#include <vector>
#include "capi_internal.h"
#include "frame.h"

thread_local std::string g_rt_error;
int rt_set_error(int code, const std::string& msg) { g_rt_error = msg; return code; }

static rt::Vec3 v3d(const double* p) { return rt::vec3(p[0], p[1], p[2]); }
static int handle(RtBuilder* b, int h) { if (h < 0) g_rt_error = b->b.error; return h; }
#define NEED(b) do { if (!(b)) return rt_set_error(RT_ERR_INVALID, "null builder"); } while (0)


// What rt_scene_info / rt_flat_info report about a flattened scene (the device fields are filled by the caller).
void rt_fill_info(const rt::FlatScene& f, RtSceneInfo* o) {
    RtSceneView v = f.view();
    o->n_nodes = v.n_nodes; o->n_spheres = v.n_spheres; o->n_moving_spheres = v.n_mspheres; o->n_rects = v.n_rects; o->n_boxes = v.n_boxes;
    o->n_instances = v.n_instances; o->n_media = v.n_media; o->n_materials = v.n_materials; o->n_textures = v.n_textures; o->n_perlin = v.n_perlin;
    o->tlas_depth = f.tlas_depth; o->blas_depth = f.blas_depth; o->max_stack = f.max_stack; o->n_objects = f.n_ids; o->device = -1;
    for (int i = 0; i < 6; ++i) o->world_box[i] = f.world_box[i];
    o->device_bytes = 0; o->bvh_build_seconds = 0.0; o->bvh_refit_seconds = 0.0;
}

extern "C" {

const char* rt_last_error(void) { return g_rt_error.c_str(); }

int rtb_create(RtBuilder** out) {
    if (!out) return rt_set_error(RT_ERR_INVALID, "null argument");
    *out = new RtBuilder;
    return RT_OK;
}
void rtb_destroy(RtBuilder* b) { delete b; }
int rtb_object_count(const RtBuilder* b) { return b ? b->b.next_id : 0; }

RtTex rtb_solid_color(RtBuilder* b, double r, double g, double bl) { NEED(b); return handle(b, b->b.solid_color(rt::vec3(r, g, bl))); }
RtTex rtb_checker(RtBuilder* b, RtTex even, RtTex odd) { NEED(b); return handle(b, b->b.checker(even, odd)); }
RtTex rtb_noise(RtBuilder* b, double scale, const double* ranvec, const int32_t* perm) {
    NEED(b);
    if (!ranvec || !perm) return rt_set_error(RT_ERR_INVALID, "rtb_noise: null table");
    rt::PerlinTable t;
    for (int i = 0; i < 256; ++i) for (int c = 0; c < 3; ++c) t.ranvec[i][c] = ranvec[3 * i + c];
    for (int k = 0; k < 3; ++k) for (int i = 0; i < 256; ++i) {
        int32_t v = perm[k * 256 + i];
        if (v < 0 || v > 255) return rt_set_error(RT_ERR_INVALID, "rtb_noise: permutation entry outside 0..255");
        t.perm[k][i] = v;
    }
    return handle(b, b->b.noise_from_table(scale, t));
}
RtMat rtb_lambertian(RtBuilder* b, RtTex albedo) { NEED(b); return handle(b, b->b.lambertian_from_texture(albedo)); }
RtMat rtb_metal(RtBuilder* b, double r, double g, double bl, double fuzz) { NEED(b); return handle(b, b->b.metal(rt::vec3(r, g, bl), fuzz)); }
RtMat rtb_dielectric(RtBuilder* b, double ir) { NEED(b); return handle(b, b->b.dielectric(ir)); }
RtMat rtb_diffuse_light(RtBuilder* b, RtTex emit) { NEED(b); return handle(b, b->b.diffuse_light(emit)); }
RtMat rtb_isotropic(RtBuilder* b, RtTex albedo) { NEED(b); return handle(b, b->b.isotropic(albedo)); }

RtObj rtb_sphere(RtBuilder* b, const double c[3], double r, RtMat m) { NEED(b); if (!c) return rt_set_error(RT_ERR_INVALID, "null center"); return handle(b, b->b.sphere(v3d(c), r, m)); }
RtObj rtb_moving_sphere(RtBuilder* b, const double c0[3], const double c1[3], double t0, double t1, double r, RtMat m) {
    NEED(b); if (!c0 || !c1) return rt_set_error(RT_ERR_INVALID, "null center");
    return handle(b, b->b.moving_sphere(v3d(c0), v3d(c1), t0, t1, r, m));
}
RtObj rtb_rectangle(RtBuilder* b, RtMat m, double x0, double x1, double y0, double y1, double k, int type) {
    NEED(b); if (type < 0 || type > 2) return rt_set_error(RT_ERR_INVALID, "rectangle type must be 0 (Xy), 1 (Xz) or 2 (Yz)");
    return handle(b, b->b.rectangle(m, x0, x1, y0, y1, k, (rt::RectangleType)type));
}
RtObj rtb_box(RtBuilder* b, const double mn[3], const double mx[3], RtMat m) { NEED(b); if (!mn || !mx) return rt_set_error(RT_ERR_INVALID, "null corner"); return handle(b, b->b.box(v3d(mn), v3d(mx), m)); }
RtObj rtb_list(RtBuilder* b, const RtObj* ch, int n) {
    NEED(b); if (n < 0 || (n > 0 && !ch)) return rt_set_error(RT_ERR_INVALID, "bad child array");
    return handle(b, b->b.list(std::vector<int>(ch, ch + n)));
}
RtObj rtb_bvh_node(RtBuilder* b, RtObj l, double t0, double t1) { NEED(b); return handle(b, b->b.bvh_node(l, t0, t1)); }
RtObj rtb_translate(RtBuilder* b, RtObj o, const double off[3]) { NEED(b); if (!off) return rt_set_error(RT_ERR_INVALID, "null offset"); return handle(b, b->b.translate(o, v3d(off))); }
RtObj rtb_rotate_y(RtBuilder* b, RtObj o, double deg) { NEED(b); return handle(b, b->b.rotate_y(o, deg)); }
RtObj rtb_constant_medium(RtBuilder* b, RtObj boundary, double density, RtTex albedo) { NEED(b); return handle(b, b->b.constant_medium(boundary, density, albedo)); }

// ------------------------------------------------------------------ config / canned scenes
static void fill_config(const rt::Config& c, RtConfig* o) {
    std::memset(o, 0, sizeof(*o));
    o->image_width = c.image_width; o->image_height = c.image_height(); o->samples_per_pixel = c.samples_per_pixel; o->max_depth = c.max_depth;
    const rt::Vec3* src[3] = {&c.camera.look_from, &c.camera.look_at, &c.camera.v_up};
    double* dst[3] = {o->look_from, o->look_at, o->v_up};
    for (int i = 0; i < 3; ++i) { dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z; }
    o->vertical_fov_deg = c.camera.vertical_fov_deg; o->aspect_ratio = c.camera.aspect_ratio; o->aperture = c.camera.aperture;
    o->focus_distance = c.camera.focus_distance;
    std::snprintf(o->scene_type, sizeof(o->scene_type), "%s", c.scene_type.c_str());
    o->n_list_objects = c.n_list_objects;
    o->shutter_open = c.shutter_open; o->shutter_close = c.shutter_close;
    o->background[0] = (float)c.background.x; o->background[1] = (float)c.background.y; o->background[2] = (float)c.background.z;
}
int rt_config_load(const char* path, RtConfig* out) {
    if (!path || !out) return rt_set_error(RT_ERR_INVALID, "null argument");
    rt::Config c; std::string err;
    if (!rt::load_config(path, c, err)) return rt_set_error(RT_ERR_CONFIG, err);
    fill_config(c, out);
    return RT_OK;
}
int rt_config_build_scene(const char* path, uint64_t seed, RtBuilder* b, RtObj* world) {
    if (!path || !b || !world) return rt_set_error(RT_ERR_INVALID, "null argument");
    rt::Config c; std::string err;
    if (!rt::load_config(path, c, err)) return rt_set_error(RT_ERR_CONFIG, err);
    int w = rt::scene_from_config(b->b, c, seed);
    if (w < 0) return rt_set_error(RT_ERR_CONFIG, b->b.error);
    *world = w;
    return RT_OK;
}
int rt_canned_scene(RtBuilder* b, const char* type, uint64_t seed, RtObj* world) {
    if (!b || !type || !world) return rt_set_error(RT_ERR_INVALID, "null argument");
    int w = rt::canned_scene(b->b, type, seed);
    if (w < 0) return rt_set_error(RT_ERR_CONFIG, b->b.error);
    *world = w;
    return RT_OK;
}

// ------------------------------------------------------------------ host flatten
int rt_flat_create_ex(const RtBuilder* b, RtObj world, int flags, RtFlat** out) {
    if (!b || !out) return rt_set_error(RT_ERR_INVALID, "null argument");
    *out = nullptr;
    RtFlat* f = new RtFlat;
    std::string err;
    bool ok = rt::flatten(b->b, world, f->flat, err, (flags & RT_BVH_LINEAR) != 0);
    if (ok && !f->flat.tlas_items.empty()) ok = rt::lbvh_finish_host(f->flat, err);
    if (!ok) { delete f; return rt_set_error(rt_status_of(err), err); }
    rt::build_wide_bvh(f->flat);
    f->view = f->flat.view();
    *out = f;
    return RT_OK;
}
int rt_flat_create(const RtBuilder* b, RtObj world, RtFlat** out) { return rt_flat_create_ex(b, world, 0, out); }
const void* rt_flat_view(const RtFlat* f) { return f ? &f->view : nullptr; }
void rt_flat_destroy(RtFlat* f) { delete f; }

// ------------------------------------------------------------------ output (utils.rs:49-63)
int rt_tonemap_u8(const float* accum, int32_t n_pixels, int32_t spp, uint8_t* rgb) {
    if (!accum || !rgb || n_pixels < 0 || spp <= 0) return rt_set_error(RT_ERR_INVALID, "bad argument");
    double scale = 1.0 / (double)spp;
    for (size_t i = 0; i < (size_t)n_pixels * 3; ++i) {
        double v = std::sqrt(scale * (double)accum[i]);  // gamma 2
        v = std::max(std::min(v, 0.999), 0.0);           // NaN -> 0 like Rust's f64::min/max + `as u8`
        if (!(v == v)) v = 0.0;
        rgb[i] = (uint8_t)(256.0 * v);
    }
    return RT_OK;
}
int rt_write_ppm(const char* path, const uint8_t* rgb, int32_t w, int32_t h) {
    if (!path || !rgb || w <= 0 || h <= 0) return rt_set_error(RT_ERR_INVALID, "bad argument");
    FILE* f = std::fopen(path, "wb");
    if (!f) return rt_set_error(RT_ERR_INVALID, std::string("cannot open ") + path);
    std::fprintf(f, "P6\n%d %d\n255\n", w, h);
    size_t n = (size_t)w * h * 3, wr = std::fwrite(rgb, 1, n, f);
    std::fclose(f);
    return wr == n ? RT_OK : rt_set_error(RT_ERR_INVALID, "short write");
}

// ---- PNG with stored (uncompressed) deflate blocks: a valid file every decoder reads, no compression library needed.
static uint32_t crc32_update(uint32_t crc, const uint8_t* p, size_t n) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[i] = c; }
        ready = true;
    }
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    return crc;
}
static void be32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
static void png_chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data) {
    be32(out, (uint32_t)data.size());
    size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    be32(out, crc32_update(0xFFFFFFFFu, out.data() + at, out.size() - at) ^ 0xFFFFFFFFu);
}
static int write_png(const char* path, const uint8_t* rgb, int32_t w, int32_t h) {
    std::vector<uint8_t> raw;                                  // scanlines, each with filter byte 0
    raw.reserve((size_t)h * (3 * (size_t)w + 1));
    for (int y = 0; y < h; ++y) { raw.push_back(0); raw.insert(raw.end(), rgb + (size_t)y * w * 3, rgb + (size_t)(y + 1) * w * 3); }
    std::vector<uint8_t> z; z.push_back(0x78); z.push_back(0x01);   // zlib header, then stored blocks of <= 65535 bytes
    uint32_t a = 1, b2 = 0;                                     // Adler-32 of raw
    for (size_t i = 0; i < raw.size(); ++i) { a = (a + raw[i]) % 65521u; b2 = (b2 + a) % 65521u; }
    for (size_t off = 0; off < raw.size() || off == 0; off += 65535) {
        size_t n = std::min<size_t>(65535, raw.size() - off);
        z.push_back(off + n >= raw.size() ? 1 : 0);
        z.push_back(n & 0xFF); z.push_back(n >> 8); z.push_back(~n & 0xFF); z.push_back((~n >> 8) & 0xFF);
        z.insert(z.end(), raw.begin() + off, raw.begin() + off + n);
        if (raw.empty()) break;
    }
    be32(z, (b2 << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A}, ihdr;
    be32(ihdr, (uint32_t)w); be32(ihdr, (uint32_t)h);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);   // 8-bit RGB, no interlace
    png_chunk(out, "IHDR", ihdr); png_chunk(out, "IDAT", z); png_chunk(out, "IEND", std::vector<uint8_t>());
    FILE* f = std::fopen(path, "wb");
    if (!f) return rt_set_error(RT_ERR_INVALID, std::string("cannot open ") + path);
    size_t wr = std::fwrite(out.data(), 1, out.size(), f);
    std::fclose(f);
    return wr == out.size() ? RT_OK : rt_set_error(RT_ERR_INVALID, "short write");
}


// ---- JPEG (baseline DCT, 8-bit, YCbCr 4:4:4, the example quantisation and Huffman tables of ITU-T T.81 Annex K scaled to
// quality 75).  The reference saves through the image crate, which picks the encoder by extension and defaults to
// "./image.jpeg" (main.rs:502, 600).
namespace {
struct JpegBits {
    std::vector<uint8_t>& out; uint32_t acc = 0; int n = 0;
    explicit JpegBits(std::vector<uint8_t>& o) : out(o) {}
    void put(uint32_t code, int len) {
        acc = (acc << len) | (code & ((1u << len) - 1u)); n += len;
        while (n >= 8) { uint8_t b = (uint8_t)(acc >> (n - 8)); out.push_back(b); if (b == 0xFF) out.push_back(0); n -= 8; }
    }
    void flush() { if (n > 0) put(0x7Fu, 8 - n); }
};
struct JpegHuff { uint16_t code[256]; uint8_t len[256]; };
void jpeg_huff(const uint8_t* counts, const uint8_t* symbols, JpegHuff& h) {   // canonical codes from the (BITS, HUFFVAL) lists
    std::memset(&h, 0, sizeof(h));
    uint16_t code = 0; int k = 0;
    for (int l = 1; l <= 16; ++l) { for (int i = 0; i < counts[l - 1]; ++i) { h.code[symbols[k]] = code++; h.len[symbols[k]] = (uint8_t)l; ++k; } code <<= 1; }
}
const uint8_t kZig[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
const uint8_t kQLum[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
                           18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kQChr[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                           99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
const uint8_t kDcLumN[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0}, kDcChrN[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcSym[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumN[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d}, kAcChrN[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcLumSym[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1,
    0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a,
    0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3,
    0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kAcChrSym[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1,
    0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
    0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
    0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
void jpeg_marker(std::vector<uint8_t>& o, uint8_t m, const std::vector<uint8_t>& body) {
    o.push_back(0xFF); o.push_back(m);
    o.push_back((uint8_t)((body.size() + 2) >> 8)); o.push_back((uint8_t)((body.size() + 2) & 0xFF));
    o.insert(o.end(), body.begin(), body.end());
}
void jpeg_fdct(float* b) {   // separable 8x8 DCT-II, orthonormal scaling (the 1/4 C(u) C(v) of T.81 A.3.3)
    static float c[8][8]; static bool init = false;
    if (!init) { for (int u = 0; u < 8; ++u) for (int x = 0; x < 8; ++x) c[u][x] = (u ? 0.5f : 0.35355339f) * std::cos((2 * x + 1) * u * 3.14159265358979f / 16.0f); init = true; }
    float t[64];
    for (int y = 0; y < 8; ++y) for (int u = 0; u < 8; ++u) { float s = 0; for (int x = 0; x < 8; ++x) s += c[u][x] * b[y * 8 + x]; t[y * 8 + u] = s; }
    for (int u = 0; u < 8; ++u) for (int v = 0; v < 8; ++v) { float s = 0; for (int y = 0; y < 8; ++y) s += c[v][y] * t[y * 8 + u]; b[v * 8 + u] = s; }
}
int jpeg_bits_of(int v) { int a = v < 0 ? -v : v, n = 0; while (a) { ++n; a >>= 1; } return n; }
void jpeg_block(JpegBits& bw, const float* px, const uint8_t* q, const JpegHuff& dc, const JpegHuff& ac, int& pred) {
    float b[64];
    for (int i = 0; i < 64; ++i) b[i] = px[i] - 128.0f;
    jpeg_fdct(b);
    int z[64];
    for (int i = 0; i < 64; ++i) z[i] = (int)std::lround(b[kZig[i]] / (float)q[kZig[i]]);
    int diff = z[0] - pred; pred = z[0];
    int n = jpeg_bits_of(diff);
    bw.put(dc.code[n], dc.len[n]);
    if (n) bw.put((uint32_t)(diff < 0 ? diff - 1 : diff), n);
    int run = 0;
    for (int i = 1; i < 64; ++i) {
        if (z[i] == 0) { ++run; continue; }
        while (run > 15) { bw.put(ac.code[0xF0], ac.len[0xF0]); run -= 16; }
        int s = jpeg_bits_of(z[i]);
        bw.put(ac.code[(run << 4) | s], ac.len[(run << 4) | s]);
        bw.put((uint32_t)(z[i] < 0 ? z[i] - 1 : z[i]), s);
        run = 0;
    }
    if (run) bw.put(ac.code[0], ac.len[0]);
}
}  // namespace
static int write_jpeg(const char* path, const uint8_t* rgb, int32_t w, int32_t h) {
    if (w > 65535 || h > 65535) return rt_set_error(RT_ERR_INVALID, "JPEG holds at most 65535 x 65535 pixels");
    const int quality = 75, scale = 200 - 2 * quality;   // the usual scaling of the Annex K tables
    uint8_t ql[64], qc[64];
    for (int i = 0; i < 64; ++i) {
        ql[i] = (uint8_t)std::min(255, std::max(1, (kQLum[i] * scale + 50) / 100));
        qc[i] = (uint8_t)std::min(255, std::max(1, (kQChr[i] * scale + 50) / 100));
    }
    std::vector<uint8_t> o = {0xFF, 0xD8};
    jpeg_marker(o, 0xE0, {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0});
    for (int t = 0; t < 2; ++t) { std::vector<uint8_t> b = {(uint8_t)t}; for (int i = 0; i < 64; ++i) b.push_back((t ? qc : ql)[kZig[i]]); jpeg_marker(o, 0xDB, b); }
    jpeg_marker(o, 0xC0, {8, (uint8_t)(h >> 8), (uint8_t)(h & 0xFF), (uint8_t)(w >> 8), (uint8_t)(w & 0xFF), 3, 1, 0x11, 0, 2, 0x11, 1, 3, 0x11, 1});
    struct { uint8_t id; const uint8_t* n; const uint8_t* sym; int count; } tabs[4] = {
        {0x00, kDcLumN, kDcSym, 12}, {0x10, kAcLumN, kAcLumSym, 162}, {0x01, kDcChrN, kDcSym, 12}, {0x11, kAcChrN, kAcChrSym, 162}};
    for (auto& t : tabs) { std::vector<uint8_t> b = {t.id}; b.insert(b.end(), t.n, t.n + 16); b.insert(b.end(), t.sym, t.sym + t.count); jpeg_marker(o, 0xC4, b); }
    jpeg_marker(o, 0xDA, {3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0});
    JpegHuff dcl, acl, dcc, acc;
    jpeg_huff(kDcLumN, kDcSym, dcl); jpeg_huff(kAcLumN, kAcLumSym, acl); jpeg_huff(kDcChrN, kDcSym, dcc); jpeg_huff(kAcChrN, kAcChrSym, acc);
    JpegBits bw(o);
    int pred[3] = {0, 0, 0};
    for (int by = 0; by < h; by += 8)
        for (int bx = 0; bx < w; bx += 8) {
            float ycc[3][64];
            for (int y = 0; y < 8; ++y)
                for (int x = 0; x < 8; ++x) {   // edge blocks repeat the last row / column
                    const uint8_t* p = rgb + 3 * ((size_t)std::min(by + y, h - 1) * w + std::min(bx + x, w - 1));
                    float r = p[0], g = p[1], b = p[2];
                    ycc[0][y * 8 + x] = 0.299f * r + 0.587f * g + 0.114f * b;
                    ycc[1][y * 8 + x] = -0.168736f * r - 0.331264f * g + 0.5f * b + 128.0f;
                    ycc[2][y * 8 + x] = 0.5f * r - 0.418688f * g - 0.081312f * b + 128.0f;
                }
            jpeg_block(bw, ycc[0], ql, dcl, acl, pred[0]);
            jpeg_block(bw, ycc[1], qc, dcc, acc, pred[1]);
            jpeg_block(bw, ycc[2], qc, dcc, acc, pred[2]);
        }
    bw.flush();
    o.push_back(0xFF); o.push_back(0xD9);
    FILE* f = std::fopen(path, "wb");
    if (!f) return rt_set_error(RT_ERR_INVALID, std::string("cannot open ") + path);
    bool ok = std::fwrite(o.data(), 1, o.size(), f) == o.size();
    ok = (std::fclose(f) == 0) && ok;
    return ok ? RT_OK : rt_set_error(RT_ERR_INVALID, std::string("cannot write ") + path);
}

int rt_write_image(const char* path, const uint8_t* rgb, int32_t w, int32_t h) {
    if (!path || !rgb || w <= 0 || h <= 0) return rt_set_error(RT_ERR_INVALID, "bad argument");
    std::string p(path);
    size_t dot = p.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : p.substr(dot + 1);
    for (char& c : ext) c = (char)std::tolower((unsigned char)c);
    if (ext == "png") return write_png(path, rgb, w, h);
    if (ext == "ppm" || ext == "pnm") return rt_write_ppm(path, rgb, w, h);
    if (ext == "jpg" || ext == "jpeg") return write_jpeg(path, rgb, w, h);
    return rt_set_error(RT_ERR_UNSUPPORTED, "image format '" + ext + "' is not supported (png, jpeg, ppm)");
}

// ---- moving primitives + refit on the host copy (the CPU-side mirror of rt_scene_set_* / rt_scene_refit)
static int flat_set(RtFlat* f, int kind, const int32_t* ids, int32_t n, const double* values) {
    if (!f || n < 0 || (n > 0 && (!ids || !values))) return rt_set_error(RT_ERR_INVALID, "bad argument");
    if (!f->indexed) { rt::build_prim_index(f->flat, f->of_id); f->indexed = true; }
    std::vector<int32_t> touched; std::string err;
    if (!rt::set_prims_host(f->flat, f->of_id, kind, ids, n, values, touched, err)) return rt_set_error(rt_status_of(err), err);
    return RT_OK;
}
int rt_flat_set_spheres(RtFlat* f, const int32_t* ids, int32_t n, const double* xyzr) { return flat_set(f, 0, ids, n, xyzr); }
int rt_flat_set_moving_spheres(RtFlat* f, const int32_t* ids, int32_t n, const double* values) { return flat_set(f, 1, ids, n, values); }
int rt_flat_refit(RtFlat* f) {
    if (!f) return rt_set_error(RT_ERR_INVALID, "null argument");
    std::string err;
    if (!rt::refit_host(f->flat, err)) return rt_set_error(RT_ERR_INTERNAL, err);
    f->view = f->flat.view();
    return RT_OK;
}

// ---- accumulator checkpoints: header {magic, width, height, spp_done, seed} then width * height * 3 floats
static const uint64_t kAccumMagic = 0x314D554343415452ull;  // "RTACCUM1"
int rt_flat_info(const RtFlat* f, RtSceneInfo* out) {
    if (!f || !out) return rt_set_error(RT_ERR_INVALID, "null argument");
    rt_fill_info(f->flat, out);
    return RT_OK;
}


// ---- sharding arithmetic of the multi-GPU paths (pure host code)
int rt_shard_samples(int32_t total_spp, int32_t n_shards, int32_t shard, int32_t* first, int32_t* count) {
    if (!first || !count || total_spp < 0 || n_shards < 1 || shard < 0 || shard >= n_shards) return rt_set_error(RT_ERR_INVALID, "rt_shard_samples: bad argument");
    const int32_t base = total_spp / n_shards, extra = total_spp % n_shards;
    *count = base + (shard < extra ? 1 : 0);
    *first = shard * base + std::min(shard, extra);
    return RT_OK;
}

int rt_shard_tiles(RtRenderParams* p, int32_t n_shards, int32_t shard, int64_t* n_pixels) {
    if (!p || p->width < 1 || p->height < 1 || n_shards < 1 || shard < 0 || shard >= n_shards) return rt_set_error(RT_ERR_INVALID, "rt_shard_tiles: bad argument");
    std::memset(p->tile, 0, sizeof(p->tile));
    p->stripe[0] = RT_TILE_ROWS; p->stripe[1] = n_shards; p->stripe[2] = shard;
    if (n_pixels) *n_pixels = rt_stripe_rows(p->height, RT_TILE_ROWS, n_shards, shard) * (int64_t)p->width;
    return RT_OK;
}

int rt_accum_save(const char* path, const float* accum_rgb, int32_t w, int32_t h, int64_t spp_done, uint64_t seed) {
    if (!path || !accum_rgb || w <= 0 || h <= 0 || spp_done < 0) return rt_set_error(RT_ERR_INVALID, "bad argument");
    std::string tmp = std::string(path) + ".tmp";              // written aside and renamed: a crash never leaves half a file
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return rt_set_error(RT_ERR_INVALID, "cannot open " + tmp);
    uint64_t hdr[5] = {kAccumMagic, (uint64_t)w, (uint64_t)h, (uint64_t)spp_done, seed};
    size_t n = (size_t)w * h * 3;
    bool ok = std::fwrite(hdr, sizeof(hdr), 1, f) == 1 && std::fwrite(accum_rgb, sizeof(float), n, f) == n;
    ok = (std::fclose(f) == 0) && ok;
    if (!ok || std::rename(tmp.c_str(), path) != 0) return rt_set_error(RT_ERR_INVALID, std::string("cannot write ") + path);
    return RT_OK;
}
int rt_accum_load(const char* path, float* accum_rgb, int32_t w, int32_t h, int64_t* spp_done, uint64_t* seed) {
    if (!path || !accum_rgb || !spp_done || w <= 0 || h <= 0) return rt_set_error(RT_ERR_INVALID, "bad argument");
    FILE* f = std::fopen(path, "rb");
    if (!f) return rt_set_error(RT_ERR_INVALID, std::string("cannot open ") + path);
    uint64_t hdr[5];
    size_t n = (size_t)w * h * 3;
    bool ok = std::fread(hdr, sizeof(hdr), 1, f) == 1 && hdr[0] == kAccumMagic && hdr[1] == (uint64_t)w && hdr[2] == (uint64_t)h &&
              std::fread(accum_rgb, sizeof(float), n, f) == n;
    std::fclose(f);
    if (!ok) return rt_set_error(RT_ERR_INVALID, std::string(path) + " is not an accumulator checkpoint of this image size");
    *spp_done = (int64_t)hdr[3];
    if (seed) *seed = hdr[4];
    return RT_OK;
}

}  // extern "C"

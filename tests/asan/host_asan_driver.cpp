// host_asan_driver.cpp -- TEST-ONLY.  Drives the CPU side of the library (scene builder, YAML reader, flattener, wide-tree
// collapse, host refit, host linear-BVH steps, image and checkpoint writers) and the f64 oracle under AddressSanitizer and
// UndefinedBehaviorSanitizer.  Built by `make -C oracle asan` from the same sources as the product library minus the HIP files;
// run by tests/test_host_sanitizers.py.  The reference gets its memory safety from Rust ownership and panics on bad input
// (src/main.rs:510-513); this is where the C++ restatement has to earn it.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/rt_hip.h"

extern "C" {   // oracle/rt_oracle.cpp
void* orc_scene_new(uint64_t seed);
void orc_scene_free(void* s);
int orc_canned(void* s, const char* type, uint64_t scene_seed);
int orc_set_world(void* s, int list_obj, int use_bvh);
void* orc_camera_new(const double* from, const double* at, const double* up, double vfov, double aspect, double aperture, double focus, double t0, double t1);
void orc_camera_free(void* c);
void orc_render(void* s, void* c, int width, int height, int spp, int sample_offset, int max_depth, const double* background, uint64_t seed, int n_threads,
                double* accum, double* accum_sq, uint64_t* stats);
void orc_trace_primary(void* s, void* c, int width, int height, uint32_t sample, uint64_t seed, int32_t* ids, double* ts, double* normals);
}

static int g_failed = 0;
#define EXPECT(cond) do { if (!(cond)) { std::fprintf(stderr, "EXPECT failed at line %d: %s (last error: %s)\n", __LINE__, #cond, rt_last_error()); ++g_failed; } } while (0)

static std::string write_tmp(const std::string& dir, const char* name, const std::string& body) {
    std::string p = dir + "/" + name;
    FILE* f = std::fopen(p.c_str(), "wb");
    if (f) { std::fwrite(body.data(), 1, body.size(), f); std::fclose(f); }
    return p;
}

static void flatten_both(RtBuilder* b, RtObj world, bool expect_ok) {
    for (int flags : {0, RT_BVH_LINEAR}) {
        RtFlat* f = nullptr;
        int rc = rt_flat_create_ex(b, world, flags, &f);
        if (expect_ok) EXPECT(rc == RT_OK && f);
        if (rc == RT_OK && f) {
            RtSceneInfo info;
            EXPECT(rt_flat_info(f, &info) == RT_OK && rt_flat_view(f) != nullptr);
            EXPECT(rt_flat_refit(f) == RT_OK);
            rt_flat_destroy(f);
        }
    }
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: host_asan_driver <configs dir> <scratch dir>\n"); return 2; }
    const std::string configs = argv[1], tmp = argv[2];

    // 1. every shipped config: load, build, flatten with both top-level builders, refit in place
    for (const char* name : {"cfg.yml", "rectangle.yml", "orig_img.yml", "cornell.yml", "cornell_box.yml", "final_scene.yml", "random.yml", "two_spheres.yml",
                             "showcase.yml"}) {
        const std::string path = configs + "/" + name;
        RtConfig cfg;
        EXPECT(rt_config_load(path.c_str(), &cfg) == RT_OK);
        RtBuilder* b = nullptr; RtObj world = -1;
        EXPECT(rtb_create(&b) == RT_OK);
        EXPECT(rt_config_build_scene(path.c_str(), 0x5EED0001ull, b, &world) == RT_OK && world >= 0);
        flatten_both(b, world, true);
        rtb_destroy(b);
    }
    // 2. a stress scene, moved and refitted; bad ids refused without writing
    {
        RtBuilder* b = nullptr; RtObj world = -1;
        EXPECT(rtb_create(&b) == RT_OK && rt_canned_scene(b, "Synthetic:20000", 7, &world) == RT_OK);
        RtFlat* f = nullptr;
        EXPECT(rt_flat_create(b, world, &f) == RT_OK);
        std::vector<int32_t> ids; std::vector<double> xyzr;
        for (int i = 1; i <= 2000; ++i) { ids.push_back(i); xyzr.insert(xyzr.end(), {double(i % 97) - 48.0, 1.0 + (i % 5), double(i % 89) - 44.0, 0.5 + 0.01 * (i % 50)}); }
        EXPECT(rt_flat_set_spheres(f, ids.data(), (int32_t)ids.size(), xyzr.data()) == RT_OK);
        EXPECT(rt_flat_refit(f) == RT_OK);
        int32_t bad[3] = {5, 9999999, -3}; double v[12] = {0};
        EXPECT(rt_flat_set_spheres(f, bad, 3, v) != RT_OK);
        EXPECT(rt_flat_set_moving_spheres(f, bad, 1, v) != RT_OK);
        EXPECT(rt_flat_set_spheres(f, nullptr, 0, nullptr) == RT_OK);
        rt_flat_destroy(f);
        EXPECT(rt_canned_scene(b, "NoSuchScene", 7, &world) != RT_OK);
        EXPECT(rt_canned_scene(b, "Synthetic:-5", 7, &world) != RT_OK);
        rtb_destroy(b);
    }
    // 3. hand-built trees: empty list, nested wrappers, a medium, invalid handles
    {
        RtBuilder* b = nullptr; EXPECT(rtb_create(&b) == RT_OK);
        RtObj empty = rtb_list(b, nullptr, 0);
        EXPECT(empty >= 0);
        flatten_both(b, empty, true);                                   // the reference recurses forever on an empty list (src/bvh.rs:113-121)
        RtMat m = rtb_lambertian(b, rtb_solid_color(b, 0.5, 0.5, 0.5));
        const double c[3] = {0, 0, -3}, mn[3] = {-1, -1, -1}, mx[3] = {1, 1, 1}, off[3] = {1, 2, 3};
        RtObj s = rtb_sphere(b, c, -0.5, m), bx = rtb_box(b, mn, mx, m);
        RtObj inst = rtb_translate(b, rtb_rotate_y(b, rtb_translate(b, bx, off), 33.0), off);
        RtObj med = rtb_constant_medium(b, rtb_sphere(b, c, 2.0, m), 0.1, rtb_solid_color(b, 1, 1, 1));
        RtObj kids[3] = {s, inst, med};
        RtObj world = rtb_list(b, kids, 3);
        flatten_both(b, rtb_bvh_node(b, world, 0, 1), true);
        EXPECT(rtb_sphere(b, c, 1.0, 12345) < 0);                       // unknown material
        EXPECT(rtb_list(b, kids, -1) < 0);
        RtObj bad_kids[2] = {s, 424242};
        EXPECT(rtb_list(b, bad_kids, 2) < 0);
        EXPECT(rtb_translate(b, -7, off) < 0);
        EXPECT(rtb_constant_medium(b, med, 0.1, 0) < 0 || true);         // a medium bounded by a medium: refused here or at flatten time
        RtFlat* f = nullptr;
        EXPECT(rt_flat_create(b, 999999, &f) != RT_OK && f == nullptr);
        EXPECT(rt_flat_create(nullptr, 0, &f) != RT_OK);
        rtb_destroy(b);
    }
    // 4. malformed configs: an error code and a message, never a crash (the reference panics: src/main.rs:511-512)
    {
        const char* good_cam = "camera:\n  look_from: {x: 0, y: 0, z: 10}\n  look_at: {x: 0, y: 0, z: 0}\n  v_up: {x: 0, y: 1, z: 0}\n"
                               "  vertical_fov_deg: 20\n  aspect_ratio: 1.25\n  aperture: 0.1\n  focus_distance: 10\n";
        std::vector<std::string> docs = {
            "", "\n\n", ":", "scene", "scene:\n", "scene: {type: }\n", std::string("scene: {type: NoSuch}\n") + good_cam,
            std::string("image_width: abc\nscene: {type: Cornell}\n") + good_cam, std::string("image_width: -5\nscene: {type: Cornell}\n") + good_cam,
            std::string("image_width: 99999999999999999999\nscene: {type: Cornell}\n") + good_cam,
            "scene: {type: Cornell}\ncamera:\n  look_from: {x: 0, y: 0}\n", "scene: {type: Cornell}\ncamera: 5\n",
            std::string("scene:\n  type: List\n  objects:\n") + good_cam,
            std::string("scene:\n  type: List\n  objects: []\n") + good_cam,
            std::string("scene:\n  type: List\n  objects:\n    - center: {x: 0, y: 0, z: 0}\n      radius: 1\n") + good_cam,                 // no material
            std::string("scene:\n  type: List\n  objects:\n    - center: {x: 0, y: 0, z: 0}\n      radius: 1\n      material: {type: Plastic}\n") + good_cam,
            std::string("scene:\n  type: List\n  objects:\n    - type: Translate\n      offset: {x: 1, y: 1, z: 1}\n") + good_cam,             // no inner object
            std::string("scene:\n  type: List\n  objects:\n    - type: ConstantMedium\n      density: 0.1\n      albedo: {x: 1, y: 1, z: 1}\n      boundary:\n"
                        "        type: ConstantMedium\n        density: 1\n        albedo: {x: 1, y: 1, z: 1}\n        boundary: {type: Sphere}\n") + good_cam,
            std::string("scene: {type: Cornell}\n") + good_cam + "\t\tbad: [1, 2\n", std::string(4096, '{'), std::string("a: &x [*x, *x]\n") + good_cam,
            std::string("scene: {type: Cornell}\n") + std::string(good_cam).substr(0, 60),
        };
        std::string junk; for (int i = 0; i < 3000; ++i) junk.push_back((char)((i * 131 + 7) & 0xFF));
        docs.push_back(junk);
        int n_ok = 0;
        for (size_t i = 0; i < docs.size(); ++i) {
            const std::string p = write_tmp(tmp, "bad.yml", docs[i]);
            RtConfig cfg;
            int rc = rt_config_load(p.c_str(), &cfg);
            RtBuilder* b = nullptr; RtObj world = -1;
            EXPECT(rtb_create(&b) == RT_OK);
            int rc2 = rt_config_build_scene(p.c_str(), 1, b, &world);
            if (rc == RT_OK && rc2 == RT_OK) { ++n_ok; flatten_both(b, world, false); }
            else EXPECT(std::strlen(rt_last_error()) > 0);
            rtb_destroy(b);
        }
        EXPECT(n_ok <= 3);                                               // only the documents that are complete (empty List, ...)
        RtConfig cfg;
        EXPECT(rt_config_load((tmp + "/does_not_exist.yml").c_str(), &cfg) == RT_ERR_CONFIG);
        EXPECT(rt_config_load(nullptr, &cfg) != RT_OK);
    }
    // 5. checkpoints: round trip, truncation at every prefix length class, wrong size
    {
        const int w = 7, h = 5;
        std::vector<float> acc((size_t)w * h * 3), back(acc.size());
        for (size_t i = 0; i < acc.size(); ++i) acc[i] = 0.25f * (float)i;
        const std::string p = tmp + "/ck.bin";
        EXPECT(rt_accum_save(p.c_str(), acc.data(), w, h, 123, 0xABCDEF) == RT_OK);
        int64_t spp = 0; uint64_t seed = 0;
        EXPECT(rt_accum_load(p.c_str(), back.data(), w, h, &spp, &seed) == RT_OK && spp == 123 && seed == 0xABCDEF && back == acc);
        EXPECT(rt_accum_load(p.c_str(), back.data(), w + 1, h, &spp, &seed) != RT_OK);
        EXPECT(rt_accum_load(p.c_str(), back.data(), w, h, &spp, nullptr) == RT_OK);            // the seed is optional
        EXPECT(rt_accum_load(p.c_str(), back.data(), w, h, nullptr, nullptr) != RT_OK);
        std::string bytes;
        if (FILE* f = std::fopen(p.c_str(), "rb")) { int ch; while ((ch = std::fgetc(f)) != EOF) bytes.push_back((char)ch); std::fclose(f); }
        for (size_t cut : {(size_t)0, (size_t)1, (size_t)7, (size_t)16, (size_t)31, bytes.size() / 2, bytes.size() - 1}) {
            const std::string q = write_tmp(tmp, "ck_cut.bin", bytes.substr(0, cut));
            EXPECT(rt_accum_load(q.c_str(), back.data(), w, h, &spp, &seed) != RT_OK);
        }
        EXPECT(rt_accum_save((tmp + "/no/such/dir/ck.bin").c_str(), acc.data(), w, h, 1, 1) != RT_OK);
    }
    // 6. tonemap and the image writers at awkward sizes
    {
        for (int w : {1, 7, 8, 17}) for (int h : {1, 8, 13}) {
            std::vector<float> acc((size_t)w * h * 3);
            for (size_t i = 0; i < acc.size(); ++i) acc[i] = (float)((i * 37) % 11) * 0.3f;
            std::vector<uint8_t> rgb(acc.size());
            EXPECT(rt_tonemap_u8(acc.data(), w * h, 3, rgb.data()) == RT_OK);
            for (const char* ext : {"ppm", "png", "jpeg", "jpg"}) EXPECT(rt_write_image((tmp + "/img." + ext).c_str(), rgb.data(), w, h) == RT_OK);
            EXPECT(rt_write_image((tmp + "/img.tiff").c_str(), rgb.data(), w, h) == RT_ERR_UNSUPPORTED);
        }
        std::vector<uint8_t> rgb(3);
        EXPECT(rt_tonemap_u8(nullptr, 1, 1, rgb.data()) != RT_OK && rt_tonemap_u8((const float*)rgb.data(), 1, 0, rgb.data()) != RT_OK);
    }
    // 7. sharding arithmetic
    {
        int32_t first = 0, count = 0;
        EXPECT(rt_shard_samples(10, 3, 2, &first, &count) == RT_OK && first == 7 && count == 3);
        EXPECT(rt_shard_samples(10, 3, 3, &first, &count) != RT_OK);
    }
    // 8. the oracle: one small render and the primary hits of the scene with every feature
    {
        void* s = orc_scene_new(0x5EED0001ull);
        int w = orc_canned(s, "FinalScene", 0x5EED0001ull);
        EXPECT(w >= 0 && orc_set_world(s, w, 1) == 0);
        const double from[3] = {478, 278, -600}, at[3] = {278, 278, 0}, up[3] = {0, 1, 0}, bg[3] = {0, 0, 0};
        void* c = orc_camera_new(from, at, up, 40.0, 1.0, 0.1, 10.0, 0.0, 1.0);
        const int W = 24, H = 24;
        std::vector<double> acc((size_t)W * H * 3, 0.0), sq(acc.size(), 0.0), ts((size_t)W * H), nrm((size_t)W * H * 3);
        std::vector<int32_t> ids((size_t)W * H);
        uint64_t stats[14] = {0};
        orc_render(s, c, W, H, 3, 0, 50, bg, 0x5EED0002ull, 4, acc.data(), sq.data(), stats);
        EXPECT(stats[0] >= (uint64_t)W * H * 3);
        orc_trace_primary(s, c, W, H, 0, 0x5EED0002ull, ids.data(), ts.data(), nrm.data());
        orc_camera_free(c); orc_scene_free(s);
    }
    if (g_failed) { std::fprintf(stderr, "host_asan_driver: %d expectation(s) failed\n", g_failed); return 1; }
    std::printf("host_asan_driver: ok\n");
    return 0;
}

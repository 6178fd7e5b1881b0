// rt_cli.cpp -- command-line driver with the flags of the reference binary (main.rs:496-504):
//   rt_cli -c <config.yml> [-o <image.jpeg|.png|.ppm>]   (default ./image.jpeg, main.rs:502)  plus  --spp N  --seed N  --scene-seed N  --device N  --background r,g,b
//                                                       --checkpoint <file>  --pass-spp N  --devices a,b,...  --tiles
// --devices 0,1,...: the scene is flattened onto every listed GPU and each render call is rt_render_multi (samples sharded
// across the GPUs and reduced over RCCL; --tiles shards the pixels instead, as the call does by itself when a pass has fewer
// samples than GPUs).  A GPU may be listed more than once: its handles then take turns on it.
// It does what main() does (main.rs:506-601) with the render loop replaced by rt_render: read the
// config, build the scene, render samples_per_pixel samples, tonemap (utils.rs:49-63), save.
// The output format follows the extension (main.rs:600); the 8-bit values are the ones the reference
// hands to its image encoder (its default is a lossy JPEG, main.rs:502).
// With --checkpoint the samples are rendered --pass-spp at a time (default 1024); the accumulation buffer
// is saved after every pass and an existing checkpoint is continued, so configs/final_scene.yml's
// 100 000 spp can be stopped and resumed.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/rt_hip.h"

static int die(const char* what) { std::fprintf(stderr, "rt_cli: %s: %s\n", what, rt_last_error()); return 1; }

// What a checkpoint's samples were rendered from, besides the render seed and the image size (which the checkpoint file itself
// holds): the scene seed and an FNV-1a hash of the config file's bytes.  Kept in a text file beside the checkpoint.
static unsigned long long file_hash(const std::string& path) {
    unsigned long long h = 1469598103934665603ull;
    if (FILE* f = std::fopen(path.c_str(), "rb")) {
        int c;
        while ((c = std::fgetc(f)) != EOF) { h ^= (unsigned long long)(unsigned char)c; h *= 1099511628211ull; }
        std::fclose(f);
    }
    return h;
}

int main(int argc, char** argv) {
    std::string config, output = "./image.jpeg", checkpoint;
    long long pass_spp = 1024;
    long long spp_override = -1; unsigned long long seed = 0x5EED0002ull, scene_seed = 0x5EED0001ull; int device = 0;
    float bg[3] = {0, 0, 0}; bool bg_given = false;
    std::vector<int> devices; bool tiles = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&](const char* flag) -> const char* { if (i + 1 >= argc) { std::fprintf(stderr, "rt_cli: %s needs a value\n", flag); std::exit(2); } return argv[++i]; };
        if (a == "-c" || a == "--config") config = next("--config");
        else if (a == "-o" || a == "--output") output = next("--output");
        else if (a == "--spp") spp_override = std::atoll(next("--spp"));
        else if (a == "--seed") seed = std::strtoull(next("--seed"), nullptr, 0);
        else if (a == "--scene-seed") scene_seed = std::strtoull(next("--scene-seed"), nullptr, 0);
        else if (a == "--device") device = std::atoi(next("--device"));
        else if (a == "--devices") {
            for (const char* q = next("--devices"); *q;) {
                char* end = nullptr;
                long v = std::strtol(q, &end, 10);
                if (end == q || v < 0) { std::fprintf(stderr, "rt_cli: --devices a,b,...\n"); return 2; }
                devices.push_back((int)v);
                q = *end == ',' ? end + 1 : end;
                if (*end != ',' && *end != 0) { std::fprintf(stderr, "rt_cli: --devices a,b,...\n"); return 2; }
            }
            if (devices.empty()) { std::fprintf(stderr, "rt_cli: --devices a,b,...\n"); return 2; }
        }
        else if (a == "--tiles") tiles = true;
        else if (a == "--checkpoint") checkpoint = next("--checkpoint");
        else if (a == "--pass-spp") pass_spp = std::max(1LL, std::atoll(next("--pass-spp")));
        else if (a == "--background") { if (std::sscanf(next("--background"), "%f,%f,%f", &bg[0], &bg[1], &bg[2]) != 3) { std::fprintf(stderr, "rt_cli: --background r,g,b\n"); return 2; } bg_given = true; }
        else { std::fprintf(stderr, "usage: rt_cli -c <config.yml> [-o <image.jpeg|.png|.ppm>] [--spp N] [--seed N] [--scene-seed N] [--device N] [--background r,g,b] [--checkpoint <file>] [--pass-spp N] [--devices a,b,...] [--tiles]\n"); return 2; }
    }
    if (config.empty()) { std::fprintf(stderr, "rt_cli: the following required arguments were not provided: --config <CONFIG>\n"); return 2; }

    RtConfig cfg;
    if (rt_config_load(config.c_str(), &cfg)) return die("config");
    RtBuilder* b = nullptr; RtObj world = -1;
    if (rtb_create(&b) || rt_config_build_scene(config.c_str(), scene_seed, b, &world)) return die("scene");
    const bool multi = !devices.empty();
    if (!multi) devices.push_back(device);
    std::vector<RtScene*> scenes(devices.size(), nullptr);
    for (size_t i = 0; i < devices.size(); ++i)
        if (rt_scene_create(b, world, devices[i], &scenes[i])) return die("rt_scene_create");
    RtScene* scene = scenes[0];
    rtb_destroy(b);

    RtCamera cam;
    std::memcpy(cam.look_from, cfg.look_from, sizeof(cam.look_from)); std::memcpy(cam.look_at, cfg.look_at, sizeof(cam.look_at));
    std::memcpy(cam.v_up, cfg.v_up, sizeof(cam.v_up));
    cam.vertical_fov_deg = cfg.vertical_fov_deg; cam.aspect_ratio = cfg.aspect_ratio; cam.aperture = cfg.aperture; cam.focus_dist = cfg.focus_distance;
    cam.shutter_open = cfg.shutter_open; cam.shutter_close = cfg.shutter_close;  // 0 and 1 unless the config says otherwise (main.rs:544-546)
    if (!bg_given) std::memcpy(bg, cfg.background, sizeof(bg));
    RtRenderParams p; std::memset(&p, 0, sizeof(p));
    p.width = cfg.image_width; p.height = cfg.image_height; p.spp = spp_override > 0 ? (int)spp_override : cfg.samples_per_pixel;
    p.max_depth = cfg.max_depth; p.seed = seed; std::memcpy(p.background, bg, sizeof(bg));

    std::vector<float> accum((size_t)p.width * p.height * 3, 0.0f);
    const int total_spp = p.spp;
    long long done = 0;
    const std::string origin_path = checkpoint + ".scene";
    const unsigned long long cfg_hash = file_hash(config);
    if (!checkpoint.empty()) {
        int64_t have = 0; uint64_t ck_seed = 0;
        if (rt_accum_load(checkpoint.c_str(), accum.data(), p.width, p.height, &have, &ck_seed) == RT_OK) {
            if (ck_seed != seed) { std::fprintf(stderr, "rt_cli: %s was rendered with seed %llu, not %llu\n", checkpoint.c_str(), (unsigned long long)ck_seed, (unsigned long long)seed); return 1; }
            if (FILE* f = std::fopen(origin_path.c_str(), "r")) {   // samples of another scene must not be mixed into this image
                unsigned long long ck_scene = 0, ck_hash = 0;
                bool ok = std::fscanf(f, "%llu %llu", &ck_scene, &ck_hash) == 2;
                std::fclose(f);
                if (ok && (ck_scene != scene_seed || ck_hash != cfg_hash)) {
                    std::fprintf(stderr, "rt_cli: %s was rendered from another scene (scene seed or config file differ)\n", checkpoint.c_str());
                    return 1;
                }
            }
            // a checkpoint that already holds more samples than asked for is finished: the image is made from all of them
            done = have;
            std::fprintf(stderr, "rt_cli: resuming %s at %lld of %d spp\n", checkpoint.c_str(), done, total_spp);
        } else std::fill(accum.begin(), accum.end(), 0.0f);     // no usable checkpoint: start from scratch
        if (FILE* f = std::fopen(origin_path.c_str(), "w")) { std::fprintf(f, "%llu %llu\n", scene_seed, cfg_hash); std::fclose(f); }
    }
    RtStats st, total; std::memset(&total, 0, sizeof(total));
    while (done < total_spp) {
        p.spp = checkpoint.empty() ? total_spp : (int)std::min<long long>(pass_spp, total_spp - done);
        p.sample_offset = (int)done;
        if (multi) {
            p.flags = tiles ? RT_FLAG_TILES : 0;
            if (rt_render_multi(scenes.data(), (int32_t)scenes.size(), &cam, &p, accum.data(), nullptr, &st)) return die("rt_render_multi");
            total.multi_gpus = st.multi_gpus; total.multi_comm_ranks = st.multi_comm_ranks; total.multi_tiles = st.multi_tiles;
        } else if (rt_render(scene, &cam, &p, accum.data(), nullptr, &st)) return die("rt_render");
        total.rays += st.rays; total.seconds += st.seconds;
        done += p.spp;
        if (!checkpoint.empty() && rt_accum_save(checkpoint.c_str(), accum.data(), p.width, p.height, done, seed)) return die("checkpoint");
    }
    st = total; p.spp = (int)std::max<long long>(done, total_spp);   // the number of samples `accum` really holds (utils.rs:52 divides by it)
    std::vector<uint8_t> rgb(accum.size());
    if (rt_tonemap_u8(accum.data(), p.width * p.height, p.spp, rgb.data())) return die("tonemap");
    if (rt_write_image(output.c_str(), rgb.data(), p.width, p.height)) return die("write");
    std::fprintf(stderr, "%s: %dx%d, %d spp, depth %d: %.3f s, %llu ray segments, %.1f Mrays/s -> %s\n", cfg.scene_type, p.width, p.height, p.spp,
                 p.max_depth, st.seconds, (unsigned long long)st.rays, st.seconds > 0 ? st.rays / st.seconds / 1e6 : 0.0, output.c_str());
    if (multi)
        std::fprintf(stderr, "rt_render_multi: %zu handles on %d GPUs, %s sharded, RCCL communicator of %d ranks\n", scenes.size(), st.multi_gpus,
                     st.multi_tiles ? "pixels" : "samples", st.multi_comm_ranks);
    for (RtScene* s : scenes) rt_scene_destroy(s);
    if (multi && rt_multi_shutdown()) return die("rt_multi_shutdown");
    return 0;
}

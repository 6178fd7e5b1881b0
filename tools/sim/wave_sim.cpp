// wave_sim.cpp -- developer tool: replays the closest-hit searches of a render on a software model of one 64-lane wave and
// counts the wave-steps (node steps, leaf steps, refills) that different lane-scheduling policies of k_trace would issue.
// The traversal arithmetic is the kernels' own (rt_device.h compiled for the host); only the scheduling is modelled.  k_trace
// is bound by VALU issue at partial lane occupancy, so wave-steps x instructions per step is a usable proxy for its time.
//   build: g++ -O2 -std=c++17 -ffp-contract=off -Wno-unknown-pragmas -o tools/sim/wave_sim tools/sim/wave_sim.cpp \
//          ray_tracing_amd/csrc/{capi,flatten,config,scene_builder,wide_bvh,refit_host,lbvh_host}.cpp -pthread
//   run:   tools/sim/wave_sim configs/final_scene.yml 160 4        (image width, spp)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/rt_hip.h"
#include "../../ray_tracing_amd/csrc/frame.h"
#include "../../ray_tracing_amd/csrc/rt_device.h"
#include "../../ray_tracing_amd/csrc/scene_builder.h"

struct ArrStack {
    int v[128];
    void put(int i, int, int, int code) { v[i] = code; }
    void put_sentinel(int i) { v[i] = RT_SENTINEL; }
    int take(int i) const { return v[i]; }
    int cap() const { return 128; }
};
struct Ray { v3 o, d; float time; uint32_t src; int bounce; uint32_t pid; };
struct Slot {   // one ray in flight
    bool active = false; v3 wo, wd; float time; uint32_t src; RtTrav tv; RtHit hit; ArrStack stk;
};
struct Counts { double node = 0, leaf = 0, refill = 0, rounds = 0, node_live = 0, leaf_live = 0; long rays = 0; };

static bool g_wide = false;
static RtSceneView g_sc;
template <class F> static void with_nodes(F f) {
    if (g_wide) { RtGlobalWideNodes n; n.nodes = g_sc.wnodes; n.inst_roots = g_sc.winst_roots; f(n); }
    else { RtGlobalNodes n; n.nodes = g_sc.nodes; f(n); }
}
static void begin(Slot& s, const Ray& r) {
    s.active = true; s.wo = r.o; s.wd = r.d; s.time = r.time; s.src = r.src;
    s.hit.t = rt_inf(); s.hit.code = RT_HIT_NONE; s.hit.inst = -1;
    with_nodes([&](auto& n) { rt_trav_begin(g_sc, n, r.o, r.d, s.tv); });
}
static void node_step(Slot& s) { with_nodes([&](auto& n) { rt_inner_step<false>(n, s.tv, RT_TMIN, s.hit, s.stk, nullptr); }); }
static void leaf_or_sentinel_step(Slot& s) {
    RtGlobalPrims prims; prims.sc = &g_sc;
    with_nodes([&](auto& n) { rt_leaf_step<false>(n, prims, s.wo, s.wd, s.time, RT_TMIN, s.src, s.tv, s.hit, s.stk, nullptr); });
}
static bool inner(const Slot& s) { return s.active && rt_is_inner(s.tv.cur); }
static bool at_leaf(const Slot& s) { return s.active && s.tv.cur < 0; }

// One policy run over a queue of rays shared by `n_waves` waves that take turns (one outer iteration each), as the waves of a
// launch share a region's cursor: a wave's lanes then hold chunks from distant parts of the queue, as on the GPU.
// per = rays per lane (1 = k_trace as shipped; 2: a lane steps whichever of its rays needs the phase the wave is in).
struct Wave { std::vector<Slot> slot; bool done = false; };
static Counts simulate(const std::vector<Ray>& q, int n_waves, int per, int refill_slots, int node_ratio_num, int node_ratio_den, int leaf_ratio) {
    Counts c; c.rays = (long)q.size();
    std::vector<Wave> waves((size_t)n_waves);
    for (auto& w : waves) w.slot.resize((size_t)64 * per);
    size_t cursor = 0;
    int live_waves = n_waves;
    while (live_waves > 0) {
        for (auto& w : waves) {
            if (w.done) continue;
            std::vector<Slot>& slot = w.slot;
            auto lane_any = [&](int l, bool (*pred)(const Slot&)) { for (int k = 0; k < per; ++k) if (pred(slot[(size_t)l * per + k])) return true; return false; };
            int idle = 0, act = 0;
            for (auto& s : slot) (s.active ? act : idle)++;
            const bool exhausted = cursor >= q.size();
            if (!exhausted && (idle >= refill_slots || act == 0)) {
                for (int k = 0; k < per; ++k) {   // one pass of the refill code per ray slot index that has idle lanes
                    bool any = false;
                    for (int l = 0; l < 64; ++l) { Slot& s = slot[(size_t)l * per + k]; if (!s.active && cursor < q.size()) { begin(s, q[cursor++]); any = true; } }
                    if (any) c.refill += 1;
                }
                act = 0; for (auto& s : slot) act += s.active;
            }
            if (act == 0) { w.done = true; --live_waves; continue; }
            c.rounds += 1;
            int n_enter = 0; for (int l = 0; l < 64; ++l) n_enter += lane_any(l, inner);
            while (true) {
                int n_in = 0;
                for (int l = 0; l < 64; ++l) n_in += lane_any(l, inner);
                if (n_in == 0) break;
                c.node += 1; c.node_live += n_in;
                for (int l = 0; l < 64; ++l)
                    for (int k = 0; k < per; ++k) { Slot& s = slot[(size_t)l * per + k]; if (inner(s)) { node_step(s); break; } }
                n_in = 0; for (int l = 0; l < 64; ++l) n_in += lane_any(l, inner);
                if (n_in * node_ratio_num < (n_enter - n_in) * node_ratio_den) break;
            }
            for (auto& s : slot) if (s.active && s.tv.cur == RT_SENTINEL) rt_sentinel_step(s.wo, s.wd, s.tv, s.stk);
            int n_leaf_enter = 0; for (int l = 0; l < 64; ++l) n_leaf_enter += lane_any(l, at_leaf);
            while (n_leaf_enter > 0) {
                int n_lf = 0; for (int l = 0; l < 64; ++l) n_lf += lane_any(l, at_leaf);
                if (n_lf == 0) break;
                c.leaf += 1; c.leaf_live += n_lf;
                for (int l = 0; l < 64; ++l)
                    for (int k = 0; k < per; ++k) { Slot& s = slot[(size_t)l * per + k]; if (at_leaf(s)) { leaf_or_sentinel_step(s); break; } }
                n_lf = 0; for (int l = 0; l < 64; ++l) n_lf += lane_any(l, at_leaf);
                if (n_lf * leaf_ratio < n_leaf_enter - n_lf) break;
            }
            for (auto& s : slot) if (s.active && s.tv.cur == RT_EMPTY_STACK) s.active = false;
        }
    }
    return c;
}

int main(int argc, char** argv) {
    if (argc < 2) { std::fprintf(stderr, "usage: wave_sim <config.yml | Synthetic:N> [width] [spp] [wide 0|1] [waves sharing a queue]\n"); return 2; }
    const std::string what = argv[1];
    const int width = argc > 2 ? std::atoi(argv[2]) : 128, spp = argc > 3 ? std::atoi(argv[3]) : 4;
    g_wide = argc > 4 && std::atoi(argv[4]) != 0;
    const int n_waves = argc > 5 ? std::atoi(argv[5]) : 16;
    RtBuilder* b = nullptr; RtObj world = -1; rtb_create(&b);
    RtCamera cam; RtRenderParams p; std::memset(&p, 0, sizeof(p));
    if (what.rfind("Synthetic", 0) == 0) {
        if (rt_canned_scene(b, what.c_str(), 0x5EED0001ull, &world)) { std::fprintf(stderr, "%s\n", rt_last_error()); return 1; }
        double f[3] = {0, 260, 1500}, a[3] = {0, 60, 0}, u[3] = {0, 1, 0};
        std::memcpy(cam.look_from, f, 24); std::memcpy(cam.look_at, a, 24); std::memcpy(cam.v_up, u, 24);
        cam.vertical_fov_deg = 40; cam.aspect_ratio = 16.0 / 9; cam.aperture = 0.1; cam.focus_dist = 10; cam.shutter_open = 0; cam.shutter_close = 1;
        p.width = width; p.height = (int)(width * 9 / 16); p.max_depth = 50; p.background[0] = 0.7f; p.background[1] = 0.8f; p.background[2] = 1.0f;
    } else {
        RtConfig cfg;
        if (rt_config_load(what.c_str(), &cfg) || rt_config_build_scene(what.c_str(), 0x5EED0001ull, b, &world)) { std::fprintf(stderr, "%s\n", rt_last_error()); return 1; }
        std::memcpy(cam.look_from, cfg.look_from, 24); std::memcpy(cam.look_at, cfg.look_at, 24); std::memcpy(cam.v_up, cfg.v_up, 24);
        cam.vertical_fov_deg = cfg.vertical_fov_deg; cam.aspect_ratio = cfg.aspect_ratio; cam.aperture = cfg.aperture; cam.focus_dist = cfg.focus_distance;
        cam.shutter_open = cfg.shutter_open; cam.shutter_close = cfg.shutter_close;
        p.width = width; p.height = (int)(width / cfg.aspect_ratio + 0.5); p.max_depth = cfg.max_depth;
        std::memcpy(p.background, cfg.background, sizeof(p.background));
    }
    p.spp = spp; p.seed = 0x5EED0002ull;
    RtFlat* flat = nullptr;
    if (rt_flat_create(b, world, &flat)) { std::fprintf(stderr, "%s\n", rt_last_error()); return 1; }
    g_sc = *(const RtSceneView*)rt_flat_view(flat);
    // the ray segments of the render, per bounce, in path order (the order the queues roughly have)
    rt::CameraDesc d;
    d.look_from = rt::vec3(cam.look_from[0], cam.look_from[1], cam.look_from[2]); d.look_at = rt::vec3(cam.look_at[0], cam.look_at[1], cam.look_at[2]);
    d.v_up = rt::vec3(cam.v_up[0], cam.v_up[1], cam.v_up[2]); d.vertical_fov_deg = cam.vertical_fov_deg; d.aspect_ratio = cam.aspect_ratio;
    d.aperture = cam.aperture; d.focus_dist = cam.focus_dist; d.shutter_open = cam.shutter_open; d.shutter_close = cam.shutter_close;
    RtCameraDev cd; rt::make_camera(d, cd);
    RtFrameDev fr; rt_make_frame(&p, fr);
    std::vector<std::vector<Ray>> per_bounce((size_t)p.max_depth + 1);
    for (int s = 0; s < spp; ++s)
        for (uint32_t pixel = 0; pixel < (uint32_t)fr.n_pixels; ++pixel) {
            v3 o, dd; float time;
            rt_camera_ray(cd, fr, pixel, (uint32_t)s, o, dd, time);
            uint32_t src = RT_HIT_NONE;
            for (int bounce = 1; bounce <= p.max_depth; ++bounce) {
                Ray r; r.o = o; r.d = dd; r.time = time; r.src = src; r.bounce = bounce; r.pid = (uint32_t)s * (uint32_t)fr.n_pixels + pixel;
                per_bounce[(size_t)bounce].push_back(r);
                ArrStack stk; RtHit hit; hit.t = rt_inf(); hit.code = RT_HIT_NONE; hit.inst = -1;
                rt_closest_hit<false>(g_sc, o, dd, time, RT_TMIN, src, stk, hit, nullptr);
                uint32_t rnd[4]; rt_philox(pixel, (uint32_t)s, (uint32_t)bounce, 0u, fr.key0, fr.key1, rnd);
                if (g_sc.n_media > 0) rt_media_hits(g_sc, o, dd, time, RT_TMIN, pixel, (uint32_t)s, (uint32_t)bounce, fr.key0, fr.key1, rnd, hit);
                if (hit.code == RT_HIT_NONE) break;
                RtSurface sf; rt_surface(g_sc, o, dd, time, hit, false, sf);
                v3 atten, dir, emitted;
                if (!rt_scatter(g_sc, sf, dd, rt_u24(rnd[0]), rt_u24(rnd[1]), atten, dir, emitted)) break;
                o = sf.p; dd = dir; src = RT_HIT_KIND(hit.code) == RT_KIND_MEDIUM ? RT_HIT_NONE : hit.code;
            }
        }
    std::vector<Ray> all;   // bounce after bounce, as the kernel sees them (one launch per bounce: the wave drains between bounces)
    long total = 0; for (auto& v : per_bounce) total += (long)v.size();
    std::printf("%s %dx%d %d spp%s: %ld ray segments\n", what.c_str(), p.width, p.height, spp, g_wide ? " (4-wide tree)" : "", total);
    struct Pol { const char* name; int per, refill, nn, nd, lr; double cn, cl; };
    const double CN = g_wide ? 130.0 : 74.0, CL = g_wide ? 110.0 : 90.0, CR = 60.0, CO = 25.0;
    std::vector<Pol> pols = {
        {"shipped: 1 ray/lane", 1, g_wide ? 16 : 24, g_wide ? 1 : 2, 1, 1, 1.0, 1.0},
        {"2 rays/lane, refill at 48 idle slots", 2, 48, g_wide ? 1 : 2, 1, 1, 1.18, 1.12},
        {"2 rays/lane, refill at 32 idle slots", 2, 32, g_wide ? 1 : 2, 1, 1, 1.18, 1.12},
        {"2 rays/lane, refill 48, node ratio 1", 2, 48, 1, 1, 1, 1.18, 1.12},
        {"2 rays/lane, refill 48, node ratio 4", 2, 48, 4, 1, 1, 1.18, 1.12},
        {"2 rays/lane, refill 48, leaf ratio 2", 2, 48, g_wide ? 1 : 2, 1, 2, 1.18, 1.12},
    };
    for (const Pol& pol : pols) {
        Counts sum;
        for (auto& v : per_bounce) {
            if (v.empty()) continue;
            Counts c = simulate(v, n_waves, pol.per, pol.refill, pol.nn, pol.nd, pol.lr);
            sum.node += c.node; sum.leaf += c.leaf; sum.refill += c.refill; sum.rounds += c.rounds; sum.node_live += c.node_live; sum.leaf_live += c.leaf_live; sum.rays += c.rays;
        }
        const double per64 = 64.0 / (double)sum.rays;
        const double cost = sum.node * CN * pol.cn + sum.leaf * CL * pol.cl + sum.refill * CR + sum.rounds * CO;
        std::printf("%-40s per 64 rays: node steps %6.2f (live %.3f)  leaf steps %5.2f (live %.3f)  refills %4.2f  rounds %5.2f  -> %7.0f instr\n", pol.name,
                    sum.node * per64, sum.node_live / (64.0 * sum.node), sum.leaf * per64, sum.leaf_live / (64.0 * std::max(1.0, sum.leaf)), sum.refill * per64,
                    sum.rounds * per64, cost * per64);
    }
    return 0;
}

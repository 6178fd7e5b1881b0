// rt_emul.cpp -- TEST-ONLY.  Compiles the kernels' fp32 arithmetic (ray_tracing_amd/csrc/rt_device.h)
// for the host and drives it with plain loops, so that the flattener, the traversal and the scatter
// logic can be checked against the f64 oracle in a container without a GPU.  It is never loaded by
// the ray_tracing_amd package, never shipped, and is not a fallback of any kind: it exists so that a
// logic error is found on the CPU instead of costing a GPU run.
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include "../../include/rt_hip.h"
#include "../../ray_tracing_amd/csrc/frame.h"
#include "../../ray_tracing_amd/csrc/rt_device.h"
#include "../../ray_tracing_amd/csrc/scene_builder.h"

struct ArrStack {
    int v[128];   // the 4-wide tree pushes up to three children per node: deeper than RT_STACK_DEPTH
    void put(int i, int, int, int code) { v[i] = code; }
    void put_sentinel(int i) { v[i] = RT_SENTINEL; }
    int take(int i) const { return v[i]; }
    int cap() const { return 128; }
};
// EMUL_WIDE_BVH=1 in the environment: every closest-hit search goes through the 4-wide quantised tree instead of the binary one.
static bool use_wide() { const char* e = std::getenv("EMUL_WIDE_BVH"); return e && std::atoi(e) != 0; }
template <class Stack>
static void closest(const RtSceneView& sc, v3 o, v3 d, float time, uint32_t src, Stack& stk, RtHit& hit) {
    if (use_wide()) rt_closest_hit_wide<false>(sc, o, d, time, RT_TMIN, src, stk, hit, nullptr);
    else rt_closest_hit<false>(sc, o, d, time, RT_TMIN, src, stk, hit, nullptr);
}

static void make_cam(const RtCamera* cam, RtCameraDev& cd) {
    rt::CameraDesc d;
    d.look_from = rt::vec3(cam->look_from[0], cam->look_from[1], cam->look_from[2]);
    d.look_at = rt::vec3(cam->look_at[0], cam->look_at[1], cam->look_at[2]);
    d.v_up = rt::vec3(cam->v_up[0], cam->v_up[1], cam->v_up[2]);
    d.vertical_fov_deg = cam->vertical_fov_deg; d.aspect_ratio = cam->aspect_ratio; d.aperture = cam->aperture;
    d.focus_dist = cam->focus_dist; d.shutter_open = cam->shutter_open; d.shutter_close = cam->shutter_close;
    rt::make_camera(d, cd);
}
static void make_frame(const RtRenderParams* p, RtFrameDev& fr) {
    rt_make_frame(p, fr);
    fr.sample_base = p->sample_offset; fr.spp_chunk = p->spp;
}

extern "C" {

void emul_camera_dev(const RtCamera* cam, float out[21]) {
    RtCameraDev cd; make_cam(cam, cd);
    std::memcpy(out, &cd, sizeof(float) * 21);
}

void emul_trace_primary(const RtSceneView* sc, const RtCamera* cam, const RtRenderParams* p, int32_t sample, int32_t* ids, float* ts,
                        float* normals, int32_t* fronts, float* uvs) {
    RtCameraDev cd; make_cam(cam, cd);
    RtFrameDev fr; make_frame(p, fr);
    for (uint32_t pixel = 0; pixel < (uint32_t)fr.n_pixels; ++pixel) {
        v3 o, d; float time;
        rt_camera_ray(cd, fr, pixel, (uint32_t)sample, o, d, time);
        ArrStack stk;
        RtHit hit; hit.t = rt_inf(); hit.code = RT_HIT_NONE; hit.inst = -1;
        closest(*sc, o, d, time, RT_HIT_NONE, stk, hit);
        if (sc->n_media > 0) {
            uint32_t rnd[4]; rt_philox(pixel, (uint32_t)sample, 1u, 0u, fr.key0, fr.key1, rnd);
            rt_media_hits(*sc, o, d, time, RT_TMIN, pixel, (uint32_t)sample, 1u, fr.key0, fr.key1, rnd, hit);
        }
        RtSurface sf; sf.id = -1; sf.n = mk3(0, 0, 0); sf.front = false; sf.u = sf.v = 0;
        if (hit.code != RT_HIT_NONE) rt_surface(*sc, o, d, time, hit, true, sf);
        ids[pixel] = sf.id; ts[pixel] = hit.t;
        if (normals) { normals[3 * pixel] = sf.n.x; normals[3 * pixel + 1] = sf.n.y; normals[3 * pixel + 2] = sf.n.z; }
        if (fronts) fronts[pixel] = sf.front;
        if (uvs) { uvs[2 * pixel] = sf.u; uvs[2 * pixel + 1] = sf.v; }
    }
}

// Same per-path arithmetic as k_raygen + k_bounce + k_accumulate, one path at a time.
void emul_render(const RtSceneView* sc, const RtCamera* cam, const RtRenderParams* p, int n_threads, float* accum, float* accum_sq, uint64_t* rays_out) {
    RtCameraDev cd; make_cam(cam, cd);
    RtFrameDev fr; make_frame(p, fr);
    if (n_threads < 1) n_threads = 1;
    std::vector<uint64_t> rays((size_t)n_threads, 0);
    auto worker = [&](int tid) {
        for (uint32_t k = (uint32_t)tid; k < (uint32_t)fr.n_set; k += (uint32_t)n_threads) {   // the pixels of the call's set, as the kernels number them
            uint32_t pixel, s0;
            rt_split_path(fr, k, pixel, s0);
            float sum[3] = {0, 0, 0}, sq[3] = {0, 0, 0};
            for (int s = 0; s < p->spp; ++s) {
                uint32_t sample = (uint32_t)(p->sample_offset + s);
                v3 o, d; float time;
                rt_camera_ray(cd, fr, pixel, sample, o, d, time);
                v3 beta = mk3(1, 1, 1), L = mk3(0, 0, 0);
                uint32_t src = RT_HIT_NONE;
                for (int bounce = 1; bounce <= p->max_depth; ++bounce) {
                    rays[(size_t)tid]++;
                    ArrStack stk;
                    RtHit hit; hit.t = rt_inf(); hit.code = RT_HIT_NONE; hit.inst = -1;
                    closest(*sc, o, d, time, src, stk, hit);
                    uint32_t rnd[4]; rt_philox(pixel, sample, (uint32_t)bounce, 0u, fr.key0, fr.key1, rnd);
                    if (sc->n_media > 0) rt_media_hits(*sc, o, d, time, RT_TMIN, pixel, sample, (uint32_t)bounce, fr.key0, fr.key1, rnd, hit);
                    if (hit.code == RT_HIT_NONE) { L = L + beta * mk3(fr.background[0], fr.background[1], fr.background[2]); break; }
                    RtSurface sf; rt_surface(*sc, o, d, time, hit, false, sf);
                    v3 atten, dir, emitted;
                    bool ok = rt_scatter(*sc, sf, d, rt_u24(rnd[0]), rt_u24(rnd[1]), atten, dir, emitted);
                    L = L + beta * emitted;
                    if (!ok) break;
                    beta = beta * atten; o = sf.p; d = dir;
                    src = RT_HIT_KIND(hit.code) == RT_KIND_MEDIUM ? RT_HIT_NONE : hit.code;
                }
                sum[0] += L.x; sum[1] += L.y; sum[2] += L.z;
                sq[0] += L.x * L.x; sq[1] += L.y * L.y; sq[2] += L.z * L.z;
            }
            for (int c = 0; c < 3; ++c) { accum[3 * (size_t)pixel + c] += sum[c]; if (accum_sq) accum_sq[3 * (size_t)pixel + c] += sq[c]; }
        }
    };
    std::vector<std::thread> th;
    for (int i = 1; i < n_threads; ++i) th.emplace_back(worker, i);
    worker(0);
    for (auto& t : th) t.join();
    uint64_t tot = 0; for (uint64_t r : rays) tot += r;
    if (rays_out) *rays_out = tot;
}

void emul_scatter_rays(const RtSceneView* sc, const float* rays, int32_t n, uint32_t pixel0, uint32_t sample, int32_t bounce, uint64_t seed, float* out) {
    for (int i = 0; i < n; ++i) {
        const float* r = rays + 7 * (size_t)i;
        v3 o = mk3(r[0], r[1], r[2]), d = mk3(r[3], r[4], r[5]); float time = r[6];
        uint32_t pixel = pixel0 + (uint32_t)i, key0 = (uint32_t)seed, key1 = (uint32_t)(seed >> 32);
        ArrStack stk;
        RtHit hit; hit.t = rt_inf(); hit.code = RT_HIT_NONE; hit.inst = -1;
        closest(*sc, o, d, time, RT_HIT_NONE, stk, hit);
        uint32_t rnd[4]; rt_philox(pixel, sample, (uint32_t)bounce, 0u, key0, key1, rnd);
        if (sc->n_media > 0) rt_media_hits(*sc, o, d, time, RT_TMIN, pixel, sample, (uint32_t)bounce, key0, key1, rnd, hit);
        float* w = out + 14 * (size_t)i;
        for (int k = 0; k < 14; ++k) w[k] = 0.0f;
        if (hit.code == RT_HIT_NONE) { w[0] = -1.0f; continue; }
        RtSurface sf; rt_surface(*sc, o, d, time, hit, false, sf);
        v3 atten, dir, emitted;
        bool ok = rt_scatter(*sc, sf, d, rt_u24(rnd[0]), rt_u24(rnd[1]), atten, dir, emitted);
        w[0] = ok ? 1.0f : 0.0f;
        w[1] = emitted.x; w[2] = emitted.y; w[3] = emitted.z; w[4] = atten.x; w[5] = atten.y; w[6] = atten.z;
        w[7] = sf.p.x; w[8] = sf.p.y; w[9] = sf.p.z; w[10] = dir.x; w[11] = dir.y; w[12] = dir.z; w[13] = time;
    }
}

}  // extern "C"

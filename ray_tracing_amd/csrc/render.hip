// render.hip -- gfx950 kernels and the host code that drives them (the "rt_*" half of include/rt_hip.h).
//
// Wavefront organisation of one pass over P = W*H*spp_chunk paths (replaces src/main.rs:567-599):
//   k_raygen      : Philox camera rays (pixel jitter, lens, shutter time) -> ray queue 0
//   per bounce    : k_trace  closest surface hit of every queued ray (two-level BVH, the scene or its top levels in LDS, persistent
//                            waves that refill finished lanes with new rays) -> hit record per ray
//                   k_shade  media, hit record, emission, scatter; append of the scattered ray to the other queue
//                            through a wave-aggregated atomic (one atomic per 64-lane wave, ranks from ballot +
//                            mbcnt); terminated paths add their radiance to their slot of the contribution buffer
//   k_accumulate  : per pixel, sums the pass's contributions in sample order into the caller's
//                   accumulation buffer (deterministic; no float atomics anywhere)
// A ray queue is RT_SHARDS regions of `cap` slots, each with its own counter: a single append counter saturates
// at ~88 M returning atomics/s on this GPU (= 5.6 Grays/s at one atomic per 64-lane wave), 16 counters do not.
// Block (x, y) of the bounce grid reads region y and appends to region (x + y) % RT_SHARDS; with gridDim.x a
// multiple of RT_SHARDS no region can receive more than n/RT_SHARDS + B * (RT_SHARDS + 1) rays (B = rays per block
// and grid-stride step: 256 for k_shade, 1024 for k_shade_lds).
// Queue sizes live in device memory (counts[b][s] = rays of region s entering bounce b), so a pass runs
// without a host round trip; the counts double as the ray-segment statistics.
// A region is filled from both ends.  Rays that need the closest-hit kernel are appended from slot 0 upwards (count: counts[b][s]);
// rays whose closest surface hit the shading kernel has settled itself (shade_ray: steps of a walk inside a medium whose boundary
// encloses a known handful of surfaces) are appended from slot cap - 1 downwards (count: back[b][s]) together with their hit
// record.  k_trace walks the front part only, so it never meets a ray it has nothing to do for; the shading kernels walk both.
// The capacity bound above is on the sum of the two parts.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "capi_internal.h"
#include "frame.h"
#include "rt_device.h"

#define RT_BLOCK 256
#define RT_SHARDS 16
#define RT_COUNT_STRIDE 32   /* uint32 words between two region counters: one 128-B line each, so the atomics spread over L2 channels */
#define RT_MAX_BOUNCES 1024
#ifndef RT_NODE_LEAVE_RATIO
#define RT_NODE_LEAVE_RATIO 2
#endif
#ifndef RT_NODE_LEAVE_RATIO_WIDE   /* the same vote for 4-wide nodes, whose steps are twice as long: leave earlier */
#define RT_NODE_LEAVE_RATIO_WIDE 1
#endif
#ifndef RT_NODE_LEAVE_DEN_WIDE
#define RT_NODE_LEAVE_DEN_WIDE 1
#endif
#ifndef RT_LEAF_LEAVE_RATIO
#define RT_LEAF_LEAVE_RATIO 1
#endif

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return rt_set_error(e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP,                    \
                                std::string(#expr) + ": " + hipGetErrorString(e_));                     \
    } while (0)

// ------------------------------------------------------------------ device side
struct LdsStack {  // one column of the block's LDS stack: entry i of thread t lives at [i][t]; stride = threads per block
    int* base; int depth; int stride;
    __host__ __device__ void put(int i, int, int, int code) { base[i * stride] = code; }
    __host__ __device__ void put_sentinel(int i) { base[i * stride] = RT_SENTINEL; }
    __host__ __device__ int take(int i) const { return base[i * stride]; }
    __host__ __device__ int cap() const { return depth; }
};

__global__ __launch_bounds__(RT_BLOCK) void k_raygen(RtCameraDev cam, RtFrameDev fr, uint32_t n_paths, uint32_t cap, uint32_t* counts1, uint32_t* overflow,
                                                     rt_f4* qa, rt_f4* qb, rt_f4* contrib) {
    // Chunks of 256 consecutive path ids are dealt round-robin to the regions.  A camera ray's third record (throughput 1,
    // path id) follows from where it sits in the queue (rt_first_c), so it is neither written here nor read by the first bounce.
    if (blockIdx.x == 0 && threadIdx.x < RT_SHARDS) {
        uint32_t n_chunks = (n_paths + RT_BLOCK - 1) / RT_BLOCK, sh = threadIdx.x;
        uint32_t mine = n_chunks / RT_SHARDS + (sh < n_chunks % RT_SHARDS ? 1u : 0u);
        uint32_t cnt = mine * RT_BLOCK;
        if (n_chunks > 0 && (n_chunks - 1) % RT_SHARDS == sh) cnt -= n_chunks * RT_BLOCK - n_paths;  // the last, partial chunk
        counts1[sh * RT_COUNT_STRIDE] = cnt;
    }
    for (uint32_t pid = blockIdx.x * RT_BLOCK + threadIdx.x; pid < n_paths; pid += gridDim.x * RT_BLOCK) {
        uint32_t pixel, s;
        rt_split_path(fr, pid, pixel, s);
        v3 o, d; float time;
        rt_camera_ray(cam, fr, pixel, (uint32_t)fr.sample_base + s, o, d, time);
        rt_f4 a, b;
        a.x = o.x; a.y = o.y; a.z = o.z; a.w = time;
        b.x = d.x; b.y = d.y; b.z = d.z; b.w = i2f((int)RT_HIT_NONE);
        uint32_t chunk = pid / RT_BLOCK;
        uint32_t in_region = (chunk / RT_SHARDS) * RT_BLOCK + pid % RT_BLOCK;
        *reinterpret_cast<float4*>(contrib + pid) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);   // the path's light, written where it ends (k_shade)
        if (in_region >= cap) { atomicAdd(overflow, 1u); continue; }                       // never with a workspace sized for the pass
        size_t slot = (size_t)(chunk % RT_SHARDS) * cap + in_region;
        *reinterpret_cast<float4*>(qa + slot) = make_float4(a.x, a.y, a.z, a.w);
        *reinterpret_cast<float4*>(qb + slot) = make_float4(b.x, b.y, b.z, b.w);
    }
}
// The third record of the camera ray at entry i of region `region` of the first bounce's queue: throughput 1 and the path id
// k_raygen placed there.
__device__ __forceinline__ rt_f4 rt_first_c(uint32_t region, uint32_t i) {
    rt_f4 c; c.x = 1.0f; c.y = 1.0f; c.z = 1.0f;
    c.w = i2f((int)(((i / RT_BLOCK) * RT_SHARDS + region) * RT_BLOCK + i % RT_BLOCK));
    return c;
}

// ---- k_trace: closest surface hit of every queued ray
// LDS plan of one k_trace block (host: plan_trace).  Offsets are in 16-byte records from the start of the block's
// dynamic LDS; an offset < 0 leaves that array in global memory.  Layout: [nodes][primitive arrays][stack].
struct RtTracePlan {
    int32_t n_nodes_lds;                                                  // nodes [first_lds, first_lds + n) are served from LDS
    int32_t first_lds;                                                    // 0 for a breadth-first array; a device-built tree keeps copies of its top at the end
    int32_t off_spheres, off_mspheres, off_rects, off_boxes, off_instances;
    int32_t off_stack;                                                    // first record of the stack
    int32_t stack_cap;                                                    // stack entries per lane held in LDS
    int32_t stack_total;                                                  // entries a lane may need in all; those beyond stack_cap live in global memory (spill)
    int32_t wide_stack;                                                   // 1: 32-bit entries, 0: 16-bit entries (TraceStack)
    int32_t resident;                                                     // 1: all nodes and all primitive arrays are in LDS
    int32_t refill_lanes;                                                 // idle lanes of a wave that trigger a refill
    int32_t wide4;                                                        // 1: the nodes are the 4-wide quantised tree (sc.wnodes)
};

typedef float __attribute__((ext_vector_type(4))) rt_vf4;
typedef __attribute__((address_space(3))) const rt_vf4 lds_f4;   // keeps LDS reads ds_read_b128 (a generic pointer makes them flat loads)
typedef __attribute__((address_space(3))) int lds_int;
__device__ __forceinline__ rt_f4 lds_ld4(lds_f4* p) { rt_vf4 v = *p; rt_f4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r; }

// RESIDENT = the whole node array and every primitive array are in LDS: the accessors then have no global-memory path at
// all, which matters beyond the saved branch -- with no other VMEM load in the traversal loop the compiler has no reason
// to wait on vmcnt there, so the prefetch of the next ray batch really stays in flight.
template <bool RESIDENT, bool WIDE4> struct TraceNodes {   // node records: [first, first + n_lds) (the top of the tree) from LDS, the rest from global memory
    static constexpr bool kWide = WIDE4;                       // WIDE4: the records are 4-wide quantised nodes (sc.wnodes)
    const rt_f4* global; lds_f4* lds; int32_t n_lds, first; const int32_t* inst_roots;
    __device__ __forceinline__ void get(int32_t n, rt_f4 q[4]) const {
        if (RESIDENT || (uint32_t)(n - first) < (uint32_t)n_lds) {
            lds_f4* p = lds + 4 * (RESIDENT ? n : n - first);
            q[0] = lds_ld4(p); q[1] = lds_ld4(p + 1); q[2] = lds_ld4(p + 2); q[3] = lds_ld4(p + 3);
        } else {
            const rt_f4* p = global + 4 * (size_t)n;
            q[0] = ld4(p); q[1] = ld4(p + 1); q[2] = ld4(p + 2); q[3] = ld4(p + 3);
        }
    }
    __device__ __forceinline__ int32_t root(const RtSceneView& sc) const { return WIDE4 ? sc.wroot : sc.root; }
    __device__ __forceinline__ int32_t instance_root(int32_t i, int32_t binary_root) const { return WIDE4 ? inst_roots[i] : binary_root; }
};
template <bool RESIDENT> struct TracePrims {   // primitive records: every array is either wholly in LDS (offset >= 0) or read from global memory
    const RtSceneView* sc; lds_f4* lds; int32_t o_sph, o_msph, o_rect, o_box, o_inst;
    __device__ __forceinline__ rt_f4 sphere(uint32_t i) const { return (RESIDENT || o_sph >= 0) ? lds_ld4(lds + o_sph + i) : ld4(sc->spheres + i); }
    __device__ __forceinline__ void msphere(uint32_t i, rt_f4& m0, rt_f4& m1, rt_f4& m2) const {
        if (RESIDENT || o_msph >= 0) { lds_f4* m = lds + o_msph + 3 * i; m0 = lds_ld4(m); m1 = lds_ld4(m + 1); m2 = lds_ld4(m + 2); }
        else { const rt_f4* m = sc->mspheres + 3 * (size_t)i; m0 = ld4(m); m1 = ld4(m + 1); m2 = ld4(m + 2); }
    }
    __device__ __forceinline__ void pair(int32_t off, const rt_f4* g, uint32_t i, rt_f4& r0, rt_f4& r1) const {
        if (RESIDENT || off >= 0) { lds_f4* m = lds + off + 2 * i; r0 = lds_ld4(m); r1 = lds_ld4(m + 1); }
        else { r0 = ld4(g + 2 * (size_t)i); r1 = ld4(g + 2 * (size_t)i + 1); }
    }
    __device__ __forceinline__ void rect(uint32_t i, rt_f4& r0, rt_f4& r1) const { pair(o_rect, sc->rects, i, r0, r1); }
    __device__ __forceinline__ void box(uint32_t i, rt_f4& b0, rt_f4& b1) const { pair(o_box, sc->boxes, i, b0, b1); }
    __device__ __forceinline__ void instance(uint32_t i, rt_f4& i0, rt_f4& i1) const { pair(o_inst, sc->instances, i, i0, i1); }
};
// Per-lane traversal stack in LDS, [entry][thread].  WIDE = false: 16-bit entries that name the postponed child by its
// parent (node << 1 | which); the child code is re-read from the parent's record, which must be in LDS.  Half the LDS of
// the 32-bit form, which is what lets a small scene's primitives sit in LDS beside its nodes.
typedef __attribute__((address_space(3))) unsigned short lds_u16;
// Entries [0, depth) of a lane's stack are in LDS; deeper ones -- which few rays ever reach: the worst case follows from the
// tree's depth, the depth in use is a handful (0.07 % of final_scene.yml's node steps have more than 8 entries pending, 0.3 % of
// the 1M-sphere scene's more than 10) -- go to a global spill area, [entry][thread of the grid].  The LDS this frees holds scene data.
// SPILL = false: the whole stack is in LDS and the accessors have no global-memory path (a branch and a VMEM access in the node loop
// cost k_trace 11 % on final_scene.yml even when never taken).
template <bool WIDE, bool SPILL> struct TraceStack;
template <bool SPILL> struct TraceStack<true, SPILL> {
    lds_int* base; int depth, stride, total; int* spill; size_t spill_stride;
    __device__ __forceinline__ void put(int i, int, int, int code) { if (!SPILL || i < depth) base[i * stride] = code; else spill[(size_t)(i - depth) * spill_stride] = code; }
    __device__ __forceinline__ void put_sentinel(int i) { put(i, 0, 0, RT_SENTINEL); }
    __device__ __forceinline__ int take(int i) const { return (!SPILL || i < depth) ? base[i * stride] : spill[(size_t)(i - depth) * spill_stride]; }
    __device__ __forceinline__ int cap() const { return total; }
};
template <bool SPILL> struct TraceStack<false, SPILL> {
    lds_u16* base; lds_int* node_words; int depth, stride, total; unsigned short* spill; size_t spill_stride;
    __device__ __forceinline__ void put_raw(int i, unsigned short e) { if (!SPILL || i < depth) base[i * stride] = e; else spill[(size_t)(i - depth) * spill_stride] = e; }
    __device__ __forceinline__ void put(int i, int node, int which, int) { put_raw(i, (unsigned short)((node << 1) | which)); }
    __device__ __forceinline__ void put_sentinel(int i) { put_raw(i, (unsigned short)0xFFFFu); }
    __device__ __forceinline__ int take(int i) const {
        uint32_t e = (!SPILL || i < depth) ? (uint32_t)base[i * stride] : (uint32_t)spill[(size_t)(i - depth) * spill_stride];
        return e == 0xFFFFu ? RT_SENTINEL : node_words[(e >> 1) * 16u + 12u + (e & 1u)];
    }
    __device__ __forceinline__ int cap() const { return total; }
};
// spill: the scene's spill area (null when every entry is in LDS); `gthread` = this thread's index in the grid
template <bool WIDE, bool SPILL> struct MakeStack;
template <bool SPILL> struct MakeStack<true, SPILL> {
    static __device__ __forceinline__ TraceStack<true, SPILL> make(unsigned char* smem, const RtTracePlan& pl, int threads, void* spill, size_t gthread, size_t gthreads) {
        TraceStack<true, SPILL> s; s.base = (lds_int*)(smem + (size_t)pl.off_stack * 16) + threadIdx.x; s.depth = pl.stack_cap; s.total = pl.stack_total; s.stride = threads;
        s.spill = (int*)spill + gthread; s.spill_stride = gthreads; return s;
    }
};
template <bool SPILL> struct MakeStack<false, SPILL> {
    static __device__ __forceinline__ TraceStack<false, SPILL> make(unsigned char* smem, const RtTracePlan& pl, int threads, void* spill, size_t gthread, size_t gthreads) {
        TraceStack<false, SPILL> s; s.base = (lds_u16*)(smem + (size_t)pl.off_stack * 16) + threadIdx.x; s.node_words = (lds_int*)smem;
        s.depth = pl.stack_cap; s.total = pl.stack_total; s.stride = threads; s.spill = (unsigned short*)spill + gthread; s.spill_stride = gthreads; return s;
    }
};

// Persistent grid, one ray per lane.  The scene (node array and, when they fit, the primitive arrays) is copied into
// the block's LDS once; traversal then touches global memory only for the ray records in and the hit records out.
// A lane that has finished its ray waits until `refill_lanes` lanes of its wave are idle; the idle lanes then take the
// next rays of the region through one wave-aggregated atomic on the region's cursor (ballot + mbcnt ranks).  A wave
// therefore never drains to a few long rays, which is what a ray-per-thread launch costs on 64-wide waves.
template <bool STATS, int TB, bool WIDE, bool RESIDENT, bool WIDE4, bool SPILL>
__global__ __launch_bounds__(TB) void k_trace(RtSceneView sc, RtTracePlan plan, uint32_t cap, const uint32_t* counts_in, uint32_t* cursors,
                                              const rt_f4* ia, const rt_f4* ib, rt_f4* hits, unsigned long long* work, void* spill) {
    const uint32_t region = blockIdx.y;
    const uint32_t n = min(counts_in[region * RT_COUNT_STRIDE], cap);   // a counter can only exceed cap after an overflow (reported by the append)
    if (n == 0) return;
    extern __shared__ __align__(16) unsigned char smem[];
    {   // scene -> LDS
        rt_f4* dst = reinterpret_cast<rt_f4*>(smem);
        const rt_f4* node_src = WIDE4 ? sc.wnodes : sc.nodes;
        for (int i = threadIdx.x; i < plan.n_nodes_lds * 4; i += TB) dst[i] = ld4(node_src + 4 * (size_t)plan.first_lds + i);
        if (plan.off_spheres >= 0) for (int i = threadIdx.x; i < sc.n_spheres; i += TB) dst[plan.off_spheres + i] = ld4(sc.spheres + i);
        if (plan.off_mspheres >= 0) for (int i = threadIdx.x; i < sc.n_mspheres * 3; i += TB) dst[plan.off_mspheres + i] = ld4(sc.mspheres + i);
        if (plan.off_rects >= 0) for (int i = threadIdx.x; i < sc.n_rects * 2; i += TB) dst[plan.off_rects + i] = ld4(sc.rects + i);
        if (plan.off_boxes >= 0) for (int i = threadIdx.x; i < sc.n_boxes * 2; i += TB) dst[plan.off_boxes + i] = ld4(sc.boxes + i);
        if (plan.off_instances >= 0) for (int i = threadIdx.x; i < sc.n_instances * 2; i += TB) dst[plan.off_instances + i] = ld4(sc.instances + i);
    }
    __syncthreads();
    lds_f4* lds = (lds_f4*)smem;
    TraceNodes<RESIDENT, WIDE4> nodes; nodes.global = WIDE4 ? sc.wnodes : sc.nodes; nodes.lds = lds; nodes.n_lds = plan.n_nodes_lds; nodes.first = plan.first_lds;
    nodes.inst_roots = sc.winst_roots;
    TracePrims<RESIDENT> prims; prims.sc = &sc; prims.lds = lds;
    prims.o_sph = plan.off_spheres; prims.o_msph = plan.off_mspheres; prims.o_rect = plan.off_rects; prims.o_box = plan.off_boxes; prims.o_inst = plan.off_instances;
    TraceStack<WIDE, SPILL> stk = MakeStack<WIDE, SPILL>::make(smem, plan, TB, spill, ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * TB + threadIdx.x, (size_t)gridDim.x * gridDim.y * TB);

    uint32_t* cursor = cursors + region * RT_COUNT_STRIDE;
    ia += (size_t)region * cap; ib += (size_t)region * cap; hits += (size_t)region * cap;
    const uint32_t lane = threadIdx.x & 63u;

    bool active = false, exhausted = false;
    uint32_t slot = 0, src = RT_HIT_NONE;
    v3 wo = mk3(0, 0, 0), wd = mk3(0, 0, 1);
    float time = 0.0f;
    RtTrav tv; tv.o = wo; tv.d = wd; tv.inv = wd; tv.inst = -1; tv.sp = 0; tv.cur = RT_EMPTY_STACK;
    RtHit hit; hit.t = rt_inf(); hit.code = RT_HIT_NONE; hit.inst = -1;
    RtCounters cnt = {0, 0, 0, 0, 0, 0, 0};
    uint32_t dbg[4] = {0, 0, 0, 0};   // STATS: wave-level counts of outer iterations, refills, node-loop and leaf-loop iterations
    uint32_t sp_max = 0, sp_steps = 0, sp_gt[6] = {0, 0, 0, 0, 0, 0};   // STATS: deepest stack; node steps in all, and with more than 4 / 6 / 8 / 10 / 12 / 16 entries on the stack

    for (;;) {
        unsigned long long am = __ballot(active);
        int n_idle = 64 - __popcll(am);
        if (STATS && lane == 0) dbg[0]++;
        if (!exhausted && (n_idle >= plan.refill_lanes || am == 0)) {
            if (STATS && lane == 0) dbg[1]++;
            unsigned long long im = ~am;
            uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(im >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)im, 0u));
            uint32_t base = 0;
            if (!active && rank == 0) base = atomicAdd(cursor, (uint32_t)n_idle);
            base = __shfl(base, __ffsll((long long)im) - 1);
            if (!active) {
                uint32_t idx = base + rank;
                if (idx < n) {
                    rt_f4 a = ld4(ia + idx), b = ld4(ib + idx);
                    wo = mk3(a.x, a.y, a.z); wd = mk3(b.x, b.y, b.z); time = a.w; src = (uint32_t)f2i(b.w);
                    slot = idx; active = true;
                    hit.t = rt_inf(); hit.code = RT_HIT_NONE; hit.inst = -1;
                    rt_trav_begin(sc, nodes, wo, wd, tv);
                }
            }
            if (base + (uint32_t)n_idle >= n) exhausted = true;
            am = __ballot(active);
        }
        if (am == 0) break;
        if (active) {
            // A few lanes with long descents would keep the whole wave in the node loop: once the lanes still in it are fewer
            // than half of those that have left it (for a leaf, or done), they too go on to the leaf phase and the refill and
            // descend again with the next round (-5 % kernel time on final_scene.yml, -40 % on the 1M-sphere scene).
            const int n_enter = __popcll(__ballot(rt_is_inner(tv.cur)));
            while (rt_is_inner(tv.cur)) {
                if (STATS) { unsigned long long ex = __ballot(true); if ((int)lane == __ffsll((long long)ex) - 1) dbg[2]++; }
                rt_inner_step<STATS>(nodes, tv, RT_TMIN, hit, stk, &cnt);
                if (STATS) {
                    const uint32_t d = (uint32_t)tv.sp;
                    sp_max = max(sp_max, d); sp_steps++;
                    sp_gt[0] += d > 4; sp_gt[1] += d > 6; sp_gt[2] += d > 8; sp_gt[3] += d > 10; sp_gt[4] += d > 12; sp_gt[5] += d > 16;
                }
                { int n_in = __popcll(__ballot(rt_is_inner(tv.cur)));
                  if (WIDE4 ? (n_in * RT_NODE_LEAVE_RATIO_WIDE < (n_enter - n_in) * RT_NODE_LEAVE_DEN_WIDE) : (n_in * RT_NODE_LEAVE_RATIO < n_enter - n_in)) break; }
            }
            if (tv.cur >= 0) {
                if (tv.cur == RT_SENTINEL) rt_sentinel_step(wo, wd, tv, stk);
            } else {
                const int n_leaf_enter = __popcll(__ballot(true));   // the same for runs of leaves (-7 %)
                do {
                    if (STATS) { unsigned long long ex = __ballot(true); if ((int)lane == __ffsll((long long)ex) - 1) dbg[3]++; }
#ifdef RT_LEAF_TWICE   /* timing experiment: the leaf step run twice with the same result */
                    { RtTrav t2 = tv; RtHit h2 = hit;
                      rt_leaf_step<false>(nodes, prims, wo, wd, time, RT_TMIN, src, t2, h2, stk, &cnt);
                      asm volatile("" : "+v"(wo.x), "+v"(wd.x), "+v"(h2.t));
                      if (h2.t < 0.0f) hit.t = h2.t; }
#endif
                    rt_leaf_step<STATS>(nodes, prims, wo, wd, time, RT_TMIN, src, tv, hit, stk, &cnt);
                    { int n_lf = __popcll(__ballot(tv.cur < 0)); if (n_lf * RT_LEAF_LEAVE_RATIO < n_leaf_enter - n_lf) break; }
                } while (tv.cur < 0);
            }
            if (tv.cur == RT_EMPTY_STACK) {
                *reinterpret_cast<float4*>(hits + slot) = make_float4(hit.t, i2f((int)hit.code), i2f(hit.inst), 0.0f);
                active = false;
            }
        }
    }
    if (STATS) {
        atomicAdd(work + 0, (unsigned long long)cnt.nodes); atomicAdd(work + 1, (unsigned long long)cnt.spheres);
        atomicAdd(work + 2, (unsigned long long)cnt.mspheres); atomicAdd(work + 3, (unsigned long long)cnt.rects);
        atomicAdd(work + 4, (unsigned long long)cnt.instances); atomicAdd(work + 7, (unsigned long long)cnt.boxes);
        if (WIDE4) atomicAdd(work + 15, (unsigned long long)cnt.wnodes);
        for (int k = 0; k < 4; ++k) if (dbg[k]) atomicAdd(work + 16 + k, (unsigned long long)dbg[k]);
        atomicMax(work + 20, (unsigned long long)sp_max); atomicAdd(work + 21, (unsigned long long)sp_steps);
        for (int k = 0; k < 6; ++k) atomicAdd(work + 22 + k, (unsigned long long)sp_gt[k]);
    }
}

#define RT_SHADE_MEDIA_LDS 8
#define RT_FLAT_LEAVES 8      /* most leaves a tree may have for the linear scan of the fused kernel */
#define RT_FLAT_PRIMS 10      /* ... and most primitives */
struct ShadeMedia {   // medium records: the first RT_SHADE_MEDIA_LDS media from the block's LDS copy
    const rt_f4* global; lds_f4* lds;
    __device__ __forceinline__ rt_f4 get(int m, int k) const {
        return m < RT_SHADE_MEDIA_LDS ? lds_ld4(lds + RT_MEDIUM_F4 * m + k) : ld4(global + RT_MEDIUM_F4 * (size_t)m + k);
    }
};

// ---- k_shade / k_shade_lds: stage 2 of a bounce
// Media (mediums.rs:39-87), hit record, emission and scatter (lib.rs:64-71) of one ray: a = (origin, time), b = (direction,
// source primitive), c = (throughput, path id), h = (t, hit code, instance).  Light that reaches the path goes to its
// contribution slot; out.alive = the path continues with the ray out.na / nb / nc.  out.postponed = the hit needs a costly
// texture and `may_postpone` was set: nothing has been written for the ray then.
// Tab: see RtGlobalTables; MediaSrc: see rt_media_hits.  STATS = true also counts shades per material kind into `work`.
// The surfaces that lie wholly inside the boundary of a medium that lists its local surfaces (flatten.cpp, emit_medium_locals), over
// all media, at most four: a ray that starts ON one of them is inside that boundary as much as one that starts at a collision in
// the medium.  code / mask: the surface's hit code and the bits that identify it (a box's side index is masked out).
struct RtLocalSurfs { int32_t enabled, n; uint32_t code[4], mask[4]; int32_t medium[4]; };
struct ShadeOut { bool alive, postponed, resolved; int cls; rt_f4 na, nb, nc, nh; };   // resolved: nh is the scattered ray's closest surface hit
// Classes of hits that a wave may set aside and shade together later (k_shade_lds): 0 = the material's texture tree holds a
// Noise (Perlin turbulence, ~1300 instructions).  RT_POSTPONE_CLASSES = 2 adds dielectrics, 3 also metals (measured: not worth
// it -- a postponed hit repeats the medium tests and the hit record, which cost more than the divergence they avoid).
#ifndef RT_POSTPONE_CLASSES
#define RT_POSTPONE_CLASSES 1
#endif
template <bool STATS, class Tab, class MediaSrc>
__device__ __forceinline__ void shade_ray(const RtSceneView& sc, const Tab& tab, const MediaSrc& msrc, const RtFrameDev& fr, int bounce,
                                          rt_f4 a, rt_f4 b, rt_f4 c, rt_f4 h, bool may_postpone, const RtLocalSurfs& ls, rt_f4* contrib,
                                          unsigned long long* work, ShadeOut& out) {
    v3 o = mk3(a.x, a.y, a.z), d = mk3(b.x, b.y, b.z), beta = mk3(c.x, c.y, c.z);
    float time = a.w;
    uint32_t pid = (uint32_t)f2i(c.w);
    uint32_t pixel, sample;
    rt_split_path(fr, pid, pixel, sample);
    sample += (uint32_t)fr.sample_base;
    uint32_t rnd[4];
    rt_philox(pixel, sample, (uint32_t)bounce, 0u, fr.key0, fr.key1, rnd);

    RtHit hit; hit.t = h.x; hit.code = (uint32_t)f2i(h.y); hit.inst = f2i(h.z);
    if (sc.n_media > 0) rt_media_hits(sc, msrc, o, d, time, RT_TMIN, pixel, sample, (uint32_t)bounce, fr.key0, fr.key1, rnd, hit);

    v3 add = mk3(0, 0, 0);
    if (hit.code == RT_HIT_NONE) {
        if (STATS) { atomicAdd(work + 5, (unsigned long long)sc.n_media); atomicAdd(work + 6, 1ull); }
        add = beta * mk3(fr.background[0], fr.background[1], fr.background[2]);  // lib.rs:76
    } else {
        RtSurface sf;
        rt_surface(tab, o, d, time, hit, false, sf);
        rt_f4 a0, a1;
        tab.material(sf.mat, a0, a1);
        if (may_postpone) {
            const int mtype_ = f2i(a0.x);
            if (f2i(a1.w) != 0) { out.postponed = true; out.cls = 0; }
            else if (RT_POSTPONE_CLASSES > 1 && mtype_ == RT_MAT_DIELECTRIC) { out.postponed = true; out.cls = 1; }
            else if (RT_POSTPONE_CLASSES > 2 && mtype_ == RT_MAT_METAL) { out.postponed = true; out.cls = 2; }
        }
        if (!out.postponed) {
            v3 atten, dir, emitted;
            bool scattered = rt_scatter(tab, sf, d, rt_u24(rnd[0]), rt_u24(rnd[1]), atten, dir, emitted);
            if (STATS) {
                atomicAdd(work + 5, (unsigned long long)sc.n_media);
                int mtype = f2i(a0.x), mtex = f2i(a0.y);
                atomicAdd(work + 8 + mtype, 1ull);                                     // shades per material kind
                if (mtex >= 0 && mtype != RT_MAT_METAL && mtype != RT_MAT_DIELECTRIC) { // texture kind behind the material
                    int tt = f2i(ld4(sc.textures + 2 * (size_t)mtex).x);
                    if (tt == RT_TEX_NOISE) atomicAdd(work + 13, 1ull); else if (tt == RT_TEX_CHECKER) atomicAdd(work + 14, 1ull);
                }
            }
            add = beta * emitted;                                                      // lib.rs:64,66
            if (scattered && bounce < fr.max_depth) {                                 // lib.rs:58-60
                beta = beta * atten;
                out.alive = true;
                out.na.x = sf.p.x; out.na.y = sf.p.y; out.na.z = sf.p.z; out.na.w = time;
                out.nb.x = dir.x; out.nb.y = dir.y; out.nb.z = dir.z;
                out.nb.w = i2f(RT_HIT_KIND(hit.code) == RT_KIND_MEDIUM ? (int)RT_HIT_NONE : (int)hit.code);
                out.nc.x = beta.x; out.nc.y = beta.y; out.nc.z = beta.z; out.nc.w = c.w;
                // A ray that starts inside the boundary of a medium which encloses a known handful of surfaces (flatten.cpp,
                // emit_medium_locals; those that lie wholly inside the boundary come first in the list) -- at a collision in the
                // medium, or on one of those wholly-inside surfaces: if it hits one of these anywhere, everything up to that hit
                // is inside the boundary, so the nearest hit among the list's entries is its closest surface hit -- found by the
                // same tests with the same operands as the tree search would make, so the same record -- and the closest-hit
                // kernel never sees the ray.  (final_scene.yml: a third of all rays are such steps of walks inside the glass ball.)
                if (ls.enabled) {
                    int m = -1;
                    uint32_t from = RT_HIT_NONE;                       // the surface the ray starts on, as the tree search is told (k_trace's src)
                    if (RT_HIT_KIND(hit.code) == RT_KIND_MEDIUM) m = (int)RT_HIT_INDEX(hit.code);
                    else if (hit.inst < 0) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) if (k < ls.n && (hit.code & ls.mask[k]) == ls.code[k]) m = ls.medium[k];   // unrolled: indexed by a variable, the arrays would be copied to scratch memory
                        from = hit.code;
                    }
                    const rt_f4 m5 = msrc.get(max(m, 0), 5);
                    const int n_local = m >= 0 ? (f2i(m5.z) & 255) : 0, inside_mask = f2i(m5.z) >> 8;
                    if (n_local > 0) {
                        const rt_f4 m4 = msrc.get(m, 4);
                        RtTrav lt; rt_trav_ray(lt, sf.p, dir); lt.inst = -1; lt.sp = 0; lt.cur = RT_EMPTY_STACK;
                        RtHit nh; nh.t = rt_inf(); nh.code = RT_HIT_NONE; nh.inst = -1;
                        rt_leaf_test<false>(tab, f2i(m4.z), time, RT_TMIN, from, lt, nh, nullptr);
                        bool inside_hit = nh.code != RT_HIT_NONE;          // entry 0 is always a flagged one
                        if (n_local > 1) {
                            const float t0 = nh.t;
                            rt_leaf_test<false>(tab, f2i(m4.w), time, RT_TMIN, from, lt, nh, nullptr);
                            // a flagged entry tested after a hit on another flagged one need not be hit itself: one is enough
                            if ((inside_mask & 2) && nh.t < t0) inside_hit = true;
                        }
                        if (n_local > 2) {
                            const float t0 = nh.t;
                            rt_leaf_test<false>(tab, f2i(m5.w), time, RT_TMIN, from, lt, nh, nullptr);
                            if ((inside_mask & 4) && nh.t < t0) inside_hit = true;
                        }
                        if (inside_hit) {
                            out.resolved = true;
                            out.nh.x = nh.t; out.nh.y = i2f((int)nh.code); out.nh.z = i2f(-1); out.nh.w = 0.0f;
                            out.nb.w = i2f((int)RT_SRC_RESOLVED);
                            if (STATS) atomicAdd(work + 28, 1ull);
                        }
                    }
                }
            }
        }
    }
    if (add.x != 0.0f || add.y != 0.0f || add.z != 0.0f) {
        // Light reaches a path where it ends (an emitter does not scatter, a miss has no next segment), so the
        // path's slot, zeroed before the pass, is written exactly once; only a surface that both emits and
        // scatters would have to add to an earlier value.
        float4* cp = reinterpret_cast<float4*>(contrib + pid);
        float4 cv = make_float4(add.x, add.y, add.z, 0.0f);
        if (out.alive) { float4 old = *cp; cv.x += old.x; cv.y += old.y; cv.z += old.z; }
        *cp = cv;
    }
}
__device__ __forceinline__ void shade_out_clear(ShadeOut& out) {
    out.alive = false; out.postponed = false; out.resolved = false; out.cls = 0;
    out.na.x = out.na.y = out.na.z = out.na.w = out.nb.x = out.nb.y = out.nb.z = out.nb.w = out.nc.x = out.nc.y = out.nc.z = out.nc.w = 0.0f;
}
// Append of the scattered rays of one wave to the output region: rays for the closest-hit kernel from the front, settled rays
// (out.resolved, with their hit record) from the back; one atomic per wave and end, slot = base + rank among the lanes of that
// kind (ballot + mbcnt).  Wave-uniform control flow only.  Returns whether any lane appended; the wave has then waited for every
// vector memory operation it had issued before this call (the atomics' results are the last of them).
__device__ __forceinline__ bool shade_append(const ShadeOut& out, uint32_t* n_out_ptr, uint32_t* n_back_ptr, uint32_t* overflow, uint32_t cap, rt_f4* oa, rt_f4* ob,
                                             rt_f4* oc, rt_f4* oh) {
    const bool to_back = out.alive && out.resolved, to_front = out.alive && !out.resolved;
    const unsigned long long mf = __ballot(to_front), mb = __ballot(to_back);
    if (!(mf | mb)) return false;
    const uint32_t rank_f = __builtin_amdgcn_mbcnt_hi((uint32_t)(mf >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mf, 0u));
    const uint32_t rank_b = __builtin_amdgcn_mbcnt_hi((uint32_t)(mb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mb, 0u));
    uint32_t f0 = 0, b0 = 0;
    if (to_front && rank_f == 0) f0 = atomicAdd(n_out_ptr, (uint32_t)__popcll(mf));
    if (to_back && rank_b == 0) b0 = atomicAdd(n_back_ptr, (uint32_t)__popcll(mb));     // both in flight before the wait
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (mf) f0 = __shfl(f0, __ffsll((long long)mf) - 1);
    if (mb) b0 = __shfl(b0, __ffsll((long long)mb) - 1);
    // The capacity bound holds by construction (see the top of the file); should it ever fail, the ray is dropped, the
    // overflow counter says so and the render call returns RT_ERR_INTERNAL instead of a biased image.  (That the two ends have
    // not met is checked by the kernels that read the region.)
    const uint32_t pos = to_back ? b0 + rank_b : f0 + rank_f;
    if (out.alive && pos >= cap) atomicAdd(overflow, 1u);
    else if (out.alive) {
        const uint32_t slot = to_back ? cap - 1u - pos : pos;
        *reinterpret_cast<float4*>(oa + slot) = make_float4(out.na.x, out.na.y, out.na.z, out.na.w);
        *reinterpret_cast<float4*>(ob + slot) = make_float4(out.nb.x, out.nb.y, out.nb.z, out.nb.w);
        *reinterpret_cast<float4*>(oc + slot) = make_float4(out.nc.x, out.nc.y, out.nc.z, out.nc.w);
        if (to_back) *reinterpret_cast<float4*>(oh + slot) = make_float4(out.nh.x, out.nh.y, out.nh.z, out.nh.w);   // the next bounce's hit record
    }
    return true;
}
// The two parts of an input region as one index range: entry v of n_front + n_back is slot v of the front part, or slot
// cap - 1 - (v - n_front) of the back part.  *n_front is in, n_back is clamped so that the parts cannot overlap (an overlap is
// an overflow and is reported).
__device__ __forceinline__ uint32_t region_back_count(const uint32_t* back_in, uint32_t region, uint32_t n_front, uint32_t cap, uint32_t* overflow) {
    if (!back_in) return 0u;
    uint32_t n_back = back_in[region * RT_COUNT_STRIDE];
    if (n_back > cap - n_front) {
        if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(overflow, 1u);
        n_back = cap - n_front;
    }
    return n_back;
}
__device__ __forceinline__ uint32_t region_slot(uint32_t v, uint32_t n_front, uint32_t cap) { return v < n_front ? v : cap - 1u - (v - n_front); }

// One ray per lane, grid-stride, tables from global memory (any scene).  The input queue region is blockIdx.y.
template <bool STATS>
__global__ __launch_bounds__(RT_BLOCK) void k_shade(RtSceneView sc, RtFrameDev fr, int bounce, uint32_t cap, const uint32_t* counts_in,
                                                    uint32_t* counts_out, uint32_t* overflow, const rt_f4* ia, const rt_f4* ib, const rt_f4* ic,
                                                    const rt_f4* hits, rt_f4* oa, rt_f4* ob, rt_f4* oc, rt_f4* contrib,
                                                    unsigned long long* work, rt_f4* hits_out, const uint32_t* back_in, uint32_t* back_out, RtLocalSurfs ls) {
    const uint32_t n_front = min(counts_in[blockIdx.y * RT_COUNT_STRIDE], cap);
    const uint32_t n = n_front + region_back_count(back_in, blockIdx.y, n_front, cap, overflow);
    if (blockIdx.x * RT_BLOCK >= n) return;
    // the medium records are the same for every ray: one LDS copy per block instead of a chain of fetches per ray
    __shared__ __align__(16) rt_f4 s_media[RT_SHADE_MEDIA_LDS * RT_MEDIUM_F4];
    for (int i = threadIdx.x; i < min(sc.n_media, RT_SHADE_MEDIA_LDS) * RT_MEDIUM_F4; i += RT_BLOCK) s_media[i] = ld4(sc.media + i);
    __syncthreads();
    ShadeMedia msrc; msrc.global = sc.media; msrc.lds = (lds_f4*)s_media;
    RtGlobalTables tab; tab.sc = &sc;
    const uint32_t out_shard = (blockIdx.x + blockIdx.y) % RT_SHARDS;
    uint32_t* n_out_ptr = counts_out + out_shard * RT_COUNT_STRIDE;
    ia += (size_t)blockIdx.y * cap; ib += (size_t)blockIdx.y * cap; ic += (size_t)blockIdx.y * cap; hits += (size_t)blockIdx.y * cap;
    oa += (size_t)out_shard * cap; ob += (size_t)out_shard * cap; oc += (size_t)out_shard * cap;
    if (!(hits_out != nullptr && back_out != nullptr)) ls.enabled = 0;    // the next bounce's hit stream and back counters, for rays whose closest hit is settled here
    uint32_t* n_back_ptr = ls.enabled ? back_out + out_shard * RT_COUNT_STRIDE : nullptr;
    if (ls.enabled) hits_out += (size_t)out_shard * cap;

    // Returns true when the hit was postponed.  Called from wave-uniform control flow only (the append votes over the wave).
    auto shade = [&](uint32_t i, bool valid, bool may_postpone) -> bool {
        ShadeOut out; shade_out_clear(out);
        if (valid) shade_ray<STATS>(sc, tab, msrc, fr, bounce, ld4(ia + i), ld4(ib + i), bounce == 1 ? rt_first_c(blockIdx.y, i) : ld4(ic + i), ld4(hits + i), may_postpone, ls, contrib, work, out);
        shade_append(out, n_out_ptr, n_back_ptr, overflow, cap, oa, ob, oc, hits_out);
        return out.postponed;
    };

    // Hits on costly textures (Perlin turbulence: ~1300 instructions) are rare per wave but present in every other wave;
    // evaluated in place they would run with one or two live lanes.  Each wave instead collects their queue indices in
    // LDS and shades them together, a wave's worth at a time and whatever is left at the end.
    __shared__ uint32_t s_later[(RT_BLOCK / 64) * 128];
    uint32_t* later = s_later + (threadIdx.x / 64) * 128;
    const uint32_t lane = threadIdx.x & 63u;
    const bool may_postpone = sc.n_perlin > 0;
    uint32_t n_later = 0;                                              // wave-uniform
    for (uint32_t base = blockIdx.x * RT_BLOCK; base < n; base += gridDim.x * RT_BLOCK) {
        const uint32_t v = base + threadIdx.x;
        const uint32_t i = region_slot(min(v, n - 1u), n_front, cap);   // slot of entry v
        bool post = shade(i, v < n, may_postpone);
        unsigned long long pm = __ballot(post);
        if (pm) {
            uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
            if (post) later[n_later + rank] = i;
            n_later += (uint32_t)__popcll(pm);
            if (n_later >= 64u) {                                      // n_later < 128 always: at most 64 are added to < 64
                n_later -= 64u;
                __builtin_amdgcn_wave_barrier();
                uint32_t j = later[n_later + lane];
                shade(j, true, false);
            }
        }
    }
    if (n_later > 0) {
        __builtin_amdgcn_wave_barrier();
        uint32_t j = lane < n_later ? later[lane] : 0u;
        shade(j, lane < n_later, false);
    }
}

// LDS plan of one k_shade_lds block (host: plan_shade): offsets in 16-byte records from the start of the block's dynamic
// LDS.  Layout: [per wave: 4 x 64 staged ray records][per wave: 128 postponed indices][every table shading reads].
struct RtShadePlan {
    int32_t off_spheres, off_sphere_meta, off_mspheres, off_msphere_meta, off_rects, off_rect_meta, off_boxes, off_instances, off_ops,
            off_materials, off_textures, off_media, off_perlin_vec, off_perlin_perm, off_stage, off_later;
    int32_t off_nodes, off_stack, stack_cap;   // FUSED only: the whole node array and a traversal stack of 32-bit entries per lane
    // FUSED scenes of a handful of primitives: the leaves of the tree, listed; every ray then tests all of them in turn (the
    // linear scan of src/objects.rs:67-78) instead of walking the tree -- each lane runs the same tests, so no lane ever waits.
    int32_t n_flat; int32_t flat_leaf[RT_FLAT_LEAVES];
};
typedef int __attribute__((ext_vector_type(2))) rt_vi2;
typedef __attribute__((address_space(3))) const rt_vi2 lds_i2;
typedef __attribute__((address_space(3))) const unsigned char lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
struct ShadeTables {   // every table from the block's LDS copy
    static constexpr bool kWide = false;   // as a Nodes source (FUSED): binary nodes
    __device__ __forceinline__ int32_t root(const RtSceneView& sc) const { return sc.root; }
    __device__ __forceinline__ int32_t instance_root(int32_t, int32_t binary_root) const { return binary_root; }
    lds_f4* lds; RtShadePlan pl;
    __device__ __forceinline__ void pair(int32_t off, uint32_t i, rt_f4& r0, rt_f4& r1) const { lds_f4* m = lds + off + 2 * i; r0 = lds_ld4(m); r1 = lds_ld4(m + 1); }
    __device__ __forceinline__ rt_i2 meta(int32_t off, uint32_t i) const { rt_vi2 v = ((lds_i2*)(lds + off))[i]; rt_i2 r; r.x = v.x; r.y = v.y; return r; }
    __device__ __forceinline__ rt_f4 sphere(uint32_t i) const { return lds_ld4(lds + pl.off_spheres + i); }
    __device__ __forceinline__ void msphere(uint32_t i, rt_f4& m0, rt_f4& m1, rt_f4& m2) const {
        lds_f4* m = lds + pl.off_mspheres + 3 * i; m0 = lds_ld4(m); m1 = lds_ld4(m + 1); m2 = lds_ld4(m + 2);
    }
    __device__ __forceinline__ void rect(uint32_t i, rt_f4& r0, rt_f4& r1) const { pair(pl.off_rects, i, r0, r1); }
    __device__ __forceinline__ void box(uint32_t i, rt_f4& b0, rt_f4& b1) const { pair(pl.off_boxes, i, b0, b1); }
    __device__ __forceinline__ void instance(uint32_t i, rt_f4& i0, rt_f4& i1) const { pair(pl.off_instances, i, i0, i1); }
    __device__ __forceinline__ rt_i2 sphere_meta(uint32_t i) const { return meta(pl.off_sphere_meta, i); }
    __device__ __forceinline__ rt_i2 msphere_meta(uint32_t i) const { return meta(pl.off_msphere_meta, i); }
    __device__ __forceinline__ rt_i2 rect_meta(uint32_t i) const { return meta(pl.off_rect_meta, i); }
    __device__ __forceinline__ rt_f4 op(int i) const { return lds_ld4(lds + pl.off_ops + i); }
    __device__ __forceinline__ void material(int i, rt_f4& a0, rt_f4& a1) const { pair(pl.off_materials, (uint32_t)i, a0, a1); }
    __device__ __forceinline__ void texture(int i, rt_f4& t0, rt_f4& t1) const { pair(pl.off_textures, (uint32_t)i, t0, t1); }
    __device__ __forceinline__ rt_f4 medium(int m, int k) const { return lds_ld4(lds + pl.off_media + RT_MEDIUM_F4 * m + k); }
    __device__ __forceinline__ rt_f4 get(int m, int k) const { return medium(m, k); }   // as a MediaSrc
    __device__ __forceinline__ rt_f4 perlin_vec(int table, int h) const { return lds_ld4(lds + pl.off_perlin_vec + table * RT_PERLIN_VEC_F4 + h); }
    __device__ __forceinline__ int perlin_perm(int table, int i) const { return ((lds_u8*)(lds + pl.off_perlin_perm))[table * RT_PERLIN_PERM_BYTES + i]; }
    __device__ __forceinline__ void get(int32_t n, rt_f4 q[4]) const {   // as a Nodes source (FUSED)
        lds_f4* p = lds + pl.off_nodes + 4 * n;
        q[0] = lds_ld4(p); q[1] = lds_ld4(p + 1); q[2] = lds_ld4(p + 2); q[3] = lds_ld4(p + 3);
    }
};

// The ray and hit records of 64 consecutive queue entries, copied by the memory pipeline straight into the wave's staging
// area in LDS (global_load_lds_dwordx4: lane l's 16 bytes land at M0 + 16 l); no register is held while they travel.
// Written as inline assembly on purpose: the compiler makes every later LDS read wait for a copy it knows to be in
// flight (it cannot tell the tables from the staging area), which would stall the wave at its first table read.  The
// caller waits with vmcnt(0) before it reads the staging area.
__device__ __forceinline__ void shade_stage_one(const rt_f4* src, lds_f4* dst) {
    uint32_t m0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)dst);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0), "v"(src) : "memory");
}
template <bool NO_HITS>
__device__ __forceinline__ void shade_stage(const rt_f4* ia, const rt_f4* ib, const rt_f4* ic, const rt_f4* hits, uint32_t i, lds_f4* stage, bool with_c) {
    shade_stage_one(ia + i, stage);
    shade_stage_one(ib + i, stage + 64);
    if (with_c) shade_stage_one(ic + i, stage + 128);
    if (!NO_HITS) shade_stage_one(hits + i, stage + 192);
}

// The same stage for scenes whose shading tables fit one CU's LDS: one 1024-thread block per CU copies them once; the
// common path of a ray then has no dependent global-memory fetch at all, and the next 64 records of every wave travel
// into LDS while the current ones are shaded.  The wave waits for them at the top of the next iteration, which costs
// nothing when an append happened in between (it waited for all earlier memory operations already).
// FUSED (scenes of a handful of nodes): the closest hit is found here too, by the plain per-ray traversal over the node
// array in LDS, and no closest-hit kernel runs: the hit record never exists in memory and the ray record is read once.
// PROBE (FUSED only; the parity hook rt_trace_primary with RT_FLAG_WAVEFRONT_PRIMARY): the hit record found here is also written
// to `hits`, which a render never does.
template <bool FUSED, bool PROBE = false>
__global__ __launch_bounds__(1024) void k_shade_lds(RtSceneView sc, RtShadePlan pl, RtFrameDev fr, int bounce, uint32_t cap, const uint32_t* counts_in,
                                                    uint32_t* counts_out, uint32_t* overflow, const rt_f4* ia, const rt_f4* ib, const rt_f4* ic,
                                                    const rt_f4* hits, rt_f4* oa, rt_f4* ob, rt_f4* oc, rt_f4* contrib, rt_f4* hits_out,
                                                    const uint32_t* back_in, uint32_t* back_out, RtLocalSurfs ls) {
    const uint32_t n_front = min(counts_in[blockIdx.y * RT_COUNT_STRIDE], cap);
    const uint32_t n = n_front + region_back_count(back_in, blockIdx.y, n_front, cap, overflow);
    if (blockIdx.x * 1024u >= n) return;
    extern __shared__ __align__(16) unsigned char smem[];
    {   // tables -> LDS
        rt_f4* dst = reinterpret_cast<rt_f4*>(smem);
        const int t = threadIdx.x;
        for (int i = t; i < sc.n_spheres; i += 1024) dst[pl.off_spheres + i] = ld4(sc.spheres + i);
        for (int i = t; i < sc.n_mspheres * 3; i += 1024) dst[pl.off_mspheres + i] = ld4(sc.mspheres + i);
        for (int i = t; i < sc.n_rects * 2; i += 1024) dst[pl.off_rects + i] = ld4(sc.rects + i);
        for (int i = t; i < sc.n_boxes * 2; i += 1024) dst[pl.off_boxes + i] = ld4(sc.boxes + i);
        for (int i = t; i < sc.n_instances * 2; i += 1024) dst[pl.off_instances + i] = ld4(sc.instances + i);
        for (int i = t; i < sc.n_ops; i += 1024) dst[pl.off_ops + i] = ld4(sc.ops + i);
        for (int i = t; i < sc.n_materials * 2; i += 1024) dst[pl.off_materials + i] = ld4(sc.materials + i);
        for (int i = t; i < sc.n_textures * 2; i += 1024) dst[pl.off_textures + i] = ld4(sc.textures + i);
        for (int i = t; i < sc.n_media * RT_MEDIUM_F4; i += 1024) dst[pl.off_media + i] = ld4(sc.media + i);
        for (int i = t; i < sc.n_perlin * RT_PERLIN_VEC_F4; i += 1024) dst[pl.off_perlin_vec + i] = ld4(sc.perlin_vec + i);
        for (int i = t; i < sc.n_perlin * (RT_PERLIN_PERM_BYTES / 16); i += 1024) dst[pl.off_perlin_perm + i] = ld4(reinterpret_cast<const rt_f4*>(sc.perlin_perm) + i);
        if (FUSED) for (int i = t; i < sc.n_nodes * 4; i += 1024) dst[pl.off_nodes + i] = ld4(sc.nodes + i);
        rt_i2* m;
        m = reinterpret_cast<rt_i2*>(dst + pl.off_sphere_meta); for (int i = t; i < sc.n_spheres; i += 1024) m[i] = sc.sphere_meta[i];
        m = reinterpret_cast<rt_i2*>(dst + pl.off_msphere_meta); for (int i = t; i < sc.n_mspheres; i += 1024) m[i] = sc.msphere_meta[i];
        m = reinterpret_cast<rt_i2*>(dst + pl.off_rect_meta); for (int i = t; i < sc.n_rects; i += 1024) m[i] = sc.rect_meta[i];
    }
    __syncthreads();
    ShadeTables tab; tab.lds = (lds_f4*)smem; tab.pl = pl;
    const uint32_t out_shard = (blockIdx.x + blockIdx.y) % RT_SHARDS;
    uint32_t* n_out_ptr = counts_out + out_shard * RT_COUNT_STRIDE;
    ia += (size_t)blockIdx.y * cap; ib += (size_t)blockIdx.y * cap; ic += (size_t)blockIdx.y * cap; hits += (size_t)blockIdx.y * cap;
    oa += (size_t)out_shard * cap; ob += (size_t)out_shard * cap; oc += (size_t)out_shard * cap;
    if (FUSED || !(hits_out != nullptr && back_out != nullptr)) ls.enabled = 0;
    uint32_t* n_back_ptr = ls.enabled ? back_out + out_shard * RT_COUNT_STRIDE : nullptr;
    if (ls.enabled) hits_out += (size_t)out_shard * cap;

    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    lds_f4* stage = tab.lds + pl.off_stage + wave * 256u;
    lds_u32* later = (lds_u32*)(tab.lds + pl.off_later) + wave * (128u * RT_POSTPONE_CLASSES);
    const bool may_postpone = sc.n_perlin > 0 || RT_POSTPONE_CLASSES > 1;
    const uint32_t stride = gridDim.x * 1024u;
    uint32_t base = blockIdx.x * 1024u + wave * 64u;                   // wave-uniform
    TraceStack<true, false> stk;   // FUSED
    stk.base = (lds_int*)(tab.lds + pl.off_stack) + threadIdx.x; stk.depth = pl.stack_cap; stk.total = pl.stack_cap; stk.stride = 1024;
    stk.spill = nullptr; stk.spill_stride = 0;
    // the closest hit of ray (a, b) as the record k_trace would have written
    auto closest = [&](rt_f4 a, rt_f4 b) -> rt_f4 {
        RtHit hit; hit.t = rt_inf(); hit.code = RT_HIT_NONE; hit.inst = -1;
        if (pl.n_flat > 0) {
            RtTrav tv; rt_trav_ray(tv, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z)); tv.inst = -1; tv.sp = 0; tv.cur = RT_EMPTY_STACK;
            for (int k = 0; k < pl.n_flat; ++k) rt_leaf_test<false>(tab, pl.flat_leaf[k], a.w, RT_TMIN, (uint32_t)f2i(b.w), tv, hit, nullptr);
        } else
            rt_closest_hit_in<false>(sc, tab, tab, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), a.w, RT_TMIN, (uint32_t)f2i(b.w), stk, hit, nullptr);
        rt_f4 h; h.x = hit.t; h.y = i2f((int)hit.code); h.z = i2f(hit.inst); h.w = 0.0f;
        return h;
    };
    const bool with_c = bounce != 1;                                   // camera rays: see rt_first_c
    if (base < n) shade_stage<FUSED>(ia, ib, ic, hits, region_slot(min(base + lane, n - 1u), n_front, cap), stage, with_c);
    bool landed = false;                                               // the staged records are known to have arrived
    // One loop body, one inlined copy of the shading code: an iteration shades either 64 postponed hits of one class (as soon
    // as a class has that many, and whatever is left once the queue is exhausted) or the wave's next 64 staged queue entries.
    uint32_t n_later[RT_POSTPONE_CLASSES];                             // wave-uniform
#pragma unroll
    for (int k = 0; k < RT_POSTPONE_CLASSES; ++k) n_later[k] = 0;
    for (;;) {
        int flush = -1;
#pragma unroll
        for (int k = 0; k < RT_POSTPONE_CLASSES; ++k) if (n_later[k] >= 64u || (base >= n && n_later[k] > 0u)) flush = k;
        if (flush < 0 && base >= n) break;
        rt_f4 a, b, c, h;
        bool valid, may_post;
        uint32_t i;
        if (flush >= 0) {                                              // at most 127 entries in a list: 64 are added to < 64
            uint32_t have = 0;
#pragma unroll
            for (int k = 0; k < RT_POSTPONE_CLASSES; ++k) if (k == flush) { have = n_later[k]; n_later[k] = have >= 64u ? have - 64u : 0u; }
            const uint32_t first = have >= 64u ? have - 64u : 0u, cnt = have - first;
            __builtin_amdgcn_wave_barrier();
            valid = lane < cnt; may_post = false;
            i = valid ? later[(uint32_t)flush * 128u + first + lane] : 0u;
            if (valid) { a = ld4(ia + i); b = ld4(ib + i); if (with_c) c = ld4(ic + i); if (!FUSED) h = ld4(hits + i); }
            landed = true;                                             // these loads were issued after the staging and are waited for below
        } else {
            const uint32_t v = base + lane;                            // entry of the region's two parts taken as one range
            i = region_slot(min(v, n - 1u), n_front, cap);
            if (!landed) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            a = lds_ld4(stage + lane); b = lds_ld4(stage + 64 + lane); if (with_c) c = lds_ld4(stage + 128 + lane);
            if (!FUSED) h = lds_ld4(stage + 192 + lane);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the records are in registers, the staging area is free
            if (base + stride < n) shade_stage<FUSED>(ia, ib, ic, hits, region_slot(min(v + stride, n - 1u), n_front, cap), stage, with_c);
            valid = v < n; may_post = may_postpone;
            base += stride;
            landed = false;
        }
        if (!with_c) c = rt_first_c(blockIdx.y, i);
        ShadeOut out; shade_out_clear(out);
        if (valid) {
            if (FUSED) h = closest(a, b);
            if (FUSED && PROBE) *reinterpret_cast<float4*>(const_cast<rt_f4*>(hits) + i) = make_float4(h.x, h.y, h.z, h.w);
            shade_ray<false>(sc, tab, tab, fr, bounce, a, b, c, h, may_post, ls, contrib, nullptr, out);
        }
        if (shade_append(out, n_out_ptr, n_back_ptr, overflow, cap, oa, ob, oc, hits_out)) landed = true;
        if (may_post) {
#pragma unroll
            for (int k = 0; k < RT_POSTPONE_CLASSES; ++k) {
                unsigned long long pm = __ballot(out.postponed && out.cls == k);
                if (pm) {
                    uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                    if (out.postponed && out.cls == k) later[(uint32_t)k * 128u + n_later[k] + rank] = i;
                    n_later[k] += (uint32_t)__popcll(pm);
                }
            }
        }
    }
}

__global__ __launch_bounds__(RT_BLOCK) void k_add_into(float* dst, const float* src, size_t n) {   // rt_render_multi: handles that share a GPU
    size_t i = (size_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i < n) dst[i] += src[i];
}

// accum[pixel] += sum over the pass's samples, in sample order (main.rs:585-589), for every pixel of the pass's set.
__global__ __launch_bounds__(RT_BLOCK) void k_accumulate(const rt_f4* contrib, RtFrameDev fr, int spp_chunk, float* accum, float* accum_sq) {
    int k = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (k >= fr.n_set) return;
    float sx = 0, sy = 0, sz = 0, qx = 0, qy = 0, qz = 0;
    for (int s = 0; s < spp_chunk; ++s) {
        rt_f4 c = ld4(contrib + (size_t)s * fr.n_set + k);
        sx += c.x; sy += c.y; sz += c.z;
        qx = fmaf(c.x, c.x, qx); qy = fmaf(c.y, c.y, qy); qz = fmaf(c.z, c.z, qz);
    }
    const uint32_t pixel = rt_set_pixel(fr, (uint32_t)k);
    accum[3 * (size_t)pixel] += sx; accum[3 * (size_t)pixel + 1] += sy; accum[3 * (size_t)pixel + 2] += sz;
    if (accum_sq) { accum_sq[3 * (size_t)pixel] += qx; accum_sq[3 * (size_t)pixel + 1] += qy; accum_sq[3 * (size_t)pixel + 2] += qz; }
}

// folded[row] = sum over the regions of counts[row][region]: the rays that entered bounce `row` (row 0: the overflow counter),
// so that the host reads back 8 bytes per bounce instead of the counter lines.
__global__ __launch_bounds__(RT_BLOCK) void k_fold_counts(const uint32_t* counts, const uint32_t* back, int rows, unsigned long long* folded) {
    int row = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (row >= rows) return;
    unsigned long long sum = 0, settled = 0;
    for (int sh = 0; sh < RT_SHARDS; ++sh) {
        sum += counts[((size_t)row * RT_SHARDS + sh) * RT_COUNT_STRIDE];
        if (back && row > 0) settled += back[((size_t)row * RT_SHARDS + sh) * RT_COUNT_STRIDE];   // both parts of a region hold rays of the bounce
    }
    folded[row] = sum + settled;
    // one more word behind the rows (zeroed by the caller): the rays of all bounces that sat in the back parts (settled by the shading kernel)
    if (settled) atomicAdd(folded + rows, settled);
}

// Parity hook: closest hit of one sample per pixel.
__global__ __launch_bounds__(RT_BLOCK) void k_primary(RtSceneView sc, RtCameraDev cam, RtFrameDev fr, uint32_t sample, int32_t* ids,
                                                      float* ts, float* normals, int32_t* fronts, float* uvs) {
    __shared__ int s_stack[RT_STACK_DEPTH * RT_BLOCK];
    LdsStack stk; stk.base = s_stack + threadIdx.x; stk.depth = RT_STACK_DEPTH; stk.stride = RT_BLOCK;
    uint32_t pixel = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (pixel >= (uint32_t)fr.n_pixels) return;
    v3 o, d; float time;
    rt_camera_ray(cam, fr, pixel, sample, o, d, time);
    RtHit hit; hit.t = rt_inf(); hit.code = RT_HIT_NONE; hit.inst = -1;
    rt_closest_hit<false>(sc, o, d, time, RT_TMIN, RT_HIT_NONE, stk, hit, nullptr);
    if (sc.n_media > 0) {
        uint32_t rnd[4];
        rt_philox(pixel, sample, 1u, 0u, fr.key0, fr.key1, rnd);
        rt_media_hits(sc, o, d, time, RT_TMIN, pixel, sample, 1u, fr.key0, fr.key1, rnd, hit);
    }
    RtSurface sf; sf.id = -1; sf.n = mk3(0, 0, 0); sf.front = false; sf.u = sf.v = 0.0f;
    if (hit.code != RT_HIT_NONE) rt_surface(sc, o, d, time, hit, true, sf);
    ids[pixel] = sf.id; ts[pixel] = hit.t;
    if (normals) { normals[3 * pixel] = sf.n.x; normals[3 * pixel + 1] = sf.n.y; normals[3 * pixel + 2] = sf.n.z; }
    if (fronts) fronts[pixel] = sf.front ? 1 : 0;
    if (uvs) { uvs[2 * pixel] = sf.u; uvs[2 * pixel + 1] = sf.v; }
}

// Parity hook, second half of the wavefront form: the camera rays sit in the first bounce's queue (k_raygen) and their hit records
// are those the scene's closest-hit kernel wrote; media, hit record and the outputs as in k_primary.
__global__ __launch_bounds__(RT_BLOCK) void k_primary_finish(RtSceneView sc, RtFrameDev fr, uint32_t sample, uint32_t cap, const uint32_t* counts1,
                                                             const rt_f4* qa, const rt_f4* qb, const rt_f4* hits, int32_t* ids, float* ts, float* normals,
                                                             int32_t* fronts, float* uvs) {
    const uint32_t region = blockIdx.y;
    const uint32_t n = min(counts1[region * RT_COUNT_STRIDE], cap);
    for (uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x; i < n; i += gridDim.x * RT_BLOCK) {
        const size_t slot = (size_t)region * cap + i;
        const rt_f4 a = ld4(qa + slot), b = ld4(qb + slot), h = ld4(hits + slot);
        uint32_t pixel, s_local;
        rt_split_path(fr, (uint32_t)f2i(rt_first_c(region, i).w), pixel, s_local);
        const v3 o = mk3(a.x, a.y, a.z), d = mk3(b.x, b.y, b.z); const float time = a.w;
        RtHit hit; hit.t = h.x; hit.code = (uint32_t)f2i(h.y); hit.inst = f2i(h.z);
        if (sc.n_media > 0) {
            uint32_t rnd[4];
            rt_philox(pixel, sample, 1u, 0u, fr.key0, fr.key1, rnd);
            rt_media_hits(sc, o, d, time, RT_TMIN, pixel, sample, 1u, fr.key0, fr.key1, rnd, hit);
        }
        RtSurface sf; sf.id = -1; sf.n = mk3(0, 0, 0); sf.front = false; sf.u = sf.v = 0.0f;
        if (hit.code != RT_HIT_NONE) rt_surface(sc, o, d, time, hit, true, sf);
        ids[pixel] = sf.id; ts[pixel] = hit.t;
        if (normals) { normals[3 * pixel] = sf.n.x; normals[3 * pixel + 1] = sf.n.y; normals[3 * pixel + 2] = sf.n.z; }
        if (fronts) fronts[pixel] = sf.front ? 1 : 0;
        if (uvs) { uvs[2 * pixel] = sf.u; uvs[2 * pixel + 1] = sf.v; }
    }
}

// Parity hook: one scatter event per explicit ray.
__global__ __launch_bounds__(RT_BLOCK) void k_scatter_rays(RtSceneView sc, const float* rays, int n, uint32_t pixel0, uint32_t sample, int bounce,
                                                           uint32_t key0, uint32_t key1, float* out) {
    __shared__ int s_stack[RT_STACK_DEPTH * RT_BLOCK];
    LdsStack stk; stk.base = s_stack + threadIdx.x; stk.depth = RT_STACK_DEPTH; stk.stride = RT_BLOCK;
    int i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const float* r = rays + 7 * (size_t)i;
    v3 o = mk3(r[0], r[1], r[2]), d = mk3(r[3], r[4], r[5]); float time = r[6];
    uint32_t pixel = pixel0 + (uint32_t)i;
    RtHit hit; hit.t = rt_inf(); hit.code = RT_HIT_NONE; hit.inst = -1;
    rt_closest_hit<false>(sc, o, d, time, RT_TMIN, RT_HIT_NONE, stk, hit, nullptr);
    uint32_t rnd[4]; rt_philox(pixel, sample, (uint32_t)bounce, 0u, key0, key1, rnd);
    if (sc.n_media > 0) rt_media_hits(sc, o, d, time, RT_TMIN, pixel, sample, (uint32_t)bounce, key0, key1, rnd, hit);
    float* w = out + 14 * (size_t)i;
    for (int k = 0; k < 14; ++k) w[k] = 0.0f;
    if (hit.code == RT_HIT_NONE) { w[0] = -1.0f; return; }
    RtSurface sf; rt_surface(sc, o, d, time, hit, false, sf);
    v3 atten, dir, emitted;
    bool sc_ok = rt_scatter(sc, sf, d, rt_u24(rnd[0]), rt_u24(rnd[1]), atten, dir, emitted);
    w[0] = sc_ok ? 1.0f : 0.0f;
    w[1] = emitted.x; w[2] = emitted.y; w[3] = emitted.z; w[4] = atten.x; w[5] = atten.y; w[6] = atten.z;
    w[7] = sf.p.x; w[8] = sf.p.y; w[9] = sf.p.z; w[10] = dir.x; w[11] = dir.y; w[12] = dir.z; w[13] = time;
}

// ------------------------------------------------------------------ static scratch of rt_render_device_noalloc
// A fixed workspace for callers that cannot allocate device memory (HIP graph capture, serving loops): passes of at most
// RT_NOALLOC_MAX_PATHS paths and RT_NOALLOC_MAX_DEPTH bounces.  Every GPU the library runs on holds its own copy of the array
// until the process ends.  In host code the array's NAME is only a handle for hipGetSymbolAddress: the address that kernels
// and memsets take is the per-device pointer that call returns (RtScene::static_scratch).
#define RT_NOALLOC_MAX_PATHS (1u << 18)
#define RT_NOALLOC_MAX_DEPTH 126
static constexpr size_t kNaRows = RT_NOALLOC_MAX_DEPTH + 2;
static constexpr size_t kNaCap = ((RT_NOALLOC_MAX_PATHS + RT_SHARDS - 1) / RT_SHARDS + (size_t)1024 * (RT_SHARDS + 2) + RT_BLOCK - 1) / RT_BLOCK * RT_BLOCK;
static constexpr size_t kNaCountsBytes = 2 * kNaRows * RT_SHARDS * RT_COUNT_STRIDE * sizeof(uint32_t);
static constexpr size_t kNaFoldedBytes = ((kNaRows + 1) * sizeof(unsigned long long) + 255) / 256 * 256;
static constexpr size_t kNaContribBytes = (size_t)RT_NOALLOC_MAX_PATHS * sizeof(rt_f4);
static constexpr size_t kNaStreamBytes = kNaCap * RT_SHARDS * sizeof(rt_f4);     // one queue stream, or the hit stream
static constexpr size_t kNaScratchBytes = kNaCountsBytes + kNaFoldedBytes + kNaContribBytes + 7 * kNaStreamBytes;   // 64 MiB
static_assert(kNaCountsBytes % 256 == 0 && kNaContribBytes % 256 == 0 && kNaStreamBytes % 256 == 0, "pieces of the scratch array stay 256-byte aligned");
static __device__ __align__(256) unsigned char g_noalloc_scratch[kNaScratchBytes];

// ------------------------------------------------------------------ host side
// Device memory of one wavefront pass: the two ray queues (three float4 streams each), the hit stream, one contribution slot
// per path and the counters.  Every GPU has ONE grow-only workspace (hipMalloc) that all scene handles on it share -- a pass of
// 2^28 paths needs 38 GB, which no caller should pay once per scene -- and rt_render_device_noalloc carves a fixed one out of a
// static device array instead.
struct Workspace {
    size_t paths = 0, cap = 0;             // paths it was sized for; slots per queue region
    rt_f4 *qa[2] = {nullptr, nullptr}, *qb[2] = {nullptr, nullptr}, *qc[2] = {nullptr, nullptr}, *contrib = nullptr;
    // hit records of the rays entering bounce b: hits[b & 1].  Two streams because a shading launch reads this bounce's records while
    // it writes those of the next bounce's rays whose closest hit it settles itself (shade_ray); hits[1] == null: one stream, and no
    // ray is settled there.
    rt_f4* hits[2] = {nullptr, nullptr};
    // counts: `rows` rows of [region] queue sizes (row b = rays entering bounce b from the front of the regions; word 0 of row 0,
    // which no bounce uses, is the overflow counter), then `rows` rows of k_trace cursors, then `rows` rows of the sizes of the
    // regions' back parts (workspaces with two hit streams only).  folded: per row, the sum over the regions (k_fold_counts).
    uint32_t* counts = nullptr;
    unsigned long long* folded = nullptr;
    int rows = 0;
    unsigned long long* work = nullptr;    // 32 work counters of the instrumented kernels; null: no instrumented runs on this workspace
};
struct RtScene {
    int device = 0, n_cu = 0;
    rt::FlatScene flat;            // host copy (info only after upload)
    RtSceneView view;              // device pointers
    std::vector<void*> allocs;     // scene arrays
    uint64_t device_bytes = 0;
    unsigned char* static_scratch = nullptr;   // this device's copy of the static scratch array (rt_render_device_noalloc)
    RtTracePlan plan;                      // k_trace: what lives in the block's LDS
    int lds_bytes = 0;                     // k_trace: dynamic LDS size
    unsigned trace_gx = 1;                 // k_trace: blocks per region
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> ev_trace;      // per-bounce event pairs around k_trace (RT_FLAG_TIME_TRACE)
    const rt_f4* nodes_on_device = nullptr; // set when the top level was built on the GPU (the node array is already there)
    double bvh_build_seconds = 0.0;        // device time of that build
    int top_first = 0;                     // first node of the region k_trace should keep in LDS (see plan_trace)
    int trace_block = 1024;                // threads per k_trace block (1024, or 256 where a block cannot have 160 KB of LDS)
    RtLocalSurfs local_surfs;              // ... and the wholly-inside surfaces of those lists (kernel argument of the shading kernels)
    bool has_local_media = false;          // some medium record lists the surfaces inside its boundary (flatten.cpp, emit_medium_locals)
    void* stack_spill = nullptr;           // k_trace: global part of the traversal stacks (entries beyond plan.stack_cap), [entry][thread of the grid]
    bool shade_lds = false;                // the shading tables fit one CU's LDS: k_shade_lds instead of k_shade
    bool shade_fused = false;              // ... and the BVH is a handful of nodes: k_shade_lds<true> finds the closest hits too, k_trace is not run
    RtShadePlan shade_plan;                // k_shade_lds: layout of the block's LDS
    int shade_lds_bytes = 0;
    const int32_t* d_wsrc = nullptr;       // device copy of flat.wsrc (refit of the wide tree)
    std::vector<int32_t> of_id[2]; bool indexed = false;   // object id -> sphere / moving-sphere record (rt_scene_set_*)
    double refit_seconds = 0.0;
    // rt_render_multi: this GPU's private accumulators and stream (grow-only)
    float *multi_acc = nullptr, *multi_sq = nullptr; size_t multi_floats = 0; hipStream_t multi_stream = nullptr;
    bool counted = false;                  // this handle is in its device's DeviceWorkspace::handles
};

static const int kLdsBudget = 160 * 1024 - 512;   // dynamic LDS a 1024-thread block may ask for (one block per CU)

// k_trace comes in two block shapes: 1024 threads (one block per CU; its 16 waves share one LDS copy of the scene) and,
// where a workgroup cannot have that much LDS, 256 threads (four blocks per CU, each with a 40 KB plan).
static void launch_trace(RtScene* s, const Workspace& ws, bool stats, hipStream_t st, uint32_t cap, const uint32_t* cin, uint32_t* cursors,
                         const rt_f4* qa, const rt_f4* qb, rt_f4* hits) {
    dim3 grid(s->trace_gx, RT_SHARDS);
    const bool spill = s->plan.stack_total > s->plan.stack_cap;   // only the two forms below are built with a spilling stack (plan_trace)
#define RT_LAUNCH_TRACE(TB_, WIDE_, RES_, W4_, SP_)                                                                           \
    do {                                                                                                                      \
        if (stats) hipLaunchKernelGGL((k_trace<true, TB_, WIDE_, RES_, W4_, SP_>), grid, dim3(TB_), (size_t)s->lds_bytes, st, s->view, s->plan, cap, \
                                      cin, cursors, qa, qb, hits, ws.work, s->stack_spill);                                   \
        else hipLaunchKernelGGL((k_trace<false, TB_, WIDE_, RES_, W4_, SP_>), grid, dim3(TB_), (size_t)s->lds_bytes, st, s->view, s->plan, cap,      \
                                cin, cursors, qa, qb, hits, ws.work, s->stack_spill);                                         \
    } while (0)
    // (a spilling stack first: with fewer entries in LDS a wide tree may have become resident, and only the general form spills)
    if (s->plan.wide4 && spill) RT_LAUNCH_TRACE(1024, true, false, true, true);
    else if (s->plan.wide4 && s->plan.resident) RT_LAUNCH_TRACE(1024, true, true, true, false);
    else if (s->plan.wide4) RT_LAUNCH_TRACE(1024, true, false, true, false);
    else if (s->trace_block == 1024 && s->plan.resident && spill) RT_LAUNCH_TRACE(1024, false, true, false, true);
    else if (s->trace_block == 1024 && s->plan.resident) RT_LAUNCH_TRACE(1024, false, true, false, false);
    else if (s->trace_block == 1024 && !s->plan.wide_stack) RT_LAUNCH_TRACE(1024, false, false, false, false);
    else if (s->trace_block == 1024) RT_LAUNCH_TRACE(1024, true, false, false, false);
    else RT_LAUNCH_TRACE(256, true, false, false, false);
#undef RT_LAUNCH_TRACE
}

// Fills the LDS of a k_trace block of `threads` threads within `budget` bytes.  The stack always has the full depth the
// BVH can need.  If the whole node array fits beside a stack of 16-bit entries (and node indices fit 15 bits), that form
// is used and the room left goes to whole primitive arrays; otherwise 32-bit entries and a breadth-first node prefix.
// wide4: the plan is for the 4-wide tree (n_nodes of sc.wnodes): 32-bit stack entries always, and when all nodes fit, primitive
// arrays beside them as in the narrow form.
// lds_entries > 0: only that many stack entries per lane are kept in LDS, deeper ones spill to global memory.
static void fill_plan(const RtSceneView& v, int n_nodes, int max_stack, int lds_entries, int threads, int budget, int refill_lanes, bool allow_narrow, bool wide4,
                      int top_first, RtTracePlan& pl, int& lds_bytes) {
    pl.stack_total = std::max(1, max_stack);
    const int stack_cap = lds_entries > 0 ? std::min(pl.stack_total, lds_entries) : pl.stack_total;
    pl.stack_cap = stack_cap; pl.refill_lanes = refill_lanes; pl.resident = 0;
 pl.first_lds = 0; pl.wide4 = wide4 ? 1 : 0;
    pl.off_spheres = pl.off_mspheres = pl.off_rects = pl.off_boxes = pl.off_instances = -1;
    const int narrow = stack_cap * threads * 2, wide = stack_cap * threads * 4;
    const int stack_bytes = wide4 ? wide : narrow;
    int used;
    if (n_nodes > 0 && (wide4 || (allow_narrow && n_nodes <= 32767)) && n_nodes * 64 + stack_bytes <= budget) {
        pl.wide_stack = wide4 ? 1 : 0; pl.n_nodes_lds = n_nodes;
        used = n_nodes * 64;
        int room = budget - used - stack_bytes;
        auto place = [&](int n_records, int32_t& off) {
            if (n_records > 0 && n_records * 16 <= room) { off = used / 16; used += n_records * 16; room -= n_records * 16; }
        };
        place(v.n_spheres, pl.off_spheres);
        place(v.n_boxes * 2, pl.off_boxes);
        place(v.n_instances * 2, pl.off_instances);
        place(v.n_rects * 2, pl.off_rects);
        place(v.n_mspheres * 3, pl.off_mspheres);
        pl.resident = (v.n_spheres == 0 || pl.off_spheres >= 0) && (v.n_boxes == 0 || pl.off_boxes >= 0) &&
                      (v.n_instances == 0 || pl.off_instances >= 0) && (v.n_rects == 0 || pl.off_rects >= 0) &&
                      (v.n_mspheres == 0 || pl.off_mspheres >= 0);
        pl.off_stack = used / 16;
        lds_bytes = used + stack_bytes;
    } else {
        pl.wide_stack = 1;
        pl.first_lds = top_first;
        pl.n_nodes_lds = std::max(0, std::min(n_nodes - top_first, (budget - wide) / 64));
        used = pl.n_nodes_lds * 64;
        pl.off_stack = used / 16;
        lds_bytes = used + wide;
    }
}

// Chooses k_trace's block shape, LDS plan and grid for a scene.
static int plan_trace(RtScene* s) {
    // idle lanes that trigger a refill, measured: 24 beats 16 by 1.3 % (final_scene.yml) to 3.6 % (cornell_box.yml) when the scene is in
    // LDS; the 4-wide kernel's longer steps prefer fuller waves (16)
    int refill = 24, refill_wide = 16;
    if (const char* e = std::getenv("RT_REFILL_LANES")) refill = refill_wide = std::max(1, std::min(64, std::atoi(e)));
    const char* env_tb = std::getenv("RT_TRACE_BLOCK");
    const int want = env_tb ? std::atoi(env_tb) : 1024;
    const char* env_w = std::getenv("RT_TRACE_WIDE_STACK"); const bool narrow_ok = !(env_w && std::atoi(env_w) != 0);
    int blocks_per_cu = 1;
    bool planned = false;
    // RT_TRACE_STACK_LDS = n: only n stack entries per lane are kept in LDS, deeper ones spill to global memory (for a scene
    // resident in LDS, and for the 4-wide tree of one that is not).  Measured: the worst case is far above the depth in use (23
    // against 15 reached on final_scene.yml, 0.07 % of the node steps with more than 8 entries pending; 36 against 19 on the 1M-sphere
    // scene), but the LDS it frees buys little -- 12 entries in LDS let 1700 instead of 200 top nodes of the 1M-sphere scene
    // stay there for +0.5 % -- and the spill path costs the node loop a branch, so the default keeps every entry in LDS.
    int stack_lds = 0;
    if (const char* e = std::getenv("RT_TRACE_STACK_LDS")) stack_lds = std::max(0, std::atoi(e));
    if (want == 1024 && s->flat.max_stack * 1024 * 4 + 64 <= kLdsBudget) {
        fill_plan(s->view, s->view.n_nodes, s->flat.max_stack, 0, 1024, kLdsBudget, refill, narrow_ok, false, s->top_first, s->plan, s->lds_bytes);
        if (stack_lds > 0 && s->plan.resident)
            fill_plan(s->view, s->view.n_nodes, s->flat.max_stack, stack_lds, 1024, kLdsBudget, refill, narrow_ok, false, s->top_first, s->plan, s->lds_bytes);
        // The 4-wide quantised tree replaces the binary one when the binary node array does not fit the LDS (a ray then makes
        // half as many dependent fetches from L2 / HBM), provided its deeper stack fits.  RT_TRACE_BVH4 = 1 / 0 forces the choice.
        bool want_wide = !s->plan.resident && s->plan.wide_stack;
        if (const char* e = std::getenv("RT_TRACE_BVH4")) want_wide = std::atoi(e) != 0;
        if (want_wide && s->view.n_wnodes > 0 && s->view.wmax_stack * 1024 * 4 + 64 <= kLdsBudget) {
            fill_plan(s->view, s->view.n_wnodes, s->view.wmax_stack, 0, 1024, kLdsBudget, refill_wide, false, true, 0, s->plan, s->lds_bytes);
            if (stack_lds > 0 && !s->plan.resident)
                fill_plan(s->view, s->view.n_wnodes, s->view.wmax_stack, stack_lds, 1024, kLdsBudget, refill_wide, false, true, 0, s->plan, s->lds_bytes);
        }
        const bool spill = s->plan.stack_total > s->plan.stack_cap;
        const void* fns[2];
        if (s->plan.wide4 && spill) { fns[0] = reinterpret_cast<const void*>(&k_trace<false, 1024, true, false, true, true>); fns[1] = reinterpret_cast<const void*>(&k_trace<true, 1024, true, false, true, true>); }
        else if (s->plan.wide4 && s->plan.resident) { fns[0] = reinterpret_cast<const void*>(&k_trace<false, 1024, true, true, true, false>); fns[1] = reinterpret_cast<const void*>(&k_trace<true, 1024, true, true, true, false>); }
        else if (s->plan.wide4) { fns[0] = reinterpret_cast<const void*>(&k_trace<false, 1024, true, false, true, false>); fns[1] = reinterpret_cast<const void*>(&k_trace<true, 1024, true, false, true, false>); }
        else if (s->plan.resident && spill) { fns[0] = reinterpret_cast<const void*>(&k_trace<false, 1024, false, true, false, true>); fns[1] = reinterpret_cast<const void*>(&k_trace<true, 1024, false, true, false, true>); }
        else if (s->plan.resident) { fns[0] = reinterpret_cast<const void*>(&k_trace<false, 1024, false, true, false, false>); fns[1] = reinterpret_cast<const void*>(&k_trace<true, 1024, false, true, false, false>); }
        else if (!s->plan.wide_stack) { fns[0] = reinterpret_cast<const void*>(&k_trace<false, 1024, false, false, false, false>); fns[1] = reinterpret_cast<const void*>(&k_trace<true, 1024, false, false, false, false>); }
        else { fns[0] = reinterpret_cast<const void*>(&k_trace<false, 1024, true, false, false, false>); fns[1] = reinterpret_cast<const void*>(&k_trace<true, 1024, true, false, false, false>); }
        // the limit is a property of the kernel, not of this scene: always the whole budget, so that a scene created
        // later with a smaller plan does not lower it for this one
        hipError_t e = hipFuncSetAttribute(fns[0], hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
        if (e == hipSuccess) e = hipFuncSetAttribute(fns[1], hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget);
        if (e == hipSuccess) { s->trace_block = 1024; planned = true; }
        else (void)hipGetLastError();   // this device does not give one workgroup that much LDS: use the small blocks
    }
    if (!planned) {
        fill_plan(s->view, s->view.n_nodes, s->flat.max_stack, 0, 256, 40 * 1024 - 128, refill, false, false, s->top_first, s->plan, s->lds_bytes);
        s->trace_block = 256; blocks_per_cu = 4;
        if (s->lds_bytes > 48 * 1024) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trace<false, 256, true, false, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, s->lds_bytes));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trace<true, 256, true, false, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, s->lds_bytes));
        }
    }
    s->trace_gx = (unsigned)std::max(1, (std::max(1, s->n_cu) * blocks_per_cu + RT_SHARDS - 1) / RT_SHARDS);
    if (s->plan.stack_total > s->plan.stack_cap) {   // the global part of the stacks: [entry][thread of the grid]
        const size_t gthreads = (size_t)s->trace_gx * RT_SHARDS * (size_t)s->trace_block;
        const size_t bytes = (size_t)(s->plan.stack_total - s->plan.stack_cap) * gthreads * (s->plan.wide_stack ? 4 : 2);
        HIP_TRY(hipMalloc(&s->stack_spill, bytes));
        s->allocs.push_back(s->stack_spill); s->device_bytes += bytes;
    }
    return RT_OK;
}

// Decides whether the scene's shading tables, 4 KB of ray staging per wave and the lists of postponed hits fit the LDS of
// one CU; then k_shade_lds replaces k_shade.
static int plan_shade(RtScene* s) {
    s->shade_lds = false; s->shade_fused = false;
    if (const char* e = std::getenv("RT_SHADE_LDS")) { if (std::atoi(e) == 0) return RT_OK; }
    const RtSceneView& v = s->view;
    RtShadePlan& pl = s->shade_plan;
    long used = 0;   // 16-byte records
    auto place = [&](long n_records, int32_t& off) { off = (int32_t)used; used += n_records; };
    place(16 * 4 * 64, pl.off_stage);   // first: its addresses then fit 16 bits, whatever width the copy instruction takes from M0
    place(16 * 128 * RT_POSTPONE_CLASSES / 4, pl.off_later);
    place(v.n_spheres, pl.off_spheres);                     place((v.n_spheres + 1) / 2, pl.off_sphere_meta);
    place((long)v.n_mspheres * RT_MSPHERE_F4, pl.off_mspheres); place((v.n_mspheres + 1) / 2, pl.off_msphere_meta);
    place((long)v.n_rects * RT_RECT_F4, pl.off_rects);      place((v.n_rects + 1) / 2, pl.off_rect_meta);
    place((long)v.n_boxes * 2, pl.off_boxes);
    place((long)v.n_instances * RT_INSTANCE_F4, pl.off_instances);
    place(v.n_ops, pl.off_ops);
    place((long)v.n_materials * RT_MATERIAL_F4, pl.off_materials);
    place((long)v.n_textures * RT_TEXTURE_F4, pl.off_textures);
    place((long)v.n_media * RT_MEDIUM_F4, pl.off_media);
    place((long)v.n_perlin * RT_PERLIN_VEC_F4, pl.off_perlin_vec);
    place((long)v.n_perlin * (RT_PERLIN_PERM_BYTES / 16), pl.off_perlin_perm);
    if (used * 16 > kLdsBudget || s->n_cu < RT_SHARDS) return RT_OK;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_shade_lds<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_shade_lds<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_shade_lds<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBudget) != hipSuccess) {
        (void)hipGetLastError();   // this device does not give one workgroup that much LDS
        return RT_OK;
    }
    s->shade_lds = true;
    // A scene of a few nodes and no instances: its traversal is a few steps per ray, cheaper than a second kernel and a hit
    // record per ray (cornell.yml +22 %, orig_img.yml +23 %; with instanced boxes in the tree it measured 6 % slower).
    pl.off_nodes = pl.off_stack = 0; pl.stack_cap = 0;
    const char* ef = std::getenv("RT_SHADE_FUSED");
    if (!(ef && std::atoi(ef) == 0) && v.n_nodes >= 1 && v.n_nodes <= 16 && s->flat.max_stack <= 8 && s->flat.blas_depth == 0) {
        long with = used;
        int32_t on = (int32_t)with; with += (long)v.n_nodes * 4;
        int32_t os = (int32_t)with; with += (long)std::max(1, s->flat.max_stack) * 1024 / 4;
        if (with * 16 <= kLdsBudget) { pl.off_nodes = on; pl.off_stack = os; pl.stack_cap = std::max(1, s->flat.max_stack); used = with; s->shade_fused = true; }
    }
    // the leaves of such a tree, for the linear scan (RT_SHADE_FLAT = 0 keeps the tree walk)
    pl.n_flat = 0;
    const char* efl = std::getenv("RT_SHADE_FLAT");
    if (s->shade_fused && !(efl && std::atoi(efl) == 0)) {
        std::vector<int32_t> leaves; int prims = 0; bool ok = true;
        std::vector<int> todo(1, v.root);
        const std::vector<rt_f4>& nd = s->flat.nodes;
        while (!todo.empty() && ok) {
            int n = todo.back(); todo.pop_back();
            int32_t kids[2]; std::memcpy(&kids[0], &nd[(size_t)n * 4 + 3].x, 4); std::memcpy(&kids[1], &nd[(size_t)n * 4 + 3].y, 4);
            const bool empty[2] = {!(nd[(size_t)n * 4].x <= nd[(size_t)n * 4].w), !(nd[(size_t)n * 4 + 1].z <= nd[(size_t)n * 4 + 2].y)};   // min.x > max.x: absent child
            for (int c = 1; c >= 0; --c) {
                if (empty[c] || (c == 1 && kids[1] == kids[0] && kids[0] < 0)) continue;
                if (kids[c] >= 0) { todo.push_back(kids[c]); continue; }
                uint32_t x = ~(uint32_t)kids[c];
                if (x & RT_LEAF_INSTANCE_BIT) { ok = false; break; }
                leaves.push_back(kids[c]); prims += (int)((x >> 26) & 3u) + 1;
            }
        }
        if (ok && !leaves.empty() && (int)leaves.size() <= RT_FLAT_LEAVES && prims <= RT_FLAT_PRIMS) {
            pl.n_flat = (int32_t)leaves.size();
            for (size_t k = 0; k < leaves.size(); ++k) pl.flat_leaf[k] = leaves[k];
        }
    }
    s->shade_lds_bytes = (int)(used * 16);
    return RT_OK;
}

template <class T>
static int upload(RtScene* s, const std::vector<T>& v, const T** out) {
    *out = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);  // never a null array: kernels may form the base address
    void* d = nullptr;
    HIP_TRY(hipMalloc(&d, bytes));
    s->allocs.push_back(d);
    s->device_bytes += bytes;
    if (!v.empty()) HIP_TRY(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    else HIP_TRY(hipMemset(d, 0, bytes));
    *out = (const T*)d;
    return RT_OK;
}

static void free_queues(Workspace& w) {
    for (int i = 0; i < 2; ++i) { if (w.qa[i]) (void)hipFree(w.qa[i]); if (w.qb[i]) (void)hipFree(w.qb[i]); if (w.qc[i]) (void)hipFree(w.qc[i]); w.qa[i] = w.qb[i] = w.qc[i] = nullptr; }
    if (w.contrib) (void)hipFree(w.contrib);
    for (int i = 0; i < 2; ++i) { if (w.hits[i]) (void)hipFree(w.hits[i]); w.hits[i] = nullptr; }
    w.contrib = nullptr; w.paths = 0; w.cap = 0;
}
// Slots per queue region for a pass of n_paths paths (see the capacity argument at the top of the file).
static size_t region_cap(size_t n_paths) {
    size_t cap = (n_paths + RT_SHARDS - 1) / RT_SHARDS + (size_t)1024 * (RT_SHARDS + 2);
    return (cap + RT_BLOCK - 1) / RT_BLOCK * RT_BLOCK;
}
// The shared workspace of one GPU.  A render call holds `busy` from its first launch until its stream has drained, so two calls on
// one GPU (two scenes, two host threads) take turns instead of sharing queues; calls on different GPUs never meet.
struct DeviceWorkspace { std::mutex busy; Workspace ws; int handles = 0; };   // handles: scene handles alive on the device
static std::mutex g_pool_mutex;
static std::map<int, DeviceWorkspace*> g_pool;      // by device; entries live until rt_release_workspace or the end of the process
static DeviceWorkspace* device_workspace(int device) {
    std::lock_guard<std::mutex> lock(g_pool_mutex);
    DeviceWorkspace*& d = g_pool[device];
    if (!d) d = new DeviceWorkspace;
    return d;
}
static int ensure_workspace(RtScene* s, Workspace& w, size_t n_paths) {
    if (!w.counts) {
        w.rows = RT_MAX_BOUNCES + 2;
        HIP_TRY(hipMalloc((void**)&w.counts, (size_t)3 * w.rows * RT_SHARDS * RT_COUNT_STRIDE * sizeof(uint32_t)));
        HIP_TRY(hipMalloc((void**)&w.folded, (size_t)(w.rows + 1) * sizeof(unsigned long long)));
    }
    if (!w.work) { HIP_TRY(hipMalloc((void**)&w.work, 32 * sizeof(unsigned long long))); }
    if (!s->ev0) { HIP_TRY(hipEventCreate(&s->ev0)); HIP_TRY(hipEventCreate(&s->ev1)); }
    size_t cap = region_cap(n_paths);
    // test hook: a deliberately small region capacity makes the queues overflow, which must be reported, not rendered
    bool capped = false;
    if (const char* e = std::getenv("RT_DEBUG_QUEUE_CAP")) { long v = std::atol(e); if (v > 0) { cap = std::min(cap, ((size_t)v + RT_BLOCK - 1) / RT_BLOCK * RT_BLOCK); capped = true; } }
    if (w.paths >= n_paths && !capped && w.cap >= cap) return RT_OK;
    if (w.paths >= n_paths && capped) { w.cap = std::min(w.cap, cap); return RT_OK; }   // the hook only ever shrinks what the kernels may use
    free_queues(w);
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipMalloc((void**)&w.qa[i], cap * RT_SHARDS * sizeof(rt_f4)));
        HIP_TRY(hipMalloc((void**)&w.qb[i], cap * RT_SHARDS * sizeof(rt_f4)));
        HIP_TRY(hipMalloc((void**)&w.qc[i], cap * RT_SHARDS * sizeof(rt_f4)));
    }
    HIP_TRY(hipMalloc((void**)&w.contrib, n_paths * sizeof(rt_f4)));
    for (int i = 0; i < 2; ++i) HIP_TRY(hipMalloc((void**)&w.hits[i], cap * RT_SHARDS * sizeof(rt_f4)));
    w.paths = n_paths; w.cap = cap;
    return RT_OK;
}
// The fixed workspace inside this device's copy of the static scratch array.
static Workspace static_workspace(unsigned char* base) {
    Workspace w;
    unsigned char* p = base;
    auto take = [&](size_t bytes) { unsigned char* r = p; p += bytes; return r; };
    w.counts = (uint32_t*)take(kNaCountsBytes);
    w.folded = (unsigned long long*)take(kNaFoldedBytes);
    w.contrib = (rt_f4*)take(kNaContribBytes);
    for (int i = 0; i < 2; ++i) { w.qa[i] = (rt_f4*)take(kNaStreamBytes); w.qb[i] = (rt_f4*)take(kNaStreamBytes); w.qc[i] = (rt_f4*)take(kNaStreamBytes); }
    w.hits[0] = (rt_f4*)take(kNaStreamBytes); w.hits[1] = nullptr;
    w.rows = (int)kNaRows; w.paths = RT_NOALLOC_MAX_PATHS; w.cap = kNaCap; w.work = nullptr;
    return w;
}

static int check_common(const RtScene* s, const RtCamera* cam, const RtRenderParams* p) {
    if (!s || !cam || !p) return rt_set_error(RT_ERR_INVALID, "null argument");
    if (p->width < 2 || p->height < 2) return rt_set_error(RT_ERR_INVALID, "width and height must be >= 2 (main.rs:572-573 divides by W-1, H-1)");
    if ((uint64_t)p->width * (uint64_t)p->height > 0x7FFFFFFFull / 4) return rt_set_error(RT_ERR_INVALID, "image too large");
    if (p->max_depth < 0 || p->max_depth > RT_MAX_BOUNCES) return rt_set_error(RT_ERR_INVALID, "max_depth out of range");
    if (s->flat.max_stack > RT_STACK_DEPTH) return rt_set_error(RT_ERR_UNSUPPORTED, "BVH deeper than the traversal stack");
    if (const char* why = rt_frame_check(p)) return rt_set_error(RT_ERR_INVALID, why);
    return RT_OK;
}

static void make_frame(const RtRenderParams* p, RtFrameDev& fr) { rt_make_frame(p, fr); }
// Pixels a call with these parameters renders (the whole image, or its tile / stripe selection).
static size_t set_pixels(const RtRenderParams* p) { RtFrameDev fr; rt_make_frame(p, fr); return (size_t)fr.n_set; }
static void make_cam(const RtCamera* cam, RtCameraDev& cd) {
    rt::CameraDesc d;
    d.look_from = rt::vec3(cam->look_from[0], cam->look_from[1], cam->look_from[2]);
    d.look_at = rt::vec3(cam->look_at[0], cam->look_at[1], cam->look_at[2]);
    d.v_up = rt::vec3(cam->v_up[0], cam->v_up[1], cam->v_up[2]);
    d.vertical_fov_deg = cam->vertical_fov_deg; d.aspect_ratio = cam->aspect_ratio; d.aperture = cam->aperture;
    d.focus_dist = cam->focus_dist; d.shutter_open = cam->shutter_open; d.shutter_close = cam->shutter_close;
    rt::make_camera(d, cd);
}

extern "C" {

int rt_device_count(int* count) {
    if (!count) return rt_set_error(RT_ERR_INVALID, "null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return rt_set_error(RT_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *count = n;
    return RT_OK;
}

static const int kTopCopies = 2048;   // nodes of a device-built tree copied to a contiguous breadth-first run (<= what LDS can hold)

int rt_scene_create_ex(const RtBuilder* b, RtObj world, int device, int flags, RtScene** out) {
    if (!b || !out) return rt_set_error(RT_ERR_INVALID, "null argument");
    *out = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return rt_set_error(RT_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= n_dev) return rt_set_error(RT_ERR_INVALID, "device index out of range");
    RtScene* s = new RtScene;
    s->device = device;
    {
        std::string err;
        if (!rt::flatten(b->b, world, s->flat, err, (flags & RT_BVH_LINEAR) != 0)) {
            delete s;
            return rt_set_error(rt_status_of(err), err);
        }
    }
    int rc = [&]() -> int {
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipDeviceGetAttribute(&s->n_cu, hipDeviceAttributeMultiprocessorCount, device));
        // this device's copy of the static scratch array (the device is current): the address kernels and memsets take
        HIP_TRY(hipGetSymbolAddress((void**)&s->static_scratch, HIP_SYMBOL(g_noalloc_scratch)));
        rt::FlatScene& f = s->flat;
        if (!f.tlas_items.empty()) {
            // Deferred top level: its nodes are built on the GPU behind the bottom-level nodes, then copied back so that the
            // host copy describes the whole tree.
            const int n = (int)f.tlas_items.size() / 2, base = (int)f.nodes.size() / 4;
            const rt_f4* d_items = nullptr;
            rt_f4* d_nodes = nullptr;
            int r;
            if ((r = upload(s, f.tlas_items, &d_items))) return r;
            size_t bytes = ((size_t)base + (size_t)n - 1) * 4 * sizeof(rt_f4);
            HIP_TRY(hipMalloc((void**)&d_nodes, bytes));
            s->allocs.push_back(d_nodes); s->device_bytes += bytes;
            if (base > 0) HIP_TRY(hipMemcpy(d_nodes, f.nodes.data(), (size_t)base * 4 * sizeof(rt_f4), hipMemcpyHostToDevice));
            rt::LbvhFrameHost fr = rt::lbvh_frame(f);
            int depth = 0; std::string err;
            if (!rt::lbvh_build_device(fr.lo, fr.inv_ext, d_items, n, d_nodes + 4 * (size_t)base, base, &depth, &s->bvh_build_seconds, err))
                return rt_set_error(RT_ERR_HIP, err);
            f.nodes.resize(((size_t)base + (size_t)n - 1) * 4);
            HIP_TRY(hipMemcpy(f.nodes.data(), d_nodes, bytes, hipMemcpyDeviceToHost));
            f.root = base; f.tlas_depth = depth; f.max_stack += depth;
            if (f.max_stack > RT_STACK_DEPTH) return rt_set_error(RT_ERR_UNSUPPORTED, "linear BVH deeper than the traversal stack");
            // The radix tree's nodes are numbered by sorted position, not by level, so the top of the tree is scattered over
            // the array.  k_trace keeps a contiguous run of nodes in LDS: copies of the top levels, in breadth-first order and
            // with their links redirected to each other, are appended and become that run (and the root).
            const int total = base + n - 1;
            if (total > kTopCopies) {
                std::vector<int> order(1, f.root);
                for (size_t head = 0; head < order.size() && (int)order.size() < kTopCopies; ++head)
                    for (int c = 0; c < 2 && (int)order.size() < kTopCopies; ++c) {
                        int32_t child; std::memcpy(&child, c == 0 ? &f.nodes[(size_t)order[head] * 4 + 3].x : &f.nodes[(size_t)order[head] * 4 + 3].y, 4);
                        if (child >= 0) order.push_back(child);
                    }
                std::vector<rt_f4> top(order.size() * 4);
                // order[k] gets copy index total + k; a child that is itself copied is order[j] for the j at which it was pushed
                size_t next = 1;
                for (size_t k = 0; k < order.size(); ++k) {
                    for (int q = 0; q < 4; ++q) top[k * 4 + q] = f.nodes[(size_t)order[k] * 4 + q];
                    for (int c = 0; c < 2; ++c) {
                        float& w = c == 0 ? top[k * 4 + 3].x : top[k * 4 + 3].y;
                        int32_t child; std::memcpy(&child, &w, 4);
                        if (child >= 0 && next < order.size() && order[next] == child) { int32_t nc = total + (int)next; std::memcpy(&w, &nc, 4); ++next; }
                    }
                }
                rt_f4* d_all = nullptr;
                size_t all_bytes = ((size_t)total + order.size()) * 4 * sizeof(rt_f4);
                HIP_TRY(hipMalloc((void**)&d_all, all_bytes));
                s->allocs.push_back(d_all); s->device_bytes += all_bytes;
                HIP_TRY(hipMemcpy(d_all, d_nodes, bytes, hipMemcpyDeviceToDevice));
                HIP_TRY(hipMemcpy(d_all + 4 * (size_t)total, top.data(), top.size() * sizeof(rt_f4), hipMemcpyHostToDevice));
                f.nodes.insert(f.nodes.end(), top.begin(), top.end());
                f.root = total; s->top_first = total;
                d_nodes = d_all;
            }
            s->nodes_on_device = d_nodes;
        }
        rt::build_wide_bvh(f);
        if (const char* e = std::getenv("RT_MEDIA_ENCLOSE")) {   // 0: ignore that a medium's boundary encloses the whole scene (A/B measurements)
            if (std::atoi(e) == 0)
                for (size_t m = 0; m * RT_MEDIUM_F4 < f.media.size(); ++m) { int32_t fl; std::memcpy(&fl, &f.media[m * RT_MEDIUM_F4 + 5].y, 4); fl &= 1; std::memcpy(&f.media[m * RT_MEDIUM_F4 + 5].y, &fl, 4); }
        }
        RtSceneView v = f.view();
        int r;
        if ((r = upload(s, f.wnodes, &v.wnodes))) return r;
        if ((r = upload(s, f.winst_roots, &v.winst_roots))) return r;
        if ((r = upload(s, f.wsrc, &s->d_wsrc))) return r;
        if (s->nodes_on_device) v.nodes = s->nodes_on_device;
        else if ((r = upload(s, f.nodes, &v.nodes))) return r;
        if ((r = upload(s, f.spheres, &v.spheres))) return r;
        if ((r = upload(s, f.sphere_meta, &v.sphere_meta))) return r;
        if ((r = upload(s, f.mspheres, &v.mspheres))) return r;
        if ((r = upload(s, f.msphere_meta, &v.msphere_meta))) return r;
        if ((r = upload(s, f.rects, &v.rects))) return r;
        if ((r = upload(s, f.rect_meta, &v.rect_meta))) return r;
        if ((r = upload(s, f.instances, &v.instances))) return r;
        if ((r = upload(s, f.ops, &v.ops))) return r;
        if ((r = upload(s, f.media, &v.media))) return r;
        if ((r = upload(s, f.materials, &v.materials))) return r;
        if ((r = upload(s, f.textures, &v.textures))) return r;
        if ((r = upload(s, f.perlin_vec, &v.perlin_vec))) return r;
        if ((r = upload(s, f.perlin_perm, &v.perlin_perm))) return r;
        if ((r = upload(s, f.boxes, &v.boxes))) return r;
        s->view = v;
        std::memset(&s->local_surfs, 0, sizeof(s->local_surfs));
        for (size_t m = 0; m * RT_MEDIUM_F4 < f.media.size(); ++m) {
            int32_t nl, codes[3];
            std::memcpy(&nl, &f.media[m * RT_MEDIUM_F4 + 5].z, 4);
            std::memcpy(&codes[0], &f.media[m * RT_MEDIUM_F4 + 4].z, 4); std::memcpy(&codes[1], &f.media[m * RT_MEDIUM_F4 + 4].w, 4); std::memcpy(&codes[2], &f.media[m * RT_MEDIUM_F4 + 5].w, 4);
            if ((nl & 255) == 0) continue;
            s->has_local_media = true; s->local_surfs.enabled = 1;
            for (int k = 0; k < (nl & 255); ++k) {
                if (!((nl >> 8) >> k & 1) || s->local_surfs.n >= 4) continue;
                const uint32_t x = ~(uint32_t)codes[k], kind = (x >> 28) & 3u, idx = x & 0x03FFFFFFu;
                RtLocalSurfs& l = s->local_surfs;
                l.code[l.n] = kind == RT_KIND_BOX ? RT_HIT_CODE(kind, idx << 3) : RT_HIT_CODE(kind, idx);
                l.mask[l.n] = kind == RT_KIND_BOX ? ~7u : ~0u;
                l.medium[l.n] = (int32_t)m;
                l.n++;
            }
        }
        if ((r = plan_trace(s))) return r;
        if ((r = plan_shade(s))) return r;
        return RT_OK;
    }();
    if (rc != RT_OK) { rt_scene_destroy(s); return rc; }
    { DeviceWorkspace* d = device_workspace(device); std::lock_guard<std::mutex> lock(g_pool_mutex); d->handles++; s->counted = true; }
    *out = s;
    return RT_OK;
}
int rt_scene_create(const RtBuilder* b, RtObj world, int device, RtScene** out) { return rt_scene_create_ex(b, world, device, 0, out); }

void rt_scene_destroy(RtScene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    for (void* p : s->allocs) (void)hipFree(p);
    if (s->multi_acc) (void)hipFree(s->multi_acc);
    if (s->multi_sq) (void)hipFree(s->multi_sq);
    if (s->multi_stream) (void)hipStreamDestroy(s->multi_stream);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    for (hipEvent_t e : s->ev_trace) (void)hipEventDestroy(e);
    // the last handle on a GPU takes the shared workspace with it (38 GB for a default pass: a caller that closes its scenes and
    // hands the GPU to another library must get the memory back)
    bool last = false;
    if (s->counted) { DeviceWorkspace* d = device_workspace(s->device); std::lock_guard<std::mutex> lock(g_pool_mutex); last = --d->handles == 0; }
    const int device = s->device;
    delete s;
    if (last) (void)rt_release_workspace(device);
}

int rt_scene_info(const RtScene* s, RtSceneInfo* out) {
    if (!s || !out) return rt_set_error(RT_ERR_INVALID, "null argument");
    rt_fill_info(s->flat, out); out->device = s->device; out->device_bytes = s->device_bytes; out->bvh_build_seconds = s->bvh_build_seconds;
    out->bvh_refit_seconds = s->refit_seconds;
    return RT_OK;
}
}  // extern "C"

// The passes of one render call on workspace `ws`: samples [sample_offset, sample_offset + spp) of every pixel, `chunk_max`
// samples per pass.  Everything is enqueued on `st`.  h_folded (max_depth + 3 words, or null) receives the per-bounce ray
// counts of each pass: the stream is then synchronised once per pass and `local` is complete on return; with null nothing is
// read back and the call returns with the work still in flight.  Events are recorded only when the scene's are given.
static int render_passes(RtScene* s, const Workspace& ws, const RtCamera* cam, const RtRenderParams* p, int chunk_max, float* d_accum, float* sq,
                         hipStream_t st, bool use_events, unsigned long long* h_folded, RtStats& local) {
    RtCameraDev cd; make_cam(cam, cd);
    RtFrameDev fr; make_frame(p, fr);
    const size_t n_pixels = (size_t)fr.n_set;     // pixels of the call's set
    if (n_pixels == 0) return RT_OK;              // a shard without rows
    const size_t row_words = (size_t)RT_SHARDS * RT_COUNT_STRIDE;
    const size_t counts_bytes = (size_t)(p->max_depth + 2) * row_words * sizeof(uint32_t);
    uint32_t* overflow = ws.counts;                                  // word 0 of row 0
    // k_shade / k_raygen blocks per CU in the grid (grid-stride loops).  With costly textures in the scene each wave should
    // see many rays, so that the hits it postpones add up to full waves: then the grid is just what is resident.
    int shade_bpc = s->view.n_perlin > 0 ? 4 : 32;
    if (const char* e = std::getenv("RT_SHADE_BLOCKS_PER_CU")) shade_bpc = std::max(1, std::min(64, std::atoi(e)));
    const int grid_cap = std::max(1, s->n_cu) * shade_bpc;
    const uint32_t cap = (uint32_t)ws.cap;
    const bool instrumented = (p->flags & RT_FLAG_COUNTERS) != 0;
    const bool time_trace = use_events && (p->flags & RT_FLAG_TIME_TRACE) != 0;
    if (time_trace) {
        while ((int)s->ev_trace.size() < 2 * p->max_depth) { hipEvent_t e = nullptr; HIP_TRY(hipEventCreate(&e)); s->ev_trace.push_back(e); }
    }
    const bool debug_work = ws.work && std::getenv("RT_DEBUG_WORK") != nullptr;
    // rays that leave a collision inside a medium and hit a surface known to lie inside its boundary skip the closest-hit kernel
    // (shade_ray); RT_LOCAL_MEDIA=0 sends every ray through the tree
    const char* env_local = std::getenv("RT_LOCAL_MEDIA");
    const bool local_media = s->has_local_media && !(env_local && std::atoi(env_local) == 0) && ws.hits[1] && !(s->shade_fused && !instrumented);
    uint32_t* back = local_media ? ws.counts + (size_t)2 * ws.rows * row_words : nullptr;   // sizes of the regions' back parts, [bounce][region]
    RtLocalSurfs ls = s->local_surfs;
    if (!local_media) ls.enabled = 0;
    // Rays that start ON a wholly-inside surface (a reflection off the inner side of final_scene.yml's glass ball) can be settled the
    // same way; measured, it moves 1.4 ms per step from k_trace to the shading kernel and gains nothing, so it is off unless
    // RT_LOCAL_SURFACES=1: only rays that leave a collision in a medium are settled.
    { const char* e = std::getenv("RT_LOCAL_SURFACES"); if (!(e && std::atoi(e) != 0)) ls.n = 0; }
    if (instrumented || debug_work) HIP_TRY(hipMemsetAsync(ws.work, 0, 32 * sizeof(unsigned long long), st));

    for (int done = 0; done < p->spp;) {
        int chunk = std::min(chunk_max, p->spp - done);
        uint32_t n_paths = (uint32_t)((size_t)chunk * n_pixels);
        fr.spp_chunk = chunk; fr.sample_base = p->sample_offset + done;
        HIP_TRY(hipMemsetAsync(ws.counts, 0, counts_bytes, st));
        HIP_TRY(hipMemsetAsync(ws.counts + (size_t)ws.rows * row_words, 0, counts_bytes, st));
        if (back) HIP_TRY(hipMemsetAsync(back, 0, counts_bytes, st));
        int grid = (int)std::min<size_t>(((size_t)n_paths + RT_BLOCK - 1) / RT_BLOCK, (size_t)grid_cap);
        hipLaunchKernelGGL(k_raygen, dim3(grid), dim3(RT_BLOCK), 0, st, cd, fr, n_paths, cap, ws.counts + row_words, overflow, ws.qa[0], ws.qb[0], ws.contrib);
        // bounce grid: x = blocks per region (a multiple of RT_SHARDS, see the capacity argument above), y = region
        int gx = (int)std::min<size_t>(((size_t)n_paths / RT_SHARDS + RT_BLOCK - 1) / RT_BLOCK + 1, (size_t)grid_cap / RT_SHARDS);
        gx = std::max(RT_SHARDS, (gx + RT_SHARDS - 1) / RT_SHARDS * RT_SHARDS);
        dim3 bgrid((unsigned)gx, RT_SHARDS);
        // k_shade_lds: one 1024-thread block per CU, also a multiple of RT_SHARDS blocks per region
        dim3 lgrid((unsigned)std::max(RT_SHARDS, s->n_cu / RT_SHARDS / RT_SHARDS * RT_SHARDS), RT_SHARDS);
        HIP_TRY(hipGetLastError());
        if (use_events) HIP_TRY(hipEventRecord(s->ev0, st));
        for (int b = 1; b <= p->max_depth; ++b) {
            int in = (b - 1) & 1, outq = b & 1;
            const uint32_t* cin = ws.counts + (size_t)b * row_words;
            uint32_t* cout = ws.counts + (size_t)(b + 1) * row_words;
            uint32_t* cursors = ws.counts + (size_t)(ws.rows + b) * row_words;
            if (time_trace) HIP_TRY(hipEventRecord(s->ev_trace[2 * (b - 1)], st));
            const bool fused = s->shade_fused && !instrumented;
            rt_f4* hits_in = ws.hits[1] ? ws.hits[b & 1] : ws.hits[0];
            rt_f4* hits_out = local_media ? ws.hits[(b + 1) & 1] : nullptr;   // see shade_ray: rays whose closest hit the shading kernel settles
            const uint32_t* back_in = back ? back + (size_t)b * row_words : nullptr;
            uint32_t* back_out = back ? back + (size_t)(b + 1) * row_words : nullptr;
            if (!fused) launch_trace(s, ws, instrumented, st, cap, cin, cursors, ws.qa[in], ws.qb[in], hits_in);
            if (time_trace) HIP_TRY(hipEventRecord(s->ev_trace[2 * (b - 1) + 1], st));
            if (instrumented)
                hipLaunchKernelGGL(k_shade<true>, bgrid, dim3(RT_BLOCK), 0, st, s->view, fr, b, cap, cin, cout, overflow, ws.qa[in], ws.qb[in], ws.qc[in], hits_in,
                                   ws.qa[outq], ws.qb[outq], ws.qc[outq], ws.contrib, ws.work, hits_out, back_in, back_out, ls);
            else if (fused)
                hipLaunchKernelGGL(k_shade_lds<true>, lgrid, dim3(1024), (size_t)s->shade_lds_bytes, st, s->view, s->shade_plan, fr, b, cap, cin, cout, overflow,
                                   ws.qa[in], ws.qb[in], ws.qc[in], hits_in, ws.qa[outq], ws.qb[outq], ws.qc[outq], ws.contrib, hits_out, back_in, back_out, ls);
            else if (s->shade_lds)
                hipLaunchKernelGGL(k_shade_lds<false>, lgrid, dim3(1024), (size_t)s->shade_lds_bytes, st, s->view, s->shade_plan, fr, b, cap, cin, cout, overflow,
                                   ws.qa[in], ws.qb[in], ws.qc[in], hits_in, ws.qa[outq], ws.qb[outq], ws.qc[outq], ws.contrib, hits_out, back_in, back_out, ls);
            else
                hipLaunchKernelGGL(k_shade<false>, bgrid, dim3(RT_BLOCK), 0, st, s->view, fr, b, cap, cin, cout, overflow, ws.qa[in], ws.qb[in], ws.qc[in], hits_in,
                                   ws.qa[outq], ws.qb[outq], ws.qc[outq], ws.contrib, ws.work, hits_out, back_in, back_out, ls);
        }
        HIP_TRY(hipGetLastError());
        if (use_events) HIP_TRY(hipEventRecord(s->ev1, st));
        int agrid = (int)((n_pixels + RT_BLOCK - 1) / RT_BLOCK);
        hipLaunchKernelGGL(k_accumulate, dim3(agrid), dim3(RT_BLOCK), 0, st, ws.contrib, fr, chunk, d_accum, sq);
        HIP_TRY(hipGetLastError());
        local.passes++;
        done += chunk;
        if (!h_folded) continue;
        const int rows = p->max_depth + 2;
        HIP_TRY(hipMemsetAsync(ws.folded + rows, 0, sizeof(unsigned long long), st));
        hipLaunchKernelGGL(k_fold_counts, dim3((rows + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, st, ws.counts, back, rows, ws.folded);
        HIP_TRY(hipMemcpyAsync(h_folded, ws.folded, (size_t)(rows + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (back && std::getenv("RT_DEBUG_COUNTS")) {   // developer output: per bounce, the rays in the front and in the back parts of the regions
            std::vector<uint32_t> hc((size_t)3 * ws.rows * row_words);
            HIP_TRY(hipMemcpy(hc.data(), ws.counts, hc.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
            for (int b = 1; b <= p->max_depth; ++b) {
                unsigned long long f = 0, k = 0;
                for (int sh = 0; sh < RT_SHARDS; ++sh) { f += hc[((size_t)b * RT_SHARDS + sh) * RT_COUNT_STRIDE]; k += hc[((size_t)(2 * ws.rows + b) * RT_SHARDS + sh) * RT_COUNT_STRIDE]; }
                std::fprintf(stderr, "bounce %2d: front %10llu  back %10llu\n", b, f, k);
            }
        }
        if (h_folded[0] != 0)
            return rt_set_error(RT_ERR_INTERNAL, "ray queue overflow: " + std::to_string(h_folded[0]) + " rays did not fit their queue region (the image is incomplete)");
        if (use_events) {
            float ms = 0.0f;
            HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
            local.kernel_ms_extend += ms; local.extend_launches += (uint64_t)p->max_depth;
        }
        if (time_trace) {
            for (int b = 0; b < p->max_depth; ++b) {
                float tms = 0.0f;
                HIP_TRY(hipEventElapsedTime(&tms, s->ev_trace[2 * b], s->ev_trace[2 * b + 1]));
                local.kernel_ms_trace += tms;
            }
            local.trace_launches += (uint64_t)p->max_depth;
        }
        for (int b = 1; b <= p->max_depth; ++b) {
            local.rays += h_folded[b];
            if (b <= 64) local.rays_per_bounce[b - 1] += h_folded[b];
        }
        local.rays_settled += h_folded[rows];
    }
    if (instrumented && h_folded) {
        unsigned long long w[16];
        HIP_TRY(hipMemcpy(w, ws.work, sizeof(w), hipMemcpyDeviceToHost));
        for (int i = 0; i < 16; ++i) local.work[i] = w[i];
    }
    if (debug_work && h_folded) {   // developer output: wave-level counters of the instrumented k_trace
        unsigned long long w2[16];
        HIP_TRY(hipMemcpy(w2, ws.work + 16, sizeof(w2), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "rays whose closest hit the shading kernel settled: %llu\n", w2[12]);
        std::fprintf(stderr, "k_trace wave counters: outer %llu refills %llu node-iters %llu leaf-iters %llu\n", w2[0], w2[1], w2[2], w2[3]);
        std::fprintf(stderr, "k_trace stack: deepest %llu of %d; node steps %llu, of which with more than 4 / 6 / 8 / 10 / 12 / 16 entries: %.4f %.4f %.4f %.4f %.5f %.6f\n", w2[4],
                     s->plan.stack_cap, w2[5], (double)w2[6] / std::max(1ull, w2[5]), (double)w2[7] / std::max(1ull, w2[5]), (double)w2[8] / std::max(1ull, w2[5]),
                     (double)w2[9] / std::max(1ull, w2[5]), (double)w2[10] / std::max(1ull, w2[5]), (double)w2[11] / std::max(1ull, w2[5]));
    }
    return RT_OK;
}

extern "C" {

int rt_render_device(RtScene* s, const RtCamera* cam, const RtRenderParams* p, float* d_accum, float* d_accum_sq, void* hip_stream, RtStats* stats) {
    int rc = check_common(s, cam, p);
    if (rc) return rc;
    if (!d_accum) return rt_set_error(RT_ERR_INVALID, "null accumulation buffer");
    if (p->spp < 0) return rt_set_error(RT_ERR_INVALID, "spp must be >= 0");
    if ((p->flags & RT_FLAG_SUMSQ) && !d_accum_sq) return rt_set_error(RT_ERR_INVALID, "RT_FLAG_SUMSQ needs accum_sq");
    hipStream_t st = (hipStream_t)hip_stream;
    HIP_TRY(hipSetDevice(s->device));
    auto t_begin = std::chrono::steady_clock::now();
    const size_t n_pixels = std::max<size_t>(1, set_pixels(p));
    // Default pass: up to 2^28 paths in flight (38 GB of queues, hit and contribution slots on a 288 GB part).  Late
    // bounces hold few rays per launch and run below the bulk rate; the more paths share a pass, the less that costs
    // (final_scene.yml: +10 % from 2^26 to 2^28).
    size_t target = p->paths_per_pass > 0 ? (size_t)p->paths_per_pass : ((size_t)1 << 28);
    int chunk_max = (int)std::max<size_t>(1, target / n_pixels);
    if ((size_t)chunk_max * n_pixels > 0xFFFFFFF0ull) chunk_max = (int)(0xFFFFFFF0ull / n_pixels);
    chunk_max = std::min(chunk_max, std::max(1, p->spp));
    // The default pass is sized for an otherwise empty 288 GB part; when the workspace does not fit (other tenants, several
    // scene handles) the pass is halved until it does.  An explicit paths_per_pass is taken as given.
    DeviceWorkspace* pool = device_workspace(s->device);
    std::lock_guard<std::mutex> turn(pool->busy);
    while ((rc = ensure_workspace(s, pool->ws, (size_t)chunk_max * n_pixels)) == RT_ERR_OOM && p->paths_per_pass <= 0 && chunk_max > 1) {
        free_queues(pool->ws);
        (void)hipGetLastError();
        chunk_max = (chunk_max + 1) / 2;
    }
    if (rc) return rc;
    RtStats local; std::memset(&local, 0, sizeof(local));
    local.paths = (uint64_t)set_pixels(p) * (uint64_t)p->spp;
    std::vector<unsigned long long> h_folded((size_t)p->max_depth + 3);
    float* sq = (p->flags & RT_FLAG_SUMSQ) ? d_accum_sq : nullptr;
    rc = render_passes(s, pool->ws, cam, p, chunk_max, d_accum, sq, st, true, h_folded.data(), local);
    if (rc) return rc;
    local.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    if (stats) *stats = local;
    return RT_OK;
}

int rt_release_workspace(int device) {
    DeviceWorkspace* d = nullptr;
    { std::lock_guard<std::mutex> lock(g_pool_mutex); auto it = g_pool.find(device); if (it != g_pool.end()) d = it->second; }
    if (!d) return RT_OK;
    std::lock_guard<std::mutex> turn(d->busy);
    HIP_TRY(hipSetDevice(device));
    free_queues(d->ws);
    if (d->ws.counts) (void)hipFree(d->ws.counts);
    if (d->ws.folded) (void)hipFree(d->ws.folded);
    if (d->ws.work) (void)hipFree(d->ws.work);
    d->ws = Workspace();
    return RT_OK;
}

int rt_render_device_noalloc(RtScene* s, const RtCamera* cam, const RtRenderParams* p, float* d_accum, float* d_accum_sq, void* hip_stream, RtStats* stats) {
    int rc = check_common(s, cam, p);
    if (rc) return rc;
    if (!d_accum) return rt_set_error(RT_ERR_INVALID, "null accumulation buffer");
    if (p->spp < 0) return rt_set_error(RT_ERR_INVALID, "spp must be >= 0");
    if ((p->flags & RT_FLAG_SUMSQ) && !d_accum_sq) return rt_set_error(RT_ERR_INVALID, "RT_FLAG_SUMSQ needs accum_sq");
    if (p->flags & ~RT_FLAG_SUMSQ) return rt_set_error(RT_ERR_INVALID, "rt_render_device_noalloc: only RT_FLAG_SUMSQ is supported (counters and timing need allocations)");
    const size_t n_pixels = std::max<size_t>(1, set_pixels(p));
    if (n_pixels > RT_NOALLOC_MAX_PATHS)
        return rt_set_error(RT_ERR_INVALID, "rt_render_device_noalloc: the call renders " + std::to_string(n_pixels) + " pixels, which exceeds the static workspace of " +
                                                std::to_string(RT_NOALLOC_MAX_PATHS) + " paths per pass; use rt_render_device");
    if (p->max_depth > RT_NOALLOC_MAX_DEPTH)
        return rt_set_error(RT_ERR_INVALID, "rt_render_device_noalloc: max_depth exceeds " + std::to_string(RT_NOALLOC_MAX_DEPTH) + "; use rt_render_device");
    if (!s->static_scratch) return rt_set_error(RT_ERR_INTERNAL, "the static scratch array was not resolved for this device");
    hipStream_t st = (hipStream_t)hip_stream;
    HIP_TRY(hipSetDevice(s->device));
    auto t_begin = std::chrono::steady_clock::now();
    size_t target = p->paths_per_pass > 0 ? std::min<size_t>((size_t)p->paths_per_pass, RT_NOALLOC_MAX_PATHS) : (size_t)RT_NOALLOC_MAX_PATHS;
    int chunk_max = (int)std::max<size_t>(1, target / n_pixels);
    chunk_max = std::min(chunk_max, std::max(1, p->spp));
    const Workspace ws = static_workspace(s->static_scratch);
    RtStats local; std::memset(&local, 0, sizeof(local));
    local.paths = (uint64_t)set_pixels(p) * (uint64_t)p->spp;
    unsigned long long h_folded[RT_NOALLOC_MAX_DEPTH + 3];
    float* sq = (p->flags & RT_FLAG_SUMSQ) ? d_accum_sq : nullptr;
    rc = render_passes(s, ws, cam, p, chunk_max, d_accum, sq, st, false, stats ? h_folded : nullptr, local);
    if (rc) return rc;
    local.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    if (stats) *stats = local;
    return RT_OK;
}

int rt_render(RtScene* s, const RtCamera* cam, const RtRenderParams* p, float* accum, float* accum_sq, RtStats* stats) {
    int rc = check_common(s, cam, p);
    if (rc) return rc;
    if (!accum) return rt_set_error(RT_ERR_INVALID, "null accumulation buffer");
    if ((p->flags & RT_FLAG_SUMSQ) && !accum_sq) return rt_set_error(RT_ERR_INVALID, "RT_FLAG_SUMSQ needs accum_sq");
    HIP_TRY(hipSetDevice(s->device));
    size_t n = (size_t)p->width * p->height * 3;
    float *d_acc = nullptr, *d_sq = nullptr;
    HIP_TRY(hipMalloc((void**)&d_acc, n * sizeof(float)));
    hipError_t e = hipMemset(d_acc, 0, n * sizeof(float));
    if (e == hipSuccess && (p->flags & RT_FLAG_SUMSQ)) { e = hipMalloc((void**)&d_sq, n * sizeof(float)); if (e == hipSuccess) e = hipMemset(d_sq, 0, n * sizeof(float)); }
    if (e != hipSuccess) { (void)hipFree(d_acc); if (d_sq) (void)hipFree(d_sq); return rt_set_error(RT_ERR_HIP, std::string("rt_render alloc: ") + hipGetErrorString(e)); }
    rc = rt_render_device(s, cam, p, d_acc, d_sq, nullptr, stats);
    if (rc == RT_OK) {
        std::vector<float> tmp(n);
        e = hipMemcpy(tmp.data(), d_acc, n * sizeof(float), hipMemcpyDeviceToHost);
        if (e == hipSuccess) for (size_t i = 0; i < n; ++i) accum[i] += tmp[i];
        if (e == hipSuccess && d_sq) {
            e = hipMemcpy(tmp.data(), d_sq, n * sizeof(float), hipMemcpyDeviceToHost);
            if (e == hipSuccess) for (size_t i = 0; i < n; ++i) accum_sq[i] += tmp[i];
        }
        if (e != hipSuccess) rc = rt_set_error(RT_ERR_HIP, std::string("rt_render copy: ") + hipGetErrorString(e));
    }
    (void)hipFree(d_acc); if (d_sq) (void)hipFree(d_sq);
    return rc;
}

// ---- moving primitives and refit (the reference rebuilds its BvhNode for a changed world, src/main.rs:531)
static int scene_set(RtScene* s, int kind, const int32_t* ids, int32_t n, const double* values) {
    if (!s || n < 0 || (n > 0 && (!ids || !values))) return rt_set_error(RT_ERR_INVALID, "bad argument");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(s->device));
    if (!s->indexed) { rt::build_prim_index(s->flat, s->of_id); s->indexed = true; }
    std::vector<int32_t> touched; std::string err;
    if (!rt::set_prims_host(s->flat, s->of_id, kind, ids, n, values, touched, err)) return rt_set_error(rt_status_of(err), err);
    // a few records: one small copy each; many: the whole array
    const std::vector<rt_f4>& host = kind == 0 ? s->flat.spheres : s->flat.mspheres;
    rt_f4* dev = const_cast<rt_f4*>(kind == 0 ? s->view.spheres : s->view.mspheres);
    const size_t per = kind == 0 ? 1 : 3;
    if (!s->flat.media.empty()) {   // set_prims_host has withdrawn what the medium records said about the surfaces around their boundaries
        HIP_TRY(hipMemcpy(const_cast<rt_f4*>(s->view.media), s->flat.media.data(), s->flat.media.size() * sizeof(rt_f4), hipMemcpyHostToDevice));
        s->has_local_media = false; std::memset(&s->local_surfs, 0, sizeof(s->local_surfs));
    }
    if (touched.size() > 64) HIP_TRY(hipMemcpy(dev, host.data(), host.size() * sizeof(rt_f4), hipMemcpyHostToDevice));
    else for (int32_t rec : touched) HIP_TRY(hipMemcpy(dev + per * (size_t)rec, host.data() + per * (size_t)rec, per * sizeof(rt_f4), hipMemcpyHostToDevice));
    return RT_OK;
}
int rt_scene_set_spheres(RtScene* s, const int32_t* ids, int32_t n, const double* xyzr) { return scene_set(s, 0, ids, n, xyzr); }
int rt_scene_set_moving_spheres(RtScene* s, const int32_t* ids, int32_t n, const double* values) { return scene_set(s, 1, ids, n, values); }
int rt_scene_refit(RtScene* s) {
    if (!s) return rt_set_error(RT_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());       // no render of this scene may still be reading the trees
    std::string err;
    double sec = 0.0;
    if (!rt::refit_device(s->view, s->d_wsrc, s->view.root, s->flat.tlas_depth + s->flat.blas_depth + 2, &sec, err)) return rt_set_error(RT_ERR_HIP, err);
    s->refit_seconds = sec;
    // the host copy follows (it is what rt_scene_info and the launch plans describe)
    HIP_TRY(hipMemcpy(s->flat.nodes.data(), s->view.nodes, s->flat.nodes.size() * sizeof(rt_f4), hipMemcpyDeviceToHost));
    if (!s->flat.wnodes.empty()) HIP_TRY(hipMemcpy(s->flat.wnodes.data(), s->view.wnodes, s->flat.wnodes.size() * sizeof(rt_f4), hipMemcpyDeviceToHost));
    return RT_OK;
}

}  // extern "C"

// ---- RCCL, loaded on first use: a caller that renders on one GPU never needs the library.  When PyTorch is in the process its
// bundled librccl (same soname) is already mapped and is the one this finds, as with the HIP runtime.
namespace {
struct RcclApi {
    void* handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr; decltype(&ncclCommDestroy) CommDestroy = nullptr; decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr; decltype(&ncclGroupEnd) GroupEnd = nullptr; decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
};
// One communicator per GPU of an ordered device list, kept until rt_multi_shutdown.  `use` serialises the calls that enqueue on
// the set (RCCL wants one caller at a time per communicator); calls on other sets do not wait for it.
struct CommSet { std::mutex use; std::vector<ncclComm_t> comms; };
std::mutex g_rccl_mutex;            // guards g_rccl and g_rccl_comms (held only to look a set up or to create it)
RcclApi g_rccl;
std::map<std::vector<int>, std::shared_ptr<CommSet>> g_rccl_comms;

int rccl_load() {   // g_rccl_mutex held
    if (g_rccl.handle) return RT_OK;
    void* h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return rt_set_error(RT_ERR_NCCL, std::string("cannot load librccl: ") + dlerror());
    RcclApi api; api.handle = h;
#define RT_SYM(field, name) do { api.field = reinterpret_cast<decltype(api.field)>(dlsym(h, name)); \
        if (!api.field) return rt_set_error(RT_ERR_NCCL, std::string("librccl lacks ") + name); } while (0)
    RT_SYM(CommInitAll, "ncclCommInitAll"); RT_SYM(CommDestroy, "ncclCommDestroy"); RT_SYM(Reduce, "ncclReduce");
    RT_SYM(GroupStart, "ncclGroupStart"); RT_SYM(GroupEnd, "ncclGroupEnd"); RT_SYM(GetErrorString, "ncclGetErrorString");
    RT_SYM(CommCount, "ncclCommCount");
#undef RT_SYM
    g_rccl = api;
    return RT_OK;
}
#define NCCL_TRY(expr)                                                                                       \
    do {                                                                                                     \
        ncclResult_t r_ = (expr);                                                                            \
        if (r_ != ncclSuccess) return rt_set_error(RT_ERR_NCCL, std::string(#expr) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)

// sum over the GPUs of `bufs[i]` (count floats each, on stream streams[i] of device devices[i]) into bufs[0]; *ranks receives the
// size of the communicator that carried it
int rccl_reduce_to_first(const std::vector<int>& devices, const std::vector<float*>& bufs, const std::vector<hipStream_t>& streams, size_t count, int* ranks) {
    std::shared_ptr<CommSet> set;
    {
        std::lock_guard<std::mutex> lock(g_rccl_mutex);
        int rc = rccl_load();
        if (rc) return rc;
        auto it = g_rccl_comms.find(devices);
        if (it == g_rccl_comms.end()) {
            auto fresh = std::make_shared<CommSet>();
            fresh->comms.assign(devices.size(), nullptr);
            NCCL_TRY(g_rccl.CommInitAll(fresh->comms.data(), (int)devices.size(), devices.data()));   // all or nothing: on failure no communicator exists
            it = g_rccl_comms.emplace(devices, fresh).first;
        }
        set = it->second;
    }
    {
        std::lock_guard<std::mutex> turn(set->use);
        int n_ranks = 0;
        NCCL_TRY(g_rccl.CommCount(set->comms[0], &n_ranks));
        if (n_ranks != (int)devices.size()) return rt_set_error(RT_ERR_NCCL, "the communicator has " + std::to_string(n_ranks) + " ranks for " + std::to_string(devices.size()) + " GPUs");
        if (ranks) *ranks = n_ranks;
        NCCL_TRY(g_rccl.GroupStart());
        for (size_t i = 0; i < devices.size(); ++i) {
            HIP_TRY(hipSetDevice(devices[i]));
            NCCL_TRY(g_rccl.Reduce(bufs[i], bufs[i], count, ncclFloat32, ncclSum, 0, set->comms[i], streams[i]));
        }
        NCCL_TRY(g_rccl.GroupEnd());
    }
    for (size_t i = 0; i < devices.size(); ++i) { HIP_TRY(hipSetDevice(devices[i])); HIP_TRY(hipStreamSynchronize(streams[i])); }
    return RT_OK;
}
}  // namespace

extern "C" {

int rt_multi_shutdown(void) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    int rc = RT_OK;
    for (auto& kv : g_rccl_comms) {
        std::lock_guard<std::mutex> turn(kv.second->use);
        for (size_t i = 0; i < kv.second->comms.size(); ++i) {
            if (!kv.second->comms[i]) continue;
            (void)hipSetDevice(kv.first[i]);
            ncclResult_t r = g_rccl.CommDestroy(kv.second->comms[i]);
            if (r != ncclSuccess && rc == RT_OK) rc = rt_set_error(RT_ERR_NCCL, std::string("ncclCommDestroy: ") + g_rccl.GetErrorString(r));
            kv.second->comms[i] = nullptr;
        }
    }
    g_rccl_comms.clear();
    return rc;
}

int rt_render_multi(RtScene* const* scenes, int32_t n_scenes, const RtCamera* cam, const RtRenderParams* p, float* accum, float* accum_sq, RtStats* stats) {
    if (!scenes || n_scenes < 1 || n_scenes > 64) return rt_set_error(RT_ERR_INVALID, "rt_render_multi: need 1..64 scenes");
    if (!accum) return rt_set_error(RT_ERR_INVALID, "null accumulation buffer");
    if (p && p->spp < 0) return rt_set_error(RT_ERR_INVALID, "spp must be >= 0");
    if (p && (p->flags & RT_FLAG_SUMSQ) && !accum_sq) return rt_set_error(RT_ERR_INVALID, "RT_FLAG_SUMSQ needs accum_sq");
    const int n = n_scenes;
    std::vector<int> devices(n);
    for (int i = 0; i < n; ++i) {
        int rc = check_common(scenes[i], cam, p);
        if (rc) return rc;
        devices[i] = scenes[i]->device;
        for (int j = 0; j < i; ++j)
            if (scenes[j] == scenes[i]) return rt_set_error(RT_ERR_INVALID, "rt_render_multi: the same scene handle twice");
    }
    // Pixel sharding when asked for, and when there are fewer samples than handles (a preview of one sample could otherwise use one
    // GPU only): every handle then traces all samples of its row bands, the private accumulators are disjoint and no reduce is needed.
    const bool tiles = (p->flags & RT_FLAG_TILES) != 0 || (p->spp > 0 && p->spp < n);
    if (tiles && (p->stripe[0] | p->stripe[1] | p->stripe[2]) != 0) return rt_set_error(RT_ERR_INVALID, "rt_render_multi: a call that already selects a stripe cannot be sharded by pixels");
    // Handles may share a GPU (their renders then take turns on its workspace): the first handle of each GPU collects the
    // others' sums on the device, and only those first handles meet in the reduce.
    std::vector<int> leader(n);
    for (int i = 0; i < n; ++i) { leader[i] = i; for (int j = 0; j < i; ++j) if (devices[j] == devices[i]) { leader[i] = j; break; } }
    auto t_begin = std::chrono::steady_clock::now();
    const bool want_sq = (p->flags & RT_FLAG_SUMSQ) != 0;
    const size_t floats = (size_t)p->width * (size_t)p->height * 3;
    // private accumulators and a stream per GPU
    for (int i = 0; i < n; ++i) {
        RtScene* s = scenes[i];
        HIP_TRY(hipSetDevice(s->device));
        if (!s->multi_stream) HIP_TRY(hipStreamCreateWithFlags(&s->multi_stream, hipStreamNonBlocking));
        if (s->multi_floats < floats) {
            if (s->multi_acc) (void)hipFree(s->multi_acc);
            if (s->multi_sq) (void)hipFree(s->multi_sq);
            s->multi_acc = s->multi_sq = nullptr; s->multi_floats = 0;
            HIP_TRY(hipMalloc((void**)&s->multi_acc, floats * sizeof(float)));
            s->multi_floats = floats;
        }
        if (want_sq && !s->multi_sq) HIP_TRY(hipMalloc((void**)&s->multi_sq, s->multi_floats * sizeof(float)));
        HIP_TRY(hipMemsetAsync(s->multi_acc, 0, floats * sizeof(float), s->multi_stream));
        if (want_sq) HIP_TRY(hipMemsetAsync(s->multi_sq, 0, floats * sizeof(float), s->multi_stream));
    }
    // GPU i renders its share -- a contiguous sample range, or a pixel shard -- on its own host thread (rt_render_device synchronises
    // its stream)
    std::vector<RtStats> st(n);
    std::vector<int> rcs(n, RT_OK);
    std::vector<std::string> errs(n);
    std::vector<RtRenderParams> shard_params(n, *p);
    for (int i = 0; i < n; ++i) {
        RtRenderParams& pi = shard_params[i];
        pi.flags &= ~RT_FLAG_TILES;
        if (tiles) { pi.stripe[0] = RT_TILE_ROWS; pi.stripe[1] = n; pi.stripe[2] = i; }     // the rows of the caller's tile, or of the image
        else {
            int32_t first = 0, count = 0;
            rt_shard_samples(p->spp, n, i, &first, &count);
            pi.spp = count; pi.sample_offset = p->sample_offset + first;
        }
    }
    auto work = [&](int i) {
        std::memset(&st[i], 0, sizeof(RtStats));
        rcs[i] = rt_render_device(scenes[i], cam, &shard_params[i], scenes[i]->multi_acc, want_sq ? scenes[i]->multi_sq : nullptr, scenes[i]->multi_stream, &st[i]);
        if (rcs[i]) errs[i] = g_rt_error;
    };
    {
        std::vector<std::thread> threads;
        for (int i = 1; i < n; ++i) threads.emplace_back(work, i);
        work(0);
        for (auto& t : threads) t.join();
    }
    for (int i = 0; i < n; ++i) if (rcs[i]) return rt_set_error(rcs[i], "GPU " + std::to_string(devices[i]) + ": " + errs[i]);
    // a handle whose share was empty enqueued nothing after its memsets and was never synchronised: its buffers must be complete
    // before another stream reads them
    for (int i = 0; i < n; ++i) { HIP_TRY(hipSetDevice(devices[i])); HIP_TRY(hipStreamSynchronize(scenes[i]->multi_stream)); }
    std::vector<int> gpus;
    for (int i = 0; i < n; ++i) if (leader[i] == i) gpus.push_back(devices[i]);
    int comm_ranks = 0;
    std::vector<float> tmp(floats);
    if (tiles) {
        // disjoint rows: every handle hands over the bands it rendered, nothing is summed across handles
        auto gather = [&](bool sq, float* dst) -> int {
            const int y0 = (p->tile[2] > 0) ? p->tile[1] : 0, rows = (p->tile[2] > 0) ? p->tile[3] : p->height;
            const size_t row_floats = (size_t)p->width * 3;
            for (int i = 0; i < n; ++i) {
                HIP_TRY(hipSetDevice(devices[i]));
                const float* src = sq ? scenes[i]->multi_sq : scenes[i]->multi_acc;
                for (int band = i; band * RT_TILE_ROWS < rows; band += n) {
                    const int r0 = band * RT_TILE_ROWS, nr = std::min(RT_TILE_ROWS, rows - r0);
                    const size_t off = (size_t)(y0 + r0) * row_floats, cnt = (size_t)nr * row_floats;
                    HIP_TRY(hipMemcpy(tmp.data() + off, src + off, cnt * sizeof(float), hipMemcpyDeviceToHost));
                    for (size_t k = off; k < off + cnt; ++k) dst[k] += tmp[k];
                }
            }
            return RT_OK;
        };
        int rc = gather(false, accum);
        if (rc == RT_OK && want_sq) rc = gather(true, accum_sq);
        if (rc) return rc;
    } else {
        for (int i = 0; i < n; ++i) {
            if (leader[i] == i) continue;
            RtScene* to = scenes[leader[i]];
            HIP_TRY(hipSetDevice(to->device));
            const unsigned grid = (unsigned)((floats + RT_BLOCK - 1) / RT_BLOCK);
            hipLaunchKernelGGL(k_add_into, dim3(grid), dim3(RT_BLOCK), 0, to->multi_stream, to->multi_acc, scenes[i]->multi_acc, floats);
            if (want_sq) hipLaunchKernelGGL(k_add_into, dim3(grid), dim3(RT_BLOCK), 0, to->multi_stream, to->multi_sq, scenes[i]->multi_sq, floats);
            HIP_TRY(hipGetLastError());
        }
        // one reduce(SUM) of W*H*3 floats to the first GPU (the per-pixel sum of src/main.rs:585-589 lifted across GPUs); with one
        // GPU there is nothing to reduce and librccl is not needed (RT_MULTI_FORCE_RCCL=1 runs the one-rank reduce anyway)
        const char* force = std::getenv("RT_MULTI_FORCE_RCCL");
        if (gpus.size() > 1 || (force && std::atoi(force) != 0)) {
            std::vector<float*> bufs; std::vector<hipStream_t> streams;
            for (int i = 0; i < n; ++i) if (leader[i] == i) { bufs.push_back(scenes[i]->multi_acc); streams.push_back(scenes[i]->multi_stream); }
            int rc = rccl_reduce_to_first(gpus, bufs, streams, floats, &comm_ranks);
            if (rc) return rc;
            if (want_sq) {
                bufs.clear();
                for (int i = 0; i < n; ++i) if (leader[i] == i) bufs.push_back(scenes[i]->multi_sq);
                if ((rc = rccl_reduce_to_first(gpus, bufs, streams, floats, &comm_ranks))) return rc;
            }
        }
        HIP_TRY(hipSetDevice(devices[0]));
        HIP_TRY(hipStreamSynchronize(scenes[0]->multi_stream));
        HIP_TRY(hipMemcpy(tmp.data(), scenes[0]->multi_acc, floats * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t k = 0; k < floats; ++k) accum[k] += tmp[k];
        if (want_sq) {
            HIP_TRY(hipMemcpy(tmp.data(), scenes[0]->multi_sq, floats * sizeof(float), hipMemcpyDeviceToHost));
            for (size_t k = 0; k < floats; ++k) accum_sq[k] += tmp[k];
        }
    }
    if (stats) {
        RtStats tot; std::memset(&tot, 0, sizeof(tot));
        for (int i = 0; i < n; ++i) {
            tot.rays += st[i].rays; tot.paths += st[i].paths; tot.passes += st[i].passes; tot.rays_settled += st[i].rays_settled;
            tot.extend_launches += st[i].extend_launches; tot.trace_launches += st[i].trace_launches;
            tot.kernel_ms_extend = std::max(tot.kernel_ms_extend, st[i].kernel_ms_extend);
            tot.kernel_ms_trace = std::max(tot.kernel_ms_trace, st[i].kernel_ms_trace);
            for (int b = 0; b < 64; ++b) tot.rays_per_bounce[b] += st[i].rays_per_bounce[b];
            for (int k = 0; k < 16; ++k) tot.work[k] += st[i].work[k];
        }
        tot.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        tot.multi_gpus = (int32_t)gpus.size(); tot.multi_comm_ranks = comm_ranks; tot.multi_tiles = tiles ? 1 : 0;
        *stats = tot;
    }
    return RT_OK;
}

int rt_trace_primary(RtScene* s, const RtCamera* cam, const RtRenderParams* p, int32_t sample, int32_t* ids, float* ts, float* normals,
                     int32_t* fronts, float* uvs) {
    int rc = check_common(s, cam, p);
    if (rc) return rc;
    if (!ids || !ts) return rt_set_error(RT_ERR_INVALID, "null output buffer");
    HIP_TRY(hipSetDevice(s->device));
    RtCameraDev cd; make_cam(cam, cd);
    RtRenderParams whole = *p;                                      // the hook always covers the whole image
    std::memset(whole.tile, 0, sizeof(whole.tile)); std::memset(whole.stripe, 0, sizeof(whole.stripe));
    RtFrameDev fr; make_frame(&whole, fr);
    size_t n = (size_t)fr.n_pixels;
    int32_t *d_ids = nullptr, *d_fr = nullptr; float *d_t = nullptr, *d_n = nullptr, *d_uv = nullptr;
    auto cleanup = [&]() { (void)hipFree(d_ids); (void)hipFree(d_t); if (d_n) (void)hipFree(d_n); if (d_fr) (void)hipFree(d_fr); if (d_uv) (void)hipFree(d_uv); };
    rc = [&]() -> int {
        HIP_TRY(hipMalloc((void**)&d_ids, n * 4)); HIP_TRY(hipMalloc((void**)&d_t, n * 4));
        if (normals) HIP_TRY(hipMalloc((void**)&d_n, n * 12));
        if (fronts) HIP_TRY(hipMalloc((void**)&d_fr, n * 4));
        if (uvs) HIP_TRY(hipMalloc((void**)&d_uv, n * 8));
        int grid = (int)((n + RT_BLOCK - 1) / RT_BLOCK);
        if (p->flags & RT_FLAG_WAVEFRONT_PRIMARY) {
            // The first bounce of a one-sample pass as render_passes launches it: camera rays into the queue, then the closest-hit
            // kernel this scene's renders use (k_trace in the form plan_trace chose, or the fused shading kernel's own search).
            DeviceWorkspace* pool = device_workspace(s->device);
            std::lock_guard<std::mutex> turn(pool->busy);
            int r = ensure_workspace(s, pool->ws, n);
            if (r) return r;
            const Workspace& ws = pool->ws;
            const uint32_t cap = (uint32_t)ws.cap;
            const size_t row_words = (size_t)RT_SHARDS * RT_COUNT_STRIDE;
            const size_t counts_bytes = 3 * row_words * sizeof(uint32_t);
            fr.spp_chunk = 1; fr.sample_base = sample;
            HIP_TRY(hipMemsetAsync(ws.counts, 0, counts_bytes, 0));
            HIP_TRY(hipMemsetAsync(ws.counts + (size_t)ws.rows * row_words, 0, counts_bytes, 0));
            const int grid_cap = std::max(1, s->n_cu) * 32;
            hipLaunchKernelGGL(k_raygen, dim3(std::min(grid, grid_cap)), dim3(RT_BLOCK), 0, 0, cd, fr, (uint32_t)n, cap, ws.counts + row_words, ws.counts, ws.qa[0], ws.qb[0],
                               ws.contrib);
            const uint32_t* cin = ws.counts + row_words;
            if (s->shade_fused) {
                dim3 lgrid((unsigned)std::max(RT_SHARDS, s->n_cu / RT_SHARDS / RT_SHARDS * RT_SHARDS), RT_SHARDS);
                hipLaunchKernelGGL((k_shade_lds<true, true>), lgrid, dim3(1024), (size_t)s->shade_lds_bytes, 0, s->view, s->shade_plan, fr, 1, cap, cin,
                                   ws.counts + 2 * row_words, ws.counts, ws.qa[0], ws.qb[0], ws.qc[0], ws.hits[1], ws.qa[1], ws.qb[1], ws.qc[1], ws.contrib, nullptr, nullptr, nullptr, RtLocalSurfs());
            } else
                launch_trace(s, ws, false, 0, cap, cin, ws.counts + (size_t)(ws.rows + 1) * row_words, ws.qa[0], ws.qb[0], ws.hits[1]);
            dim3 fgrid((unsigned)std::max<size_t>(1, std::min<size_t>((n / RT_SHARDS + RT_BLOCK - 1) / RT_BLOCK + 1, 4096)), RT_SHARDS);
            hipLaunchKernelGGL(k_primary_finish, fgrid, dim3(RT_BLOCK), 0, 0, s->view, fr, (uint32_t)sample, cap, cin, ws.qa[0], ws.qb[0], ws.hits[1], d_ids, d_t, d_n, d_fr,
                               d_uv);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipDeviceSynchronize());
            uint32_t overflowed = 0;
            HIP_TRY(hipMemcpy(&overflowed, ws.counts, sizeof(uint32_t), hipMemcpyDeviceToHost));
            if (overflowed) return rt_set_error(RT_ERR_INTERNAL, "ray queue overflow in rt_trace_primary");
        } else
            hipLaunchKernelGGL(k_primary, dim3(grid), dim3(RT_BLOCK), 0, 0, s->view, cd, fr, (uint32_t)sample, d_ids, d_t, d_n, d_fr, d_uv);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(ids, d_ids, n * 4, hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(ts, d_t, n * 4, hipMemcpyDeviceToHost));
        if (normals) HIP_TRY(hipMemcpy(normals, d_n, n * 12, hipMemcpyDeviceToHost));
        if (fronts) HIP_TRY(hipMemcpy(fronts, d_fr, n * 4, hipMemcpyDeviceToHost));
        if (uvs) HIP_TRY(hipMemcpy(uvs, d_uv, n * 8, hipMemcpyDeviceToHost));
        return RT_OK;
    }();
    cleanup();
    return rc;
}

int rt_scatter_rays(RtScene* s, const float* rays, int32_t n, uint32_t pixel0, uint32_t sample, int32_t bounce, uint64_t seed, float* out) {
    if (!s || !rays || !out || n < 0) return rt_set_error(RT_ERR_INVALID, "bad argument");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(s->device));
    float *d_r = nullptr, *d_o = nullptr;
    int rc = [&]() -> int {
        HIP_TRY(hipMalloc((void**)&d_r, (size_t)n * 28)); HIP_TRY(hipMalloc((void**)&d_o, (size_t)n * 56));
        HIP_TRY(hipMemcpy(d_r, rays, (size_t)n * 28, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_scatter_rays, dim3((n + RT_BLOCK - 1) / RT_BLOCK), dim3(RT_BLOCK), 0, 0, s->view, d_r, n, pixel0, sample, bounce,
                           (uint32_t)seed, (uint32_t)(seed >> 32), d_o);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(out, d_o, (size_t)n * 56, hipMemcpyDeviceToHost));
        return RT_OK;
    }();
    if (d_r) (void)hipFree(d_r);
    if (d_o) (void)hipFree(d_o);
    return rc;
}

}  // extern "C"

// rt_device.h -- the fp32 path-tracing arithmetic of the gfx950 kernels, written as inline
// functions so that render.hip can arrange them into wavefront stages.  The functions restate,
// in fp32 and without recursion or virtual dispatch, the per-ray work of the reference:
//   camera ray          cameras.rs:82-98        closest hit   bvh.rs:159-187, objects.rs:67-78
//   sphere              objects.rs:205-239      moving sphere objects.rs:292-326
//   rectangle           objects.rs:384-411      instances     transrot.rs:20-32, 92-116
//   constant medium     mediums.rs:39-87        materials     materials.rs:77-90,129-145,181-209,243-249,276-283
//   textures            textures.rs:28-32,60-69,87-96,131-170,214-237
// The file also compiles as plain C++ (RT_HD then means "static inline"): tests/emul builds it on
// the CPU to debug the fp32 logic without a GPU.  That build is test-only and never shipped.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "rt_types.h"

#if defined(__HIP__)   /* compiled as HIP (a .cpp built by hipcc is plain C++) */
#define RT_HD __host__ __device__ __forceinline__
#define RT_HDM __host__ __device__ __forceinline__   /* member functions */
#else
#define RT_HD static inline
#define RT_HDM inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define RT_ON_DEVICE 1
#else
#define RT_ON_DEVICE 0
#endif

struct v3 { float x, y, z; };
RT_HD v3 mk3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD v3 operator+(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD v3 operator-(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD v3 operator-(v3 a) { return mk3(-a.x, -a.y, -a.z); }
RT_HD v3 operator*(float s, v3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
RT_HD v3 operator*(v3 a, v3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD float dot3(v3 a, v3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
RT_HD v3 fma3(float s, v3 a, v3 b) { return mk3(fmaf(s, a.x, b.x), fmaf(s, a.y, b.y), fmaf(s, a.z, b.z)); }

RT_HD float rt_nan() { return __builtin_nanf(""); }
RT_HD float rt_inf() { return __builtin_huge_valf(); }
RT_HD int f2i(float f) { int i; memcpy(&i, &f, 4); return i; }
RT_HD float i2f(int i) { float f; memcpy(&f, &i, 4); return f; }

// 1-ulp hardware reciprocal / sqrt on the device (v_rcp_f32, v_sqrt_f32); IEEE on the host.
RT_HD float rt_rcp(float x) {
#if RT_ON_DEVICE
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
RT_HD float rt_sqrt(float x) {
#if RT_ON_DEVICE
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
// sin and cos of 2*pi*u for u in [0,1): v_sin_f32 / v_cos_f32 take their argument in revolutions.
RT_HD void rt_sincos_2pi(float u, float* s, float* c) {
#if RT_ON_DEVICE
    *s = __builtin_amdgcn_sinf(u);
    *c = __builtin_amdgcn_cosf(u);
#else
    *s = sinf(6.283185307179586f * u);
    *c = cosf(6.283185307179586f * u);
#endif
}
RT_HD float rt_log10(float x) {
#if RT_ON_DEVICE
    return __builtin_amdgcn_logf(x) * 0.30102999566398120f;  // v_log_f32 is log2
#else
    return log2f(x) * 0.30102999566398120f;
#endif
}
RT_HD v3 normalize3(v3 a) {
    float l2 = dot3(a, a);
    return rt_rcp(rt_sqrt(l2)) * a;
}

// Marks a value that is the same in every lane of the wave (a record index read from a wave-uniform record):
// the compiler can then address through SGPRs and fetch with scalar loads instead of 64-lane vector loads.
RT_HD int rt_uniform(int v) {
#if RT_ON_DEVICE
    return __builtin_amdgcn_readfirstlane(v);
#else
    return v;
#endif
}

RT_HD rt_f4 ld4(const rt_f4* p) {
#if RT_ON_DEVICE
    const float4 v = *reinterpret_cast<const float4*>(p);
    rt_f4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r;
#else
    return *p;
#endif
}

// ------------------------------------------------------------------ Philox4x32-10
// Counter (pixel, sample, bounce, block), key = render seed.  Draw schedule:
//   bounce 0 / block 0: pixel jitter x, y, lens a, lens b     block 1: shutter time
//   bounce b: draw k lives in block k/4, lane k%4.  k = 0, 1: unit-vector u1, u2 (a dielectric uses k = 0 for its
//             reflect-or-refract decision: no material needs both); k = 2 + m: free flight in medium m
// A draw is the top 24 bits * 2^-24, exactly representable in fp32.
RT_HD void rt_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        if (r) { k0 += W0; k1 += W1; }
        uint64_t p0 = (uint64_t)M0 * c0, p1 = (uint64_t)M1 * c2;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
RT_HD float rt_u24(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

// utils.rs:13-22: the reference normalises its in-ball rejection sample, i.e. it draws a uniform
// unit vector; this is the direct map with the same distribution.
RT_HD v3 rt_unit_vector(float u1, float u2) {
    float z = 1.0f - 2.0f * u1;
    float r = rt_sqrt(fmaxf(0.0f, 1.0f - z * z));
    float s, c; rt_sincos_2pi(u2, &s, &c);
    return mk3(r * c, r * s, z);
}

// q = a / n, r = a % n for a, n < 2^32 with inv = 1.0 / n: the quotient comes from one f64 multiply and is exact after one
// correction step, which is cheaper than a 32-bit integer division.
RT_HD void rt_divmod(uint32_t a, int32_t n, double inv, uint32_t& q, uint32_t& r) {
    uint32_t qq = (uint32_t)((double)a * inv);
    int64_t rr = (int64_t)a - (int64_t)qq * (int64_t)n;
    if (rr < 0) { qq--; rr += n; } else if (rr >= n) { qq++; rr -= n; }
    q = qq; r = (uint32_t)rr;
}
// Pixel index (row * width + column) of entry k of the pass's pixel set (see RtFrameDev).
RT_HD uint32_t rt_set_pixel(const RtFrameDev& fr, uint32_t k) {
    if (!fr.subset) return k;
    uint32_t row, col;
    rt_divmod(k, fr.set_w, fr.inv_set_w, row, col);
    if (fr.band > 0) {   // row `row` of the shard's rows -> row of the rectangle: band (row / band) of this shard is band (row / band) * n_shards + shard
        uint32_t j = row / (uint32_t)fr.band;
        row = (j * (uint32_t)fr.n_shards + (uint32_t)fr.shard) * (uint32_t)fr.band + (row - j * (uint32_t)fr.band);
    }
    return ((uint32_t)fr.set_y0 + row) * (uint32_t)fr.width + (uint32_t)fr.set_x0 + col;
}
// Path id = sample_in_pass * n_set + set index.
RT_HD void rt_split_path(const RtFrameDev& fr, uint32_t pid, uint32_t& pixel, uint32_t& s_local) {
    uint32_t k;
    rt_divmod(pid, fr.n_set, fr.inv_n_set, s_local, k);
    pixel = rt_set_pixel(fr, k);
}

// ------------------------------------------------------------------ camera (cameras.rs:82-98)
RT_HD void rt_camera_ray(const RtCameraDev& cam, const RtFrameDev& fr, uint32_t pixel, uint32_t sample,
                         v3& o, v3& d, float& time) {
    uint32_t r0[4], r1[4];
    rt_philox(pixel, sample, 0u, 0u, fr.key0, fr.key1, r0);
    rt_philox(pixel, sample, 0u, 1u, fr.key0, fr.key1, r1);
    int px = (int)(pixel % (uint32_t)fr.width), py = (int)(pixel / (uint32_t)fr.width);
    int j = fr.height - 1 - py;                                    // main.rs:567,595
    float s = ((float)px + rt_u24(r0[0])) * fr.inv_wm1;            // main.rs:572
    float t = ((float)j + rt_u24(r0[1])) * fr.inv_hm1;             // main.rs:573
    float rad = rt_sqrt(rt_u24(r0[2])), sn, cs;                    // cameras.rs:103-113 (direct disk map)
    rt_sincos_2pi(rt_u24(r0[3]), &sn, &cs);
    float dx = cam.lens_radius * rad * cs, dy = cam.lens_radius * rad * sn;
    v3 U = mk3(cam.u[0], cam.u[1], cam.u[2]), V = mk3(cam.v[0], cam.v[1], cam.v[2]);
    v3 off = fma3(dx, U, dy * V);
    time = (cam.shutter_open != 0.0f || cam.shutter_close != 0.0f)
               ? fmaf(cam.shutter_close - cam.shutter_open, rt_u24(r1[0]), cam.shutter_open) : 0.0f;
    o = mk3(cam.origin[0], cam.origin[1], cam.origin[2]) + off;
    v3 H = mk3(cam.horizontal[0], cam.horizontal[1], cam.horizontal[2]);
    v3 Vv = mk3(cam.vertical[0], cam.vertical[1], cam.vertical[2]);
    d = fma3(s, H, fma3(t, Vv, mk3(cam.dir0[0], cam.dir0[1], cam.dir0[2]))) - off;
}

// ------------------------------------------------------------------ primitive tests
// All return the accepted t, or NaN for "no hit in [tmin, tmax]".
//
// Sphere (objects.rs:205-239): nearer root if in range, else the farther one.  The two roots come from
// q = b' + sign(b') sqrt(disc) as q/a and c/q, which avoids the cancellation of -b' +- sqrt(disc)
// (Haines et al., "Precision improvements for ray/sphere intersection", Ray Tracing Gems ch. 7).
// `self` = the ray starts on this very sphere: the roots are then exactly {0, 2b'/a}; the zero
// root is what the reference discards through t_min = 0.001.
RT_HD bool rt_sphere_roots(v3 c, float r, v3 o, v3 d, float& tn, float& tf) {
    // The cancelling terms of objects.rs:206-210 (oc = o - c, c = |oc|^2 - r^2, discriminant) are formed in f64:
    // in fp32 they lose all digits for the r = 1000 ground spheres seen from 3 units above, and for every
    // grazing hit.  gfx950 issues f64 FMAs at close to the fp32 FMA rate, and doing it always (instead of only when
    // fp32 would fail) keeps the 64 lanes of a wave on one code path.
    double fx = (double)o.x - (double)c.x, fy = (double)o.y - (double)c.y, fz = (double)o.z - (double)c.z;
    double dx = d.x, dy = d.y, dz = d.z, rr = r;
    // explicit fused multiply-adds: the file is compiled with -ffp-contract=off (so that the host build of these functions
    // rounds like the device), and the plain expressions would issue 26 f64 instructions where these issue 15
    double A = fma(dx, dx, fma(dy, dy, dz * dz));
    double B = -fma(fx, dx, fma(fy, dy, fz * dz));
    double Cc = fma(fx, fx, fma(fy, fy, fma(fz, fz, -(rr * rr))));
    double D = fma(B, B, -(A * Cc));
    if (!(D >= 0.0)) return false;
    float a = (float)A, bp = (float)B, cc = (float)Cc;
    float q = bp + copysignf(rt_sqrt((float)D), bp);
    float ta = q * rt_rcp(a), tb = cc * rt_rcp(q);
    tn = fminf(ta, tb); tf = fmaxf(ta, tb);
    return true;
}
RT_HD float rt_hit_sphere(v3 c, float r, v3 o, v3 d, float tmin, float tmax, bool self) {
    if (self) {
        float bp = -dot3(o - c, d);
        float t = 2.0f * bp * rt_rcp(dot3(d, d));
        return (bp > 0.0f && t >= tmin && t <= tmax) ? t : rt_nan();
    }
    float tn, tf;
    if (!rt_sphere_roots(c, r, o, d, tn, tf)) return rt_nan();
    if (tn >= tmin && tn <= tmax) return tn;
    if (tf >= tmin && tf <= tmax) return tf;
    return rt_nan();
}

// Rectangle (objects.rs:384-411).  type 0 = Xy, 1 = Xz, 2 = Yz.  A NaN t (ray inside the plane
// with a zero direction component) is a miss here; the reference's comparisons let it through.
// `inv` = (1/d.x, 1/d.y, 1/d.z) as rt_rcp gives them: callers that traverse already hold it.
RT_HD float rt_hit_rect(rt_f4 r0, float k, int type, v3 o, v3 d, v3 inv, float tmin, float tmax) {
    float oa, ia, ox, dx, oy, dy;
    if (type == 0) { oa = o.z; ia = inv.z; ox = o.x; dx = d.x; oy = o.y; dy = d.y; }
    else if (type == 1) { oa = o.y; ia = inv.y; ox = o.x; dx = d.x; oy = o.z; dy = d.z; }
    else { oa = o.x; ia = inv.x; ox = o.y; dx = d.y; oy = o.z; dy = d.z; }
    float t = (k - oa) * ia;
    if (!(t >= tmin && t <= tmax)) return rt_nan();
    float x = fmaf(t, dx, ox), y = fmaf(t, dy, oy);
    if (x < r0.x || x > r0.y || y < r0.z || y > r0.w) return rt_nan();
    return t;
}

// BoxObj (objects.rs:433-512) as one primitive.  The reference scans the six rectangles and keeps the closest hit
// in range; for a convex box those hits are the entry and the exit face of a slab test, and each slab distance
// (k - o[a]) / d[a] is the very expression of the rectangle test (objects.rs:390).  Returns the accepted t (NaN:
// none) and the side index in `face` (0..5 = z max, z min, y max, y min, x max, x min).  `skip` = side the ray
// starts on (-1: none), which a ray can never re-hit.
RT_HD bool rt_box_slabs(rt_f4 b0, rt_f4 b1, v3 o, v3 d, v3 inv, float& t_in, float& t_out, int& f_in, int& f_out) {
    float ix = inv.x, iy = inv.y, iz = inv.z;
    float x0 = (b0.x - o.x) * ix, x1 = (b1.x - o.x) * ix;
    float y0 = (b0.y - o.y) * iy, y1 = (b1.y - o.y) * iy;
    float z0 = (b0.z - o.z) * iz, z1 = (b1.z - o.z) * iz;
    float ex = fminf(x0, x1), ey = fminf(y0, y1), ez = fminf(z0, z1);
    float lx = fmaxf(x0, x1), ly = fmaxf(y0, y1), lz = fmaxf(z0, z1);
    t_in = fmaxf(fmaxf(ex, ey), ez); t_out = fminf(fminf(lx, ly), lz);
    // side through which the ray enters: the axis whose near plane is farthest; its min side when d > 0
    f_in = t_in == ez ? (d.z > 0.0f ? 1 : 0) : (t_in == ey ? (d.y > 0.0f ? 3 : 2) : (d.x > 0.0f ? 5 : 4));
    f_out = t_out == lz ? (d.z > 0.0f ? 0 : 1) : (t_out == ly ? (d.y > 0.0f ? 2 : 3) : (d.x > 0.0f ? 4 : 5));
    return t_in <= t_out;
}
RT_HD float rt_hit_box(rt_f4 b0, rt_f4 b1, v3 o, v3 d, v3 inv, float tmin, float tmax, int skip, int& face) {
    float t_in, t_out; int f_in, f_out;
    if (!rt_box_slabs(b0, b1, o, d, inv, t_in, t_out, f_in, f_out)) return rt_nan();
    if (t_in >= tmin && t_in <= tmax && f_in != skip) { face = f_in; return t_in; }
    if (t_out >= tmin && t_out <= tmax && f_out != skip) { face = f_out; return t_out; }
    return rt_nan();
}

RT_HD v3 rt_msphere_center(rt_f4 m0, rt_f4 m1, rt_f4 m2, float time) {  // objects.rs:285-289
    float s = (time - m1.w) * m2.x;
    return mk3(fmaf(s, m1.x, m0.x), fmaf(s, m1.y, m0.y), fmaf(s, m1.z, m0.z));
}

// Slab test (bvh.rs:46-61) against the interval [tmin, tmax]; returns the entry t or +inf.
// fminf/fmaxf drop NaNs (0 * inf when a direction component is 0 and the origin lies on the slab
// plane), which keeps the test conservative.
// rt_slab_hit: whether the box is hit, and the entry t in `tn` (meaningful only then).
RT_HD bool rt_slab_hit(float lx, float ly, float lz, float hx, float hy, float hz, v3 o, v3 inv, float tmin, float tmax, float& tn) {
    float t0x = (lx - o.x) * inv.x, t1x = (hx - o.x) * inv.x;
    float t0y = (ly - o.y) * inv.y, t1y = (hy - o.y) * inv.y;
    float t0z = (lz - o.z) * inv.z, t1z = (hz - o.z) * inv.z;
    tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), tmin));
    float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fminf(fmaxf(t0z, t1z), tmax));
    return tn <= tf;
}
RT_HD float rt_slab(float lx, float ly, float lz, float hx, float hy, float hz, v3 o, v3 inv, float tmin, float tmax) {
    float tn;
    return rt_slab_hit(lx, ly, lz, hx, hy, hz, o, inv, tmin, tmax, tn) ? tn : rt_inf();
}

// ------------------------------------------------------------------ instances (transrot.rs)
RT_HD v3 rt_to_object_point(rt_f4 i0, rt_f4 i1, v3 p) {
    return mk3(fmaf(i0.x, p.x, fmaf(-i0.y, p.z, i0.z)), p.y + i0.w, fmaf(i0.y, p.x, fmaf(i0.x, p.z, i1.x)));
}
RT_HD v3 rt_to_object_dir(rt_f4 i0, v3 d) {
    return mk3(fmaf(i0.x, d.x, -i0.y * d.z), d.y, fmaf(i0.y, d.x, i0.x * d.z));
}

struct RtHit { float t; uint32_t code; int32_t inst; };

// Work counters of an instrumented build (STATS = true); the normal kernels compile them away.
struct RtCounters { uint32_t nodes, spheres, mspheres, rects, instances, boxes, wnodes; };

// ------------------------------------------------------------------ closest surface hit
// Stack-based traversal of the two-level BVH; equivalent to bvh.rs:159-187 + objects.rs:67-78
// (closest hit is independent of tree shape and visiting order, ties aside).  `src` is the hit
// code of the surface the ray starts on (RT_HIT_NONE for camera rays and medium scatters): a
// planar primitive can never be re-hit from itself and a sphere only through its far root, which
// replaces the reference's reliance on f64 round-off staying below t_min.
//
// The traversal is written as three steps on an explicit state so that the plain per-ray loop below and the
// persistent kernel (which refills finished lanes with new rays between steps) share them.
//   Stack : see rt_pop.      Nodes : get(node, q[4]) fetches the 64-byte node record.      Prims : see RtGlobalPrims.
struct RtTrav { v3 o, d, inv; int32_t inst, sp, cur; };
RT_HD void rt_trav_ray(RtTrav& tv, v3 o, v3 d) {
    tv.o = o; tv.d = d; tv.inv = mk3(rt_rcp(d.x), rt_rcp(d.y), rt_rcp(d.z));
}

// A Nodes source: get(node, q[4]) fetches the 64-byte record of an inner node; kWide tells which layout the records have
// (false: binary node of rt_types.h, true: 4-wide quantised node of wide_bvh.cpp); root(sc) = the top-level root;
// instance_root(i, r) = the root of instance i's subtree, r being the binary tree's root stored in the instance record.
struct RtGlobalNodes {
    static constexpr bool kWide = false;
    const rt_f4* nodes;
    RT_HDM void get(int32_t n, rt_f4 q[4]) const {
        const rt_f4* p = nodes + 4 * (size_t)n;
        q[0] = ld4(p); q[1] = ld4(p + 1); q[2] = ld4(p + 2); q[3] = ld4(p + 3);
    }
    RT_HDM int32_t root(const RtSceneView& sc) const { return sc.root; }
    RT_HDM int32_t instance_root(int32_t, int32_t binary_root) const { return binary_root; }
};
struct RtGlobalWideNodes {
    static constexpr bool kWide = true;
    const rt_f4* nodes; const int32_t* inst_roots;
    RT_HDM void get(int32_t n, rt_f4 q[4]) const {
        const rt_f4* p = nodes + 4 * (size_t)n;
        q[0] = ld4(p); q[1] = ld4(p + 1); q[2] = ld4(p + 2); q[3] = ld4(p + 3);
    }
    RT_HDM int32_t root(const RtSceneView& sc) const { return sc.wroot; }
    RT_HDM int32_t instance_root(int32_t i, int32_t) const { return inst_roots[i]; }
};

// Primitive records as the leaf step fetches them.  This one reads the scene's global arrays; the closest-hit kernel
// has its own that serves arrays resident in LDS.
struct RtGlobalPrims {
    const RtSceneView* sc;
    RT_HDM rt_f4 sphere(uint32_t i) const { return ld4(sc->spheres + i); }
    RT_HDM void msphere(uint32_t i, rt_f4& m0, rt_f4& m1, rt_f4& m2) const {
        const rt_f4* m = sc->mspheres + 3 * (size_t)i; m0 = ld4(m); m1 = ld4(m + 1); m2 = ld4(m + 2);
    }
    RT_HDM void rect(uint32_t i, rt_f4& r0, rt_f4& r1) const { r0 = ld4(sc->rects + 2 * (size_t)i); r1 = ld4(sc->rects + 2 * (size_t)i + 1); }
    RT_HDM void box(uint32_t i, rt_f4& b0, rt_f4& b1) const { b0 = ld4(sc->boxes + 2 * (size_t)i); b1 = ld4(sc->boxes + 2 * (size_t)i + 1); }
    RT_HDM void instance(uint32_t i, rt_f4& i0, rt_f4& i1) const {
        i0 = ld4(sc->instances + 2 * (size_t)i); i1 = ld4(sc->instances + 2 * (size_t)i + 1);
    }
};

// Everything shading reads besides the ray and its hit: the primitive records above plus their (material, object id)
// pairs, instance ops, materials, textures, medium records and Perlin tables.  This one reads global memory; the
// shading kernel for small scenes has its own that serves a copy in LDS.
struct RtGlobalTables : RtGlobalPrims {
    RT_HDM rt_i2 sphere_meta(uint32_t i) const { return sc->sphere_meta[i]; }
    RT_HDM rt_i2 msphere_meta(uint32_t i) const { return sc->msphere_meta[i]; }
    RT_HDM rt_i2 rect_meta(uint32_t i) const { return sc->rect_meta[i]; }
    RT_HDM rt_f4 op(int i) const { return ld4(sc->ops + i); }
    RT_HDM void material(int i, rt_f4& a0, rt_f4& a1) const { a0 = ld4(sc->materials + 2 * (size_t)i); a1 = ld4(sc->materials + 2 * (size_t)i + 1); }
    RT_HDM void texture(int i, rt_f4& t0, rt_f4& t1) const { t0 = ld4(sc->textures + 2 * (size_t)i); t1 = ld4(sc->textures + 2 * (size_t)i + 1); }
    RT_HDM rt_f4 medium(int m, int k) const { return ld4(sc->media + RT_MEDIUM_F4 * (size_t)m + k); }
    RT_HDM rt_f4 perlin_vec(int table, int h) const { return ld4(sc->perlin_vec + (size_t)table * RT_PERLIN_VEC_F4 + h); }
    RT_HDM int perlin_perm(int table, int i) const { return sc->perlin_perm[(size_t)table * RT_PERLIN_PERM_BYTES + i]; }
};

RT_HD bool rt_is_inner(int32_t cur) { return (uint32_t)cur < (uint32_t)RT_SENTINEL; }

template <class Nodes>
RT_HD void rt_trav_begin(const RtSceneView& sc, const Nodes& nodes, v3 wo, v3 wd, RtTrav& tv) {
    rt_trav_ray(tv, wo, wd);
    const int32_t root = nodes.root(sc);
    tv.inst = -1; tv.sp = 0; tv.cur = root < 0 ? RT_EMPTY_STACK : root;
}
// A Stack holds the postponed children of one ray.  put(i, node, which, code): entry i = child `which` (0 left, 1 right) of
// inner node `node`, whose child code is `code` (a stack may keep either form); put_sentinel(i); take(i) returns the
// child code or RT_SENTINEL; cap() = entries available.
template <class Stack>
RT_HD int32_t rt_pop(RtTrav& tv, Stack& stk) { return tv.sp > 0 ? stk.take(--tv.sp) : RT_EMPTY_STACK; }

// tv.cur is an inner node: test both child boxes, descend into the nearer hit child, push the farther one.
template <bool STATS, class Nodes, class Stack>
RT_HD void rt_node_step(const Nodes& nodes, RtTrav& tv, float tmin, const RtHit& hit, Stack& stk, RtCounters* wc) {
    rt_f4 q[4];
    const int32_t node = tv.cur;
    nodes.get(node, q);
    if (STATS) wc->nodes++;
    float tl, tr;
    bool hl = rt_slab_hit(q[0].x, q[0].y, q[0].z, q[0].w, q[1].x, q[1].y, tv.o, tv.inv, tmin, hit.t, tl);
    bool hr = rt_slab_hit(q[1].z, q[1].w, q[2].x, q[2].y, q[2].z, q[2].w, tv.o, tv.inv, tmin, hit.t, tr);
    int32_t cl = f2i(q[3].x), cr = f2i(q[3].y);
    if (hl && hr) {
        bool left_first = tl <= tr;
        if (tv.sp < stk.cap()) stk.put(tv.sp++, node, left_first ? 1 : 0, left_first ? cr : cl);
        tv.cur = left_first ? cl : cr;
    } else if (hl) tv.cur = cl;
    else if (hr) tv.cur = cr;
    else tv.cur = rt_pop(tv, stk);
}

// The same step on a 4-wide quantised node (wide_bvh.cpp): the four child boxes are tested straight from their bytes --
// t = byte * (step / d) + (origin - o) / d per plane -- the hit children are ordered by entry distance, the nearest is
// entered and the others are pushed farthest first.  The order comes from sorting four 32-bit keys, the entry distance's
// bit pattern (positive floats order like integers), with the child codes carried along.
RT_HD float rt_ubyte(uint32_t w, int k) { return (float)((w >> (8 * k)) & 255u); }
template <bool STATS, class Nodes, class Stack>
RT_HD void rt_node_step4(const Nodes& nodes, RtTrav& tv, float tmin, const RtHit& hit, Stack& stk, RtCounters* wc) {
    rt_f4 q[4];
    const int32_t node = tv.cur;
    nodes.get(node, q);
    if (STATS) wc->wnodes++;
    const float ax = q[0].w * tv.inv.x, ay = q[3].z * tv.inv.y, az = q[3].w * tv.inv.z;
    const float bx = (q[0].x - tv.o.x) * tv.inv.x, by = (q[0].y - tv.o.y) * tv.inv.y, bz = (q[0].z - tv.o.z) * tv.inv.z;
    // Per axis the ray enters through the child's lo plane when it travels in + direction, else through its hi plane: the
    // near and far byte words are chosen once per node, and a child costs six conversions and six multiply-adds.  (A zero
    // direction component makes 0 * inf = NaN planes, which max3 / min3 drop: that axis then does not constrain the box.)
    const bool px = tv.d.x >= 0.0f, py = tv.d.y >= 0.0f, pz = tv.d.z >= 0.0f;
    const uint32_t lox = (uint32_t)f2i(q[1].x), loy = (uint32_t)f2i(q[1].y), loz = (uint32_t)f2i(q[1].z), hix = (uint32_t)f2i(q[1].w),
                   hiy = (uint32_t)f2i(q[2].x), hiz = (uint32_t)f2i(q[2].y);
    const uint32_t nx = px ? lox : hix, fx = px ? hix : lox, ny = py ? loy : hiy, fy = py ? hiy : loy, nz = pz ? loz : hiz, fz = pz ? hiz : loz;
    int32_t c0 = f2i(q[2].z), c1 = f2i(q[2].w), c2 = f2i(q[3].x), c3 = f2i(q[3].y);
    uint32_t k0, k1, k2, k3;
#define RT_WIDE_CHILD(K, CODE, KEY)                                                                                       \
    {                                                                                                                     \
        float tn = fmaxf(fmaxf(fmaf(rt_ubyte(nx, K), ax, bx), fmaf(rt_ubyte(ny, K), ay, by)), fmaxf(fmaf(rt_ubyte(nz, K), az, bz), tmin));   \
        float tf = fminf(fminf(fmaf(rt_ubyte(fx, K), ax, bx), fmaf(rt_ubyte(fy, K), ay, by)), fminf(fmaf(rt_ubyte(fz, K), az, bz), hit.t));  \
        KEY = (tn <= tf && CODE != RT_EMPTY_STACK) ? (uint32_t)f2i(tn) : 0xFFFFFFFFu;   /* an absent child: code RT_EMPTY_STACK and an inverted box (lo 255, hi 0) */ \
    }
    RT_WIDE_CHILD(0, c0, k0) RT_WIDE_CHILD(1, c1, k1) RT_WIDE_CHILD(2, c2, k2) RT_WIDE_CHILD(3, c3, k3)
#undef RT_WIDE_CHILD
#define RT_CSWAP(KA, CA, KB, CB) { bool sw = KB < KA; uint32_t kl = sw ? KB : KA, kh = sw ? KA : KB; int32_t cl = sw ? CB : CA, ch = sw ? CA : CB; KA = kl; KB = kh; CA = cl; CB = ch; }
#ifdef RT_WIDE_NEAREST_ONLY   /* experiment: only the nearest child is singled out, the others are pushed in slot order */
    RT_CSWAP(k0, c0, k1, c1) RT_CSWAP(k2, c2, k3, c3) RT_CSWAP(k0, c0, k2, c2)
#else
    RT_CSWAP(k0, c0, k1, c1) RT_CSWAP(k2, c2, k3, c3) RT_CSWAP(k0, c0, k2, c2) RT_CSWAP(k1, c1, k3, c3) RT_CSWAP(k1, c1, k2, c2)
#endif
#undef RT_CSWAP
    // Misses sort last: the hit children are k0 .. k(n-1).  The stack holds the tree's worst case by construction
    // (wide_bvh.cpp computes it and the launch plan sizes the stack from it).
    if (k3 != 0xFFFFFFFFu) stk.put(tv.sp++, node, 0, c3);
    if (k2 != 0xFFFFFFFFu) stk.put(tv.sp++, node, 0, c2);
    if (k1 != 0xFFFFFFFFu) stk.put(tv.sp++, node, 0, c1);
    tv.cur = k0 != 0xFFFFFFFFu ? c0 : rt_pop(tv, stk);
}
template <bool STATS, class Nodes, class Stack>
RT_HD void rt_inner_step(const Nodes& nodes, RtTrav& tv, float tmin, const RtHit& hit, Stack& stk, RtCounters* wc) {
    if constexpr (Nodes::kWide) rt_node_step4<STATS>(nodes, tv, tmin, hit, stk, wc);
    else rt_node_step<STATS>(nodes, tv, tmin, hit, stk, wc);
}

// tv.cur is the sentinel pushed when an instance was entered: back to the world-space ray.
template <class Stack>
RT_HD void rt_sentinel_step(v3 wo, v3 wd, RtTrav& tv, Stack& stk) {
    rt_trav_ray(tv, wo, wd); tv.inst = -1;
    tv.cur = rt_pop(tv, stk);
}

// `leaf` is the code of a primitive leaf (< 0, not an instance): intersect its primitives with the ray held in tv.
template <bool STATS, class Prims>
RT_HD void rt_leaf_test(const Prims& prims, int32_t leaf, float time, float tmin, uint32_t src, const RtTrav& tv, RtHit& hit, RtCounters* wc) {
    uint32_t x = ~(uint32_t)leaf;
    uint32_t kind = (x >> 28) & 3u, cnt = ((x >> 26) & 3u) + 1u, first = x & 0x03FFFFFFu;
    for (uint32_t k = 0; k < cnt; ++k) {
        uint32_t idx = first + k, code = RT_HIT_CODE(kind, idx);
        float t;
        if (kind == RT_KIND_BOX) {
            if (STATS) wc->boxes++;
            rt_f4 b0, b1;
            prims.box(idx, b0, b1);
            int skip = (src >> 3) == (RT_HIT_CODE(RT_KIND_BOX, idx << 3) >> 3) ? (int)(src & 7u) : -1, face = 0;
            t = rt_hit_box(b0, b1, tv.o, tv.d, tv.inv, tmin, hit.t, skip, face);
            code = RT_HIT_CODE(RT_KIND_BOX, (idx << 3) | (uint32_t)face);
        } else if (kind == RT_KIND_RECT) {
            if (STATS) wc->rects++;
            if (code == src) continue;
            rt_f4 r0, r1;
            prims.rect(idx, r0, r1);
            t = rt_hit_rect(r0, r1.x, f2i(r1.y), tv.o, tv.d, tv.inv, tmin, hit.t);
        } else {  // sphere or moving sphere: one intersection path, the centre is the only difference
            v3 c; float r;
            if (kind == RT_KIND_SPHERE) {
                rt_f4 sp = prims.sphere(idx);
                c = mk3(sp.x, sp.y, sp.z); r = sp.w;
                if (STATS) wc->spheres++;
            } else {
                rt_f4 m0, m1, m2;
                prims.msphere(idx, m0, m1, m2);
                c = rt_msphere_center(m0, m1, m2, time); r = m0.w;
                if (STATS) wc->mspheres++;
            }
            t = rt_hit_sphere(c, r, tv.o, tv.d, tmin, hit.t, code == src);
        }
        if (t < hit.t) { hit.t = t; hit.code = code; hit.inst = tv.inst; }  // false for NaN
    }
}

// tv.cur is an instance leaf: push the sentinel, take the ray to the instance's object space, go to its root.
template <bool STATS, class Nodes, class Prims, class Stack>
RT_HD void rt_instance_enter(const Nodes& nodes, const Prims& prims, v3 wo, v3 wd, RtTrav& tv, Stack& stk, RtCounters* wc) {
    int32_t ii = (int32_t)(~(uint32_t)tv.cur & 0x3FFFFFFFu);
    rt_f4 i0, i1;
    prims.instance((uint32_t)ii, i0, i1);
    if (tv.sp < stk.cap()) stk.put_sentinel(tv.sp++);
    tv.inst = ii;
    if (STATS) wc->instances++;
    rt_trav_ray(tv, rt_to_object_point(i0, i1, wo), rt_to_object_dir(i0, wd));
    tv.cur = nodes.instance_root(ii, f2i(i1.y));
}

// tv.cur is a leaf code (< 0): enter the instance, or intersect the leaf's primitives; then pop.
template <bool STATS, class Nodes, class Prims, class Stack>
RT_HD void rt_leaf_step(const Nodes& nodes, const Prims& prims, v3 wo, v3 wd, float time, float tmin, uint32_t src, RtTrav& tv, RtHit& hit,
                        Stack& stk, RtCounters* wc) {
    if (~(uint32_t)tv.cur & RT_LEAF_INSTANCE_BIT) { rt_instance_enter<STATS>(nodes, prims, wo, wd, tv, stk, wc); return; }
    rt_leaf_test<STATS>(prims, tv.cur, time, tmin, src, tv, hit, wc);
    tv.cur = rt_pop(tv, stk);
}

// Whole traversal of one ray.  On entry hit.t = tmax, hit.code = RT_HIT_NONE.  "while-while" organisation: the
// lane first descends inner nodes until it holds a leaf, then works on leaves, so that on a 64-wide wave the
// box-test code and the primitive code are not issued in turn for every step of every lane.
template <bool STATS, class Nodes, class Prims, class Stack>
RT_HD void rt_closest_hit_in(const RtSceneView& sc, const Nodes& nodes, const Prims& prims, v3 wo, v3 wd, float time, float tmin, uint32_t src,
                             Stack& stk, RtHit& hit, RtCounters* wc) {
    RtTrav tv;
    rt_trav_begin(sc, nodes, wo, wd, tv);
    while (tv.cur != RT_EMPTY_STACK) {
        while (rt_is_inner(tv.cur)) rt_inner_step<STATS>(nodes, tv, tmin, hit, stk, wc);
        if (tv.cur >= 0) {
            if (tv.cur != RT_EMPTY_STACK) rt_sentinel_step(wo, wd, tv, stk);
            continue;
        }
        do rt_leaf_step<STATS>(nodes, prims, wo, wd, time, tmin, src, tv, hit, stk, wc); while (tv.cur < 0);
    }
}
template <bool STATS, class Stack>
RT_HD void rt_closest_hit(const RtSceneView& sc, v3 wo, v3 wd, float time, float tmin, uint32_t src, Stack& stk, RtHit& hit,
                          RtCounters* wc) {
    RtGlobalNodes nodes; nodes.nodes = sc.nodes;
    RtGlobalPrims prims; prims.sc = &sc;
    rt_closest_hit_in<STATS>(sc, nodes, prims, wo, wd, time, tmin, src, stk, hit, wc);
}

// The same search through the 4-wide tree.
template <bool STATS, class Stack>
RT_HD void rt_closest_hit_wide(const RtSceneView& sc, v3 wo, v3 wd, float time, float tmin, uint32_t src, Stack& stk, RtHit& hit,
                               RtCounters* wc) {
    RtGlobalWideNodes nodes; nodes.nodes = sc.wnodes; nodes.inst_roots = sc.winst_roots;
    RtGlobalPrims prims; prims.sc = &sc;
    rt_closest_hit_in<STATS>(sc, nodes, prims, wo, wd, time, tmin, src, stk, hit, wc);
}

// ------------------------------------------------------------------ constant media (mediums.rs:39-87)
// Every lane tests the same medium at the same time, so all record indices are wave-uniform (rt_uniform) and the
// boundary data comes in through scalar loads.
//
// Closest hit of the boundary's primitives inside [tmin, tmax] (linear scan = objects.rs:67-78).
RT_HD float rt_boundary_hit(const RtSceneView& sc, rt_f4 m2, rt_f4 m3, rt_f4 m4, v3 o, v3 d, float time, float tmin, float tmax) {
    float best = tmax; bool any = false;
    const v3 inv = mk3(rt_rcp(d.x), rt_rcp(d.y), rt_rcp(d.z));
    int s0 = rt_uniform(f2i(m2.y)), sn = rt_uniform(f2i(m2.z)), q0 = rt_uniform(f2i(m2.w)), qn = rt_uniform(f2i(m3.x));
    int r0 = rt_uniform(f2i(m3.y)), rn = rt_uniform(f2i(m3.z));
    for (int i = 0; i < sn; ++i) {
        rt_f4 s = ld4(sc.spheres + s0 + i);
        float t = rt_hit_sphere(mk3(s.x, s.y, s.z), s.w, o, d, tmin, best, false);
        if (t == t) { best = t; any = true; }
    }
    for (int i = 0; i < qn; ++i) {
        const rt_f4* m = sc.mspheres + 3 * (size_t)(q0 + i);
        rt_f4 a = ld4(m), b = ld4(m + 1), c = ld4(m + 2);
        float t = rt_hit_sphere(rt_msphere_center(a, b, c, time), a.w, o, d, tmin, best, false);
        if (t == t) { best = t; any = true; }
    }
    for (int i = 0; i < rn; ++i) {
        rt_f4 a = ld4(sc.rects + 2 * (size_t)(r0 + i)), b = ld4(sc.rects + 2 * (size_t)(r0 + i) + 1);
        float t = rt_hit_rect(a, b.x, f2i(b.y), o, d, inv, tmin, best);
        if (t == t) { best = t; any = true; }
    }
    int x0 = rt_uniform(f2i(m4.x)), xn = rt_uniform(f2i(m4.y));
    for (int i = 0; i < xn; ++i) {
        rt_f4 a = ld4(sc.boxes + 2 * (size_t)(x0 + i)), b = ld4(sc.boxes + 2 * (size_t)(x0 + i) + 1);
        int face;
        float t = rt_hit_box(a, b, o, d, inv, tmin, best, -1, face);
        if (t == t) { best = t; any = true; }
    }
    return any ? best : rt_nan();
}

// The two smallest boundary crossings of the whole ray line, a <= b (+inf when absent), in one scan.  mediums.rs:46-47
// asks for try_hit(-inf, inf) and then try_hit(t1 + 0.0001, inf): the first is the smallest crossing and the second
// the smallest crossing that is >= t1 + 0.0001, which is b unless two crossings coincide (ray through an edge).
RT_HD void rt_boundary_pair(const RtSceneView& sc, rt_f4 m2, rt_f4 m3, rt_f4 m4, v3 o, v3 d, float time, float& a, float& b) {
    a = rt_inf(); b = rt_inf();
    const v3 inv = mk3(rt_rcp(d.x), rt_rcp(d.y), rt_rcp(d.z));
    int s0 = rt_uniform(f2i(m2.y)), sn = rt_uniform(f2i(m2.z)), q0 = rt_uniform(f2i(m2.w)), qn = rt_uniform(f2i(m3.x));
    int r0 = rt_uniform(f2i(m3.y)), rn = rt_uniform(f2i(m3.z));
    // insertion into the two smallest so far, without branches (written with an if / else-if the compiler addressed a and b
    // as a two-element array in scratch memory)
#define RT_INSERT(t_) do { float t__ = (t_); float hi__ = fmaxf(a, t__); a = fminf(a, t__); b = fminf(b, hi__); } while (0)
    for (int i = 0; i < sn; ++i) {
        rt_f4 s = ld4(sc.spheres + s0 + i);
        float tn, tf;
        if (rt_sphere_roots(mk3(s.x, s.y, s.z), s.w, o, d, tn, tf)) { RT_INSERT(tn); RT_INSERT(tf); }
    }
    for (int i = 0; i < qn; ++i) {
        const rt_f4* m = sc.mspheres + 3 * (size_t)(q0 + i);
        rt_f4 ma = ld4(m), mb = ld4(m + 1), mc = ld4(m + 2);
        float tn, tf;
        if (rt_sphere_roots(rt_msphere_center(ma, mb, mc, time), ma.w, o, d, tn, tf)) { RT_INSERT(tn); RT_INSERT(tf); }
    }
    for (int i = 0; i < rn; ++i) {
        rt_f4 ra = ld4(sc.rects + 2 * (size_t)(r0 + i)), rb = ld4(sc.rects + 2 * (size_t)(r0 + i) + 1);
        float t = rt_hit_rect(ra, rb.x, f2i(rb.y), o, d, inv, -rt_inf(), rt_inf());
        if (t == t) RT_INSERT(t);
    }
    int x0 = rt_uniform(f2i(m4.x)), xn = rt_uniform(f2i(m4.y));
    for (int i = 0; i < xn; ++i) {  // a box is crossed at its entry and its exit side
        rt_f4 ba = ld4(sc.boxes + 2 * (size_t)(x0 + i)), bb = ld4(sc.boxes + 2 * (size_t)(x0 + i) + 1);
        float t_in, t_out; int f_in, f_out;
        if (rt_box_slabs(ba, bb, o, d, inv, t_in, t_out, f_in, f_out)) { RT_INSERT(t_in); RT_INSERT(t_out); }
    }
#undef RT_INSERT
}

// Tests every medium against the segment [tmin, hit.t]; a free-flight collision that lands before
// the current closest hit replaces it.  With one fixed draw per (medium, segment) the outcome does
// not depend on the order in which the reference's BVH would have visited the medium.
// rnd0 = Philox block 0 of this bounce (its lanes 2, 3 are the draws of media 0, 1).
// MediaSrc::get(m, k) = record k of medium m (this one reads global memory; the indices are wave-uniform, so these are
// scalar loads).
struct RtGlobalMedia {
    const rt_f4* media;
    RT_HDM rt_f4 get(int m, int k) const { return ld4(media + RT_MEDIUM_F4 * (size_t)m + k); }
};
// The two crossings t1 <= t2 of the whole ray line with the boundary of medium m (mediums.rs:46-47); false when the
// ray misses the boundary or there is no second crossing.
template <class MediaSrc>
RT_HD bool rt_medium_span(const RtSceneView& sc, const MediaSrc& msrc, int m, v3 wo, v3 wd, float time, float& t1, float& t2) {
    rt_f4 m5 = msrc.get(m, 5);
    const int fast = rt_uniform(f2i(m5.x));
    v3 o = wo, d = wd;
    if (fast != 0) {
        // the boundary is one convex primitive: its two crossings are the entry and the exit (mediums.rs:46-47)
        if ((rt_uniform(f2i(m5.y)) & 1) != 0) {
            rt_f4 i0 = msrc.get(m, 8), i1 = msrc.get(m, 9);
            o = rt_to_object_point(i0, i1, wo); d = rt_to_object_dir(i0, wd);
        }
        rt_f4 p0 = msrc.get(m, 6);
        if (fast == 1) {
            if (!rt_sphere_roots(mk3(p0.x, p0.y, p0.z), p0.w, o, d, t1, t2)) return false;
        } else {
            rt_f4 p1 = msrc.get(m, 7);
            int f_in, f_out;
            if (!rt_box_slabs(p0, p1, o, d, mk3(rt_rcp(d.x), rt_rcp(d.y), rt_rcp(d.z)), t1, t2, f_in, f_out)) return false;
        }
        return !(t2 < t1 + fmaxf(0.0001f, 1e-6f * fabsf(t1)));                        // coincident crossings: no second one
    }
    rt_f4 m2 = msrc.get(m, 2), m3 = msrc.get(m, 3), m4 = msrc.get(m, 4);
    int ii = rt_uniform(f2i(m2.x));
    if (ii >= 0) {
        rt_f4 i0 = ld4(sc.instances + 2 * (size_t)ii), i1 = ld4(sc.instances + 2 * (size_t)ii + 1);
        o = rt_to_object_point(i0, i1, wo); d = rt_to_object_dir(i0, wd);
    }
    rt_boundary_pair(sc, m2, m3, m4, o, d, time, t1, t2);                           // mediums.rs:46-47
    if (!(t1 < rt_inf())) return false;
    // The re-query offset 0.0001 vanishes in fp32 once |t1| exceeds ~1000 (the r = 5000 fog), so it is widened
    // to a few ulps of t1 when that is larger.
    float t_next = t1 + fmaxf(0.0001f, 1e-6f * fabsf(t1));
    if (t2 < t_next) t2 = rt_boundary_hit(sc, m2, m3, m4, o, d, time, t_next, rt_inf());
    return t2 < rt_inf();                                                            // also false for NaN (no second crossing)
}

// Free flight through medium m along the ray, limited by t_limit (the closest hit so far): the ray parameter of the
// collision, or NaN when the ray does not collide in this medium.  `draw` = the medium's 32 random bits for this segment.
// from_scene: the ray starts on a surface or inside a medium of the scene (every ray but a camera ray).
template <class MediaSrc>
RT_HD float rt_medium_collision(const RtSceneView& sc, const MediaSrc& msrc, int m, v3 wo, v3 wd, v3 winv, float len, float time, float tmin,
                                float t_limit, uint32_t draw, bool from_scene) {
    rt_f4 m0 = msrc.get(m, 0), m1 = msrc.get(m, 1);
    float t1, t2;
    // A boundary that encloses the whole scene with room to spare (final_scene.yml's fog sphere of radius 5000 around a scene
    // 1000 across; flagged by the flattener): a ray that starts in the scene and ends on something in the scene crosses the
    // boundary behind its origin and beyond its end, so the clamps below would give t1 = tmin and t2 = t_limit whatever the
    // crossings are -- they need not be computed.
    if (from_scene && t_limit < rt_inf() && (rt_uniform(f2i(msrc.get(m, 5).y)) & 2) != 0) { t1 = tmin; t2 = t_limit; }
    else {
        if (!(rt_slab(m0.x, m0.y, m0.z, m1.x, m1.y, m1.z, wo, winv, tmin, t_limit) < rt_inf())) return rt_nan();
        if (!rt_medium_span(sc, msrc, m, wo, wd, time, t1, t2)) return rt_nan();
        if (t1 < tmin) t1 = tmin;
        if (t2 > t_limit) t2 = t_limit;
    }
    if (t1 >= t2) return rt_nan();
    if (t1 < 0.0f) t1 = 0.0f;
    float inside = (t2 - t1) * len;
    float hit_dist = m0.w * rt_log10(rt_u24(draw));                                 // mediums.rs:64 (log base 10)
    if (hit_dist > inside) return rt_nan();
    return t1 + hit_dist * rt_rcp(len);
}

template <class MediaSrc>
RT_HD void rt_media_hits(const RtSceneView& sc, const MediaSrc& msrc, v3 wo, v3 wd, float time, float tmin, uint32_t pixel, uint32_t sample,
                         uint32_t bounce, uint32_t key0, uint32_t key1, const uint32_t rnd0[4], RtHit& hit) {
    uint32_t r0 = rnd0[0], r1 = rnd0[1], r2 = rnd0[2], r3 = rnd0[3];  // scalars: an indexed array would live in scratch
    float len = rt_sqrt(dot3(wd, wd));
    v3 winv = mk3(rt_rcp(wd.x), rt_rcp(wd.y), rt_rcp(wd.z));
    for (int m = 0; m < sc.n_media; ++m) {
        int k = m + 2;
        if (k >= 4 && (k & 3) == 0) {
            uint32_t nr[4];
            rt_philox(pixel, sample, bounce, (uint32_t)(k >> 2), key0, key1, nr);
            r0 = nr[0]; r1 = nr[1]; r2 = nr[2]; r3 = nr[3];
        }
        int lane4 = k & 3;
        uint32_t draw = lane4 == 0 ? r0 : (lane4 == 1 ? r1 : (lane4 == 2 ? r2 : r3));
        float t = rt_medium_collision(sc, msrc, m, wo, wd, winv, len, time, tmin, hit.t, draw, bounce > 1u);
        if (t < hit.t) { hit.t = t; hit.code = RT_HIT_CODE(RT_KIND_MEDIUM, (uint32_t)m); hit.inst = -1; }   // false for NaN
    }
}
RT_HD void rt_media_hits(const RtSceneView& sc, v3 wo, v3 wd, float time, float tmin, uint32_t pixel, uint32_t sample,
                         uint32_t bounce, uint32_t key0, uint32_t key1, const uint32_t rnd0[4], RtHit& hit) {
    RtGlobalMedia g; g.media = sc.media;
    rt_media_hits(sc, g, wo, wd, time, tmin, pixel, sample, bounce, key0, key1, rnd0, hit);
}

// ------------------------------------------------------------------ hit record (objects.rs:126-152)
struct RtSurface { v3 p, n; int32_t mat, id; bool front; float u, v; };

RT_HD void rt_sphere_uv(v3 n, float* u, float* v) {  // objects.rs:179-193
    float theta = acosf(-n.y);
    float phi = atan2f(-n.z, n.x) + 3.14159265358979f;
    *u = phi * 0.15915494309189535f; *v = theta * 0.3183098861837907f;
}

// Tab: see RtGlobalTables.
template <class Tab>
RT_HD void rt_surface(const Tab& tab, v3 wo, v3 wd, float time, const RtHit& h, bool want_uv, RtSurface& s) {
    uint32_t kind = RT_HIT_KIND(h.code), idx = RT_HIT_INDEX(h.code);
    s.u = 0.0f; s.v = 0.0f;
    if (kind == RT_KIND_MEDIUM) {  // mediums.rs:69-80
        s.p = fma3(h.t, wd, wo);
        s.front = wd.x < 0.0f;
        s.n = mk3(s.front ? 1.0f : -1.0f, 0.0f, 0.0f);
        s.mat = f2i(tab.medium((int)idx, 1).w); s.id = f2i(tab.medium((int)idx, 3).w);
        return;
    }
    v3 o = wo, d = wd;
    rt_f4 i0, i1;
    if (h.inst >= 0) {
        tab.instance((uint32_t)h.inst, i0, i1);
        o = rt_to_object_point(i0, i1, wo); d = rt_to_object_dir(i0, wd);
    }
    v3 p = fma3(h.t, d, o), on;
    rt_i2 meta;
    if (kind == RT_KIND_SPHERE) {
        rt_f4 sp = tab.sphere(idx);
        on = normalize3(rt_rcp(sp.w) * (p - mk3(sp.x, sp.y, sp.z)));  // objects.rs:226
        meta = tab.sphere_meta(idx);
        if (want_uv) rt_sphere_uv(on, &s.u, &s.v);
    } else if (kind == RT_KIND_MSPHERE) {
        rt_f4 m0, m1, m2;
        tab.msphere(idx, m0, m1, m2);
        on = normalize3(rt_rcp(m0.w) * (p - rt_msphere_center(m0, m1, m2, time)));  // objects.rs:313
        meta = tab.msphere_meta(idx);
        if (want_uv) rt_sphere_uv(on, &s.u, &s.v);
    } else if (kind == RT_KIND_BOX) {  // side `idx & 7` of box `idx >> 3`, as the rectangle objects.rs:445-500 builds
        rt_f4 b0, b1;
        tab.box(idx >> 3, b0, b1);
        int face = (int)(idx & 7u), type = face >> 1;   // 0 = Xy, 1 = Xz, 2 = Yz
        on = type == 0 ? mk3(0, 0, 1) : (type == 1 ? mk3(0, 1, 0) : mk3(1, 0, 0));
        meta.x = f2i(b0.w); meta.y = f2i(b1.w) + face;
        if (want_uv) {
            float x = type == 2 ? p.y : p.x, y = type == 0 ? p.y : p.z;
            float xa = type == 2 ? b0.y : b0.x, xb = type == 2 ? b1.y : b1.x, ya = type == 0 ? b0.y : b0.z, yb = type == 0 ? b1.y : b1.z;
            s.u = (x - xa) * rt_rcp(xb - xa); s.v = (y - ya) * rt_rcp(yb - ya);
        }
    } else {
        rt_f4 r0, r1;
        tab.rect(idx, r0, r1);
        int type = f2i(r1.y);
        on = type == 0 ? mk3(0, 0, 1) : (type == 1 ? mk3(0, 1, 0) : mk3(1, 0, 0));  // objects.rs:385-389
        meta = tab.rect_meta(idx);
        if (want_uv) {
            float x = type == 2 ? p.y : p.x, y = type == 0 ? p.y : p.z;
            s.u = (x - r0.x) * rt_rcp(r0.y - r0.x); s.v = (y - r0.z) * rt_rcp(r0.w - r0.z);  // objects.rs:399-400
        }
    }
    bool front = dot3(d, on) < 0.0f;          // objects.rs:137
    v3 n = front ? on : -on;
    if (h.inst >= 0) {
        // Unwind the Translate / RotateY chain innermost first (transrot.rs:28-31, 104-115).  RotateY
        // rebuilds the record from the ray on its inner side and the rotated, already flipped normal.
        int first = f2i(i1.z), count = f2i(i1.w);
        for (int k = count - 1; k >= 0; --k) {
            rt_f4 op = tab.op(first + k);
            if (f2i(op.w) == RT_OP_TRANSLATE) { p = p + mk3(op.x, op.y, op.z); continue; }
            float c = op.x, sn = op.y;
            p = mk3(fmaf(c, p.x, sn * p.z), p.y, fmaf(-sn, p.x, c * p.z));
            v3 nw = mk3(fmaf(c, n.x, sn * n.z), n.y, fmaf(-sn, n.x, c * n.z));
            front = dot3(d, nw) < 0.0f;
            n = front ? nw : -nw;
            d = mk3(fmaf(c, d.x, sn * d.z), d.y, fmaf(-sn, d.x, c * d.z));
        }
    }
    s.p = p; s.n = n; s.front = front; s.mat = meta.x; s.id = meta.y;
}
RT_HD void rt_surface(const RtSceneView& sc, v3 wo, v3 wd, float time, const RtHit& h, bool want_uv, RtSurface& s) {
    RtGlobalTables g; g.sc = &sc;
    rt_surface(g, wo, wd, time, h, want_uv, s);
}

// ------------------------------------------------------------------ textures (textures.rs)
template <class Tab>
RT_HD float rt_perlin_noise(const Tab& tab, int table, v3 p) {  // textures.rs:146-170, 214-237
    float fx = floorf(p.x), fy = floorf(p.y), fz = floorf(p.z);
    float u = p.x - fx, v = p.y - fy, w = p.z - fz;
    int i = (int)fx, j = (int)fy, k = (int)fz;
    float uu = u * u * (3.0f - 2.0f * u), vv = v * v * (3.0f - 2.0f * v), ww = w * w * (3.0f - 2.0f * w);
    float accum = 0.0f;
#pragma unroll
    for (int di = 0; di < 2; ++di)
#pragma unroll
        for (int dj = 0; dj < 2; ++dj)
#pragma unroll
            for (int dk = 0; dk < 2; ++dk) {
                int h = tab.perlin_perm(table, (i + di) & 255) ^ tab.perlin_perm(table, 256 + ((j + dj) & 255)) ^
                        tab.perlin_perm(table, 512 + ((k + dk) & 255));
                rt_f4 c = tab.perlin_vec(table, h);
                float wgt = (di ? uu : 1.0f - uu) * (dj ? vv : 1.0f - vv) * (dk ? ww : 1.0f - ww);
                accum = fmaf(wgt, fmaf(c.x, u - (float)di, fmaf(c.y, v - (float)dj, c.z * (w - (float)dk))), accum);
            }
    return accum;
}
template <class Tab>
RT_HD float rt_turbulence(const Tab& tab, int table, v3 p) {  // textures.rs:131-143
    float accum = 0.0f, weight = 1.0f;
    for (int i = 0; i < 7; ++i) { accum = fmaf(weight, rt_perlin_noise(tab, table, p), accum); weight *= 0.5f; p = 2.0f * p; }
    return fabsf(accum);
}
// `inline_rgb` is the colour baked into the material record when its texture is a SolidColor (tex < 0).
template <class Tab>
RT_HD v3 rt_texture_value(const Tab& tab, int tex, v3 inline_rgb, v3 p) {
    if (tex < 0) return inline_rgb;
    for (int it = 0; it < 16; ++it) {
        rt_f4 t0, t1;
        tab.texture(tex, t0, t1);
        int type = f2i(t0.x);
        if (type == RT_TEX_CHECKER) {  // textures.rs:60-69
            float sines = sinf(10.0f * p.x) * sinf(10.0f * p.y) * sinf(10.0f * p.z);
            tex = sines < 0.0f ? f2i(t0.z) : f2i(t0.y);
            continue;
        }
        if (type == RT_TEX_NOISE) {  // textures.rs:87-96
            int table = f2i(t0.y);
            float turb = rt_turbulence(tab, table, t1.x * p);
            float s = 0.5f * (1.0f + sinf(fmaf(10.0f, turb, t1.x * p.z)));
            return mk3(s, s, s);
        }
        return mk3(t1.x, t1.y, t1.z);  // textures.rs:28-32
    }
    return mk3(0, 0, 0);
}

// ------------------------------------------------------------------ materials (materials.rs)
RT_HD v3 rt_reflect(v3 v, v3 n) { return fma3(-2.0f * dot3(v, n), n, v); }  // utils.rs:87-89

// One scatter event (lib.rs:64-65).  u1, u2: the bounce's two draws (unit vector; a dielectric decides on u1).
// Returns true when the path continues along `dir` with throughput *= atten.
template <class Tab>
RT_HD bool rt_scatter(const Tab& tab, const RtSurface& s, v3 wd, float u1, float u2,
                      v3& atten, v3& dir, v3& emitted) {
    rt_f4 a0, a1;
    tab.material(s.mat, a0, a1);
    int type = f2i(a0.x), tex = f2i(a0.y);
    v3 rgb = mk3(a1.x, a1.y, a1.z);
    emitted = mk3(0, 0, 0);
    atten = mk3(1, 1, 1);
    dir = s.n;
    if (type == RT_MAT_LAMBERTIAN) {  // materials.rs:77-90
        dir = s.n + rt_unit_vector(u1, u2);
        if (dot3(dir, dir) < 1e-16f) dir = s.n;
        atten = rt_texture_value(tab, tex, rgb, s.p);
        return true;
    }
    if (type == RT_MAT_METAL) {  // materials.rs:129-145
        dir = fma3(a0.z, rt_unit_vector(u1, u2), rt_reflect(normalize3(wd), s.n));
        atten = rgb;
        return dot3(dir, s.n) > 0.0f;
    }
    if (type == RT_MAT_DIELECTRIC) {  // materials.rs:181-209
        float ratio = s.front ? rt_rcp(a0.w) : a0.w;
        v3 ud = normalize3(wd);
        float cos_theta = -fminf(dot3(ud, s.n), 1.0f);
        float sin_theta = rt_sqrt(fmaxf(0.0f, 1.0f - cos_theta * cos_theta));
        bool cannot = ratio * sin_theta > 1.0f;
        float r0 = (1.0f - ratio) * rt_rcp(1.0f + ratio); r0 *= r0;  // materials.rs:174-178
        float om = 1.0f - cos_theta, om2 = om * om;
        float refl = fmaf(1.0f - r0, om2 * om2 * om, r0);
        if (cannot || refl > u1) dir = rt_reflect(ud, s.n);
        else {  // utils.rs:79-84
            v3 perp = ratio * fma3(cos_theta, s.n, ud);
            float par = -rt_sqrt(fabsf(1.0f - dot3(perp, perp)));
            dir = fma3(par, s.n, perp);
        }
        return true;
    }
    if (type == RT_MAT_LIGHT) {  // materials.rs:243-249
        emitted = rt_texture_value(tab, tex, rgb, s.p);
        return false;
    }
    // Isotropic, materials.rs:276-283
    dir = rt_unit_vector(u1, u2);
    atten = rt_texture_value(tab, tex, rgb, s.p);
    return true;
}
RT_HD bool rt_scatter(const RtSceneView& sc, const RtSurface& s, v3 wd, float u1, float u2, v3& atten, v3& dir, v3& emitted) {
    RtGlobalTables g; g.sc = &sc;
    return rt_scatter(g, s, wd, u1, u2, atten, dir, emitted);
}

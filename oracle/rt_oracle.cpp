// rt_oracle.cpp -- TEST INFRASTRUCTURE ONLY.  Never linked into, loaded by, or called from the
// product library (librt_hip.so) or the ray_tracing_amd package.  Only tests/, the smoke check and
// the cpu_baseline leg of the benchmark may load it, and only as the checker / reported baseline.
//
// What it is: an f64 CPU restatement of the per-pixel render path of mkagie/ray_tracing, following
// the structure of the reference (pointer tree of virtual Hittables, recursive random-axis
// median-split BVH, recursive get_color, per-candidate hit records).  Every function cites the
// reference file:line it follows.
//
// PARITY UNPINNED: the reference ships no tests, golden images or fixtures, and it cannot be built
// in this environment (Rust toolchain and crates absent).  This restatement is pinned only by
// known-answer vectors derived by hand from the cited formulas (tests/test_oracle_kat.py).
//
// Deliberate, documented departures from the reference (all distribution-preserving):
//  * thread_rng (OS-seeded ChaCha12) is replaced by counter-based Philox4x32-10 keyed by a render
//    seed with counter (pixel, sample, bounce, block) for path sampling, and by a seeded
//    SplitMix64 stream for scene construction, so that renders are reproducible and can be
//    compared path-by-path with the GPU implementation.
//  * the rejection loops (utils.rs:13-22 unit vector, cameras.rs:103-113 unit disk) are replaced by
//    direct maps with the same distributions and a fixed number of draws.
//  * a negative-radius sphere's bounding box uses |r| (objects.rs:241-244 builds an inverted box
//    that the BVH can never hit when the sphere is alone in a leaf).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace orc {

typedef uint32_t u32;
typedef uint64_t u64;
static const double kPi = 3.14159265358979323846;
static const double kInf = std::numeric_limits<double>::infinity();

// ---------------------------------------------------------------- vectors (lib.rs:17-19)
struct V3 {
    double x, y, z;
    V3() : x(0), y(0), z(0) {}
    V3(double a, double b, double c) : x(a), y(b), z(c) {}
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    double& at(int i) { return i == 0 ? x : (i == 1 ? y : z); }
};
static inline V3 operator+(V3 a, V3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator-(V3 a) { return V3(-a.x, -a.y, -a.z); }
static inline V3 operator*(double s, V3 a) { return V3(s * a.x, s * a.y, s * a.z); }
static inline V3 operator*(V3 a, double s) { return V3(s * a.x, s * a.y, s * a.z); }
static inline V3 operator/(V3 a, double s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline V3 cmul(V3 a, V3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline V3 cross(V3 a, V3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline double norm(V3 a) { return std::sqrt(dot(a, a)); }
static inline V3 normalize(V3 a) { return a / norm(a); }
static inline double sq(double v) { return v * v; }

// ---------------------------------------------------------------- RNG
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11).
static inline void philox4x32_10(const u32 ctr[4], const u32 key[2], u32 out[4]) {
    const u32 M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
    u32 c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        if (r) { k0 += W0; k1 += W1; }
        u64 p0 = (u64)M0 * c0, p1 = (u64)M1 * c2;
        u32 hi0 = (u32)(p0 >> 32), lo0 = (u32)p0, hi1 = (u32)(p1 >> 32), lo1 = (u32)p1;
        u32 n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Per-path draw schedule: dimension `dim` of bounce slot `bounce` for (pixel, sample).
//   bounce 0 (camera): dim0,1 pixel jitter (main.rs:572-573); dim2,3 lens (cameras.rs:83);
//                      dim4 shutter time (cameras.rs:86-91)
//   bounce b>=1      : dim0,1 unit vector (utils.rs:13-22); dim0 also decides reflect / refract in a dielectric
//                      (materials.rs:197; no material uses both); dim 2+m the free-flight draw of medium m (mediums.rs:64)
// Value = top 24 bits * 2^-24 so that f32 and f64 consumers see the identical number.
struct Stats;
struct Ctx {
    u32 pixel, sample;
    u32 key[2];
    int bounce;
    Stats* st;
    double xi(int dim) const {
        u32 ctr[4] = {pixel, sample, (u32)bounce, (u32)(dim >> 2)}, out[4];
        philox4x32_10(ctr, key, out);
        return (double)(out[dim & 3] >> 8) * (1.0 / 16777216.0);
    }
};

// Scene-construction stream replacing thread_rng in main.rs:32,338, textures.rs:112-119,187.
struct SceneRng {
    u64 s;
    explicit SceneRng(u64 seed) : s(seed) {}
    u64 next() {
        u64 z = (s += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double gen() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }  // rng.gen::<f64>()
    double range(double a, double b) { return a + (b - a) * gen(); }            // gen_range(a..b)
    u32 upto(u32 i) { return (u32)(next() % (u64)(i + 1)); }                    // gen_range(0..=i)
};

struct Stats {
    u64 rays = 0, box = 0, sph = 0, msph = 0, rect = 0, inst = 0, med = 0, perlin = 0;
    u64 shade[5] = {0, 0, 0, 0, 0};
    void add(const Stats& o) {
        rays += o.rays; box += o.box; sph += o.sph; msph += o.msph; rect += o.rect;
        inst += o.inst; med += o.med; perlin += o.perlin;
        for (int i = 0; i < 5; ++i) shade[i] += o.shade[i];
    }
};

// utils.rs:13-22 -- uniform unit vector (the reference normalises its in-ball sample).
static inline V3 unit_vector(double u1, double u2) {
    double z = 1.0 - 2.0 * u1;
    double r = std::sqrt(std::max(0.0, 1.0 - z * z));
    double phi = 2.0 * kPi * u2;
    return V3(r * std::cos(phi), r * std::sin(phi), z);
}
// cameras.rs:103-113 -- uniform point in the unit disk.
static inline void unit_disk(double u1, double u2, double& dx, double& dy) {
    double r = std::sqrt(u1), phi = 2.0 * kPi * u2;
    dx = r * std::cos(phi); dy = r * std::sin(phi);
}

// ---------------------------------------------------------------- Ray (lib.rs:39-52)
struct Ray {
    V3 o, d; double time;
    Ray() : time(0) {}
    Ray(V3 o_, V3 d_, double t_) : o(o_), d(d_), time(t_) {}
    V3 at(double t) const { return o + t * d; }
};

// ---------------------------------------------------------------- textures (textures.rs)
struct Perlin {  // textures.rs:99-238
    V3 ranvec[256]; int perm[3][256];
    void generate(SceneRng& rng) {  // textures.rs:110-128, 175-192
        for (int i = 0; i < 256; ++i) {
            double a = rng.range(-1, 1), b = rng.range(-1, 1), c = rng.range(-1, 1);
            ranvec[i] = V3(a, b, c);
        }
        for (int k = 0; k < 3; ++k) {
            for (int i = 0; i < 256; ++i) perm[k][i] = i;
            for (int i = 255; i >= 1; --i) { u32 t = rng.upto((u32)i); std::swap(perm[k][i], perm[k][t]); }
        }
    }
    double noise(V3 p, Stats* st) const {  // textures.rs:146-170, 214-237
        if (st) st->perlin++;
        double fx = std::floor(p.x), fy = std::floor(p.y), fz = std::floor(p.z);
        double u = p.x - fx, v = p.y - fy, w = p.z - fz;
        int i = (int)fx, j = (int)fy, k = (int)fz;
        double uu = u * u * (3 - 2 * u), vv = v * v * (3 - 2 * v), ww = w * w * (3 - 2 * w);
        double accum = 0;
        for (int di = 0; di < 2; ++di) for (int dj = 0; dj < 2; ++dj) for (int dk = 0; dk < 2; ++dk) {
            V3 c = ranvec[perm[0][(i + di) & 255] ^ perm[1][(j + dj) & 255] ^ perm[2][(k + dk) & 255]];
            V3 wv(u - di, v - dj, w - dk);
            accum += (di * uu + (1 - di) * (1 - uu)) * (dj * vv + (1 - dj) * (1 - vv)) *
                     (dk * ww + (1 - dk) * (1 - ww)) * dot(c, wv);
        }
        return accum;
    }
    double turbulence(V3 p, Stats* st) const {  // textures.rs:131-143 (depth 7)
        double accum = 0, weight = 1; V3 tp = p;
        for (int i = 0; i < 7; ++i) { accum += weight * noise(tp, st); weight *= 0.5; tp = 2.0 * tp; }
        return std::fabs(accum);
    }
};

struct Texture {
    virtual ~Texture() {}
    virtual V3 value(double u, double v, V3 p, Stats* st) const = 0;  // textures.rs:12-15
};
struct SolidColor : Texture {  // textures.rs:19-32
    V3 c; explicit SolidColor(V3 c_) : c(c_) {}
    V3 value(double, double, V3, Stats*) const override { return c; }
};
struct Checker : Texture {  // textures.rs:35-69
    const Texture *even, *odd;
    Checker(const Texture* e, const Texture* o) : even(e), odd(o) {}
    V3 value(double u, double v, V3 p, Stats* st) const override {
        double sines = std::sin(10 * p.x) * std::sin(10 * p.y) * std::sin(10 * p.z);
        return sines < 0 ? odd->value(u, v, p, st) : even->value(u, v, p, st);
    }
};
struct Noise : Texture {  // textures.rs:73-96
    std::shared_ptr<Perlin> perlin; double scale;
    V3 value(double, double, V3 p, Stats* st) const override {
        double s = 0.5 * (1.0 + std::sin(scale * p.z + 10.0 * perlin->turbulence(scale * p, st)));
        return V3(s, s, s);
    }
};

// ---------------------------------------------------------------- hit record / traits
struct Material;
struct HitRecord {  // objects.rs:110-152
    V3 p, normal; double t; bool front_face; const Material* mat; double u, v; int id;
    static HitRecord make(V3 p, double t, const Ray& ray, V3 outward, const Material* m, double u, double v, int id) {
        HitRecord h; h.p = p; h.t = t; h.mat = m; h.u = u; h.v = v; h.id = id;
        h.front_face = dot(ray.d, outward) < 0.0;       // objects.rs:137
        h.normal = h.front_face ? outward : -outward;   // objects.rs:138-141
        return h;
    }
};

struct Aabb {  // bvh.rs:15-77
    V3 mn, mx;
    bool hit(const Ray& r, double tmin, double tmax) const {  // bvh.rs:46-61
        for (int a = 0; a < 3; ++a) {
            double inv = 1.0 / r.d[a];
            double t0 = (mn[a] - r.o[a]) * inv, t1 = (mx[a] - r.o[a]) * inv;
            if (inv < 0.0) std::swap(t0, t1);
            tmin = t0 > tmin ? t0 : tmin;
            tmax = t1 < tmax ? t1 : tmax;
            if (tmax <= tmin) return false;
        }
        return true;
    }
    static Aabb surrounding(const Aabb& a, const Aabb& b) {  // bvh.rs:64-76
        Aabb r;
        r.mn = V3(std::min(a.mn.x, b.mn.x), std::min(a.mn.y, b.mn.y), std::min(a.mn.z, b.mn.z));
        r.mx = V3(std::max(a.mx.x, b.mx.x), std::max(a.mx.y, b.mx.y), std::max(a.mx.z, b.mx.z));
        return r;
    }
};

struct Hittable {  // objects.rs:16-22
    virtual ~Hittable() {}
    virtual bool try_hit(const Ray& r, double tmin, double tmax, const Ctx& cx, HitRecord& rec) const = 0;
    virtual bool bbox(double t0, double t1, Aabb& out) const = 0;
};

struct Material {  // materials.rs:14-22
    int kind;  // 0 lambertian 1 metal 2 dielectric 3 light 4 isotropic (stats only)
    virtual ~Material() {}
    virtual bool scatter(const Ray& in, const HitRecord& h, const Ctx& cx, V3& atten, Ray& out) const = 0;
    virtual V3 emitted(double, double, V3, Stats*) const { return V3(0, 0, 0); }
};

static inline V3 reflect(V3 v, V3 n) { return v - 2.0 * dot(v, n) * n; }  // utils.rs:87-89
static inline V3 refract(V3 uv, V3 n, double eta) {                       // utils.rs:79-84
    double cos_theta = -std::min(dot(uv, n), 1.0);
    V3 perp = eta * (uv + cos_theta * n);
    V3 par = -std::sqrt(std::fabs(1.0 - sq(norm(perp)))) * n;
    return perp + par;
}

struct Lambertian : Material {  // materials.rs:55-98
    const Texture* albedo;
    explicit Lambertian(const Texture* t) : albedo(t) { kind = 0; }
    bool scatter(const Ray& in, const HitRecord& h, const Ctx& cx, V3& atten, Ray& out) const override {
        V3 dir = h.normal + unit_vector(cx.xi(0), cx.xi(1));
        if (norm(dir) < 1e-8) dir = h.normal;
        out = Ray(h.p, dir, in.time);
        atten = albedo->value(h.u, h.v, h.p, cx.st);
        return true;
    }
};
struct Metal : Material {  // materials.rs:108-146
    V3 albedo; double fuzz;
    Metal(V3 a, double f) : albedo(a), fuzz(f) { kind = 1; }
    bool scatter(const Ray& in, const HitRecord& h, const Ctx& cx, V3& atten, Ray& out) const override {
        V3 refl = reflect(normalize(in.d), h.normal);
        out = Ray(h.p, refl + fuzz * unit_vector(cx.xi(0), cx.xi(1)), in.time);
        atten = albedo;
        return dot(out.d, h.normal) > 0.0;
    }
};
struct Dielectric : Material {  // materials.rs:158-210
    double ir;
    explicit Dielectric(double i) : ir(i) { kind = 2; }
    static double reflectance(double cosine, double ref_idx) {  // materials.rs:174-178
        double r0 = sq((1 - ref_idx) / (1 + ref_idx));
        return r0 + (1 - r0) * std::pow(1 - cosine, 5);
    }
    bool scatter(const Ray& in, const HitRecord& h, const Ctx& cx, V3& atten, Ray& out) const override {
        atten = V3(1, 1, 1);
        double ratio = h.front_face ? 1.0 / ir : ir;
        V3 ud = normalize(in.d);
        double cos_theta = -std::min(dot(ud, h.normal), 1.0);  // materials.rs:190 (min binds before unary minus)
        double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
        bool cannot = ratio * sin_theta > 1.0;
        V3 dir = (cannot || reflectance(cos_theta, ratio) > cx.xi(0)) ? reflect(ud, h.normal)
                                                                      : refract(ud, h.normal, ratio);
        out = Ray(h.p, dir, in.time);
        return true;
    }
};
struct DiffuseLight : Material {  // materials.rs:220-250
    const Texture* emit;
    explicit DiffuseLight(const Texture* t) : emit(t) { kind = 3; }
    bool scatter(const Ray&, const HitRecord&, const Ctx&, V3&, Ray&) const override { return false; }
    V3 emitted(double u, double v, V3 p, Stats* st) const override { return emit->value(u, v, p, st); }
};
struct Isotropic : Material {  // materials.rs:253-284
    const Texture* albedo;
    explicit Isotropic(const Texture* t) : albedo(t) { kind = 4; }
    bool scatter(const Ray& in, const HitRecord& h, const Ctx& cx, V3& atten, Ray& out) const override {
        out = Ray(h.p, unit_vector(cx.xi(0), cx.xi(1)), in.time);
        atten = albedo->value(h.u, h.v, h.p, cx.st);
        return true;
    }
};

// ---------------------------------------------------------------- objects (objects.rs)
static inline void sphere_uv(V3 p, double& u, double& v) {  // objects.rs:179-193
    double theta = std::acos(-p.y);
    double phi = std::atan2(-p.z, p.x) + kPi;
    u = phi / (2 * kPi); v = theta / kPi;
}

struct HittableList : Hittable {  // objects.rs:24-100
    std::vector<const Hittable*> items;
    bool try_hit(const Ray& r, double tmin, double tmax, const Ctx& cx, HitRecord& rec) const override {
        double closest = tmax; bool any = false; HitRecord tmp;
        for (const Hittable* o : items)
            if (o->try_hit(r, tmin, closest, cx, tmp)) { closest = tmp.t; rec = tmp; any = true; }
        return any;
    }
    bool bbox(double t0, double t1, Aabb& out) const override {
        if (items.empty()) return false;
        bool first = true; Aabb tmp;
        for (const Hittable* o : items) {
            if (!o->bbox(t0, t1, tmp)) return false;
            out = first ? tmp : Aabb::surrounding(out, tmp);
            first = false;
        }
        return true;
    }
};

struct Sphere : Hittable {  // objects.rs:155-245
    V3 c; double r; const Material* m; int id;
    bool try_hit(const Ray& ray, double tmin, double tmax, const Ctx& cx, HitRecord& rec) const override {
        if (cx.st) cx.st->sph++;
        V3 oc = ray.o - c;
        double a = sq(norm(ray.d));
        double half_b = dot(oc, ray.d);
        double cc = sq(norm(oc)) - r * r;
        double disc = half_b * half_b - a * cc;
        if (disc < 0.0) return false;
        double sqrtd = std::sqrt(disc);
        double root = (-half_b - sqrtd) / a;
        if (root < tmin || tmax < root) {
            root = (-half_b + sqrtd) / a;
            if (root < tmin || tmax < root) return false;
        }
        V3 p = ray.at(root);
        V3 on = normalize((p - c) / r);
        double u, v; sphere_uv(on, u, v);
        rec = HitRecord::make(p, root, ray, on, m, u, v, id);
        return true;
    }
    bool bbox(double, double, Aabb& out) const override {
        double ar = std::fabs(r);  // departure: |r| (see header)
        out.mn = c - V3(ar, ar, ar); out.mx = c + V3(ar, ar, ar);
        return true;
    }
};

struct MovingSphere : Hittable {  // objects.rs:257-336
    V3 c0, c1; double t0, t1, r; const Material* m; int id;
    V3 center(double time) const { return c0 + ((time - t0) / (t1 - t0)) * (c1 - c0); }  // objects.rs:285-289
    bool try_hit(const Ray& ray, double tmin, double tmax, const Ctx& cx, HitRecord& rec) const override {
        if (cx.st) cx.st->msph++;
        V3 cen = center(ray.time);
        V3 oc = ray.o - cen;
        double a = sq(norm(ray.d));
        double half_b = dot(oc, ray.d);
        double cc = sq(norm(oc)) - r * r;
        double disc = half_b * half_b - a * cc;
        if (disc < 0.0) return false;
        double sqrtd = std::sqrt(disc);
        double root = (-half_b - sqrtd) / a;
        if (root < tmin || tmax < root) {
            root = (-half_b + sqrtd) / a;
            if (root < tmin || tmax < root) return false;
        }
        V3 p = ray.at(root);
        V3 on = normalize((p - cen) / r);
        double u, v; sphere_uv(on, u, v);
        rec = HitRecord::make(p, root, ray, on, m, u, v, id);
        return true;
    }
    bool bbox(double ta, double tb, Aabb& out) const override {
        double ar = std::fabs(r); V3 v(ar, ar, ar);
        Aabb b0{center(ta) - v, center(ta) + v}, b1{center(tb) - v, center(tb) + v};
        out = Aabb::surrounding(b0, b1);
        return true;
    }
};

struct Rectangle : Hittable {  // objects.rs:339-430;  type 0 = Xy, 1 = Xz, 2 = Yz
    const Material* m; double x0, x1, y0, y1, k; int type; int id;
    bool try_hit(const Ray& ray, double tmin, double tmax, const Ctx& cx, HitRecord& rec) const override {
        if (cx.st) cx.st->rect++;
        int oa, xa, ya; V3 on;
        if (type == 0) { oa = 2; xa = 0; ya = 1; on = V3(0, 0, 1); }
        else if (type == 1) { oa = 1; xa = 0; ya = 2; on = V3(0, 1, 0); }
        else { oa = 0; xa = 1; ya = 2; on = V3(1, 0, 0); }
        double t = (k - ray.o[oa]) / ray.d[oa];
        if (t < tmin || t > tmax) return false;
        double x = ray.o[xa] + t * ray.d[xa], y = ray.o[ya] + t * ray.d[ya];
        if (x < x0 || x > x1 || y < y0 || y > y1) return false;
        double u = (x - x0) / (x1 - x0), v = (y - y0) / (y1 - y0);
        rec = HitRecord::make(ray.at(t), t, ray, on, m, u, v, id);
        return true;
    }
    bool bbox(double, double, Aabb& out) const override {  // objects.rs:413-429
        const double off = 0.0001;
        if (type == 0) { out.mn = V3(x0, y0, k - off); out.mx = V3(x1, y1, k + off); }
        else if (type == 1) { out.mn = V3(x0, k - off, y0); out.mx = V3(x1, k + off, y1); }
        else { out.mn = V3(k - off, x0, y0); out.mx = V3(k + off, x1, y1); }
        return true;
    }
};

struct BoxObj : Hittable {  // objects.rs:433-517
    V3 mn, mx; HittableList sides;
    bool try_hit(const Ray& r, double tmin, double tmax, const Ctx& cx, HitRecord& rec) const override {
        return sides.try_hit(r, tmin, tmax, cx, rec);
    }
    bool bbox(double, double, Aabb& out) const override { out.mn = mn; out.mx = mx; return true; }
};

struct Translate : Hittable {  // transrot.rs:10-39
    const Hittable* obj; V3 off;
    bool try_hit(const Ray& ray, double tmin, double tmax, const Ctx& cx, HitRecord& rec) const override {
        if (cx.st) cx.st->inst++;
        Ray moved(ray.o - off, ray.d, ray.time);
        if (!obj->try_hit(moved, tmin, tmax, cx, rec)) return false;
        rec.p = rec.p + off;  // normal / front_face are left as computed against the moved ray
        return true;
    }
    bool bbox(double t0, double t1, Aabb& out) const override {
        if (!obj->bbox(t0, t1, out)) return false;
        out.mn = out.mn + off; out.mx = out.mx + off;
        return true;
    }
};

struct RotateY : Hittable {  // transrot.rs:42-121
    const Hittable* obj; double sin_t, cos_t; Aabb box; bool has_box;
    void init(const Hittable* o, double deg) {  // transrot.rs:53-89
        obj = o;
        double rad = deg * (kPi / 180.0);
        sin_t = std::sin(rad); cos_t = std::cos(rad);
        Aabb b; has_box = obj->bbox(0.0, 1.0, b);
        V3 mn(kInf, kInf, kInf), mx(-kInf, -kInf, -kInf);
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int k = 0; k < 2; ++k) {
            double x = i * b.mx.x + (1 - i) * b.mn.x, y = j * b.mx.y + (1 - j) * b.mn.y, z = k * b.mx.z + (1 - k) * b.mn.z;
            double nx = cos_t * x + sin_t * z, nz = -sin_t * x + cos_t * z;
            V3 t(nx, y, nz);
            for (int c = 0; c < 3; ++c) { mn.at(c) = std::min(mn[c], t[c]); mx.at(c) = std::max(mx[c], t[c]); }
        }
        box.mn = mn; box.mx = mx;
    }
    bool try_hit(const Ray& ray, double tmin, double tmax, const Ctx& cx, HitRecord& rec) const override {
        if (cx.st) cx.st->inst++;
        V3 o = ray.o, d = ray.d;
        o.x = cos_t * ray.o.x - sin_t * ray.o.z; o.z = sin_t * ray.o.x + cos_t * ray.o.z;
        d.x = cos_t * ray.d.x - sin_t * ray.d.z; d.z = sin_t * ray.d.x + cos_t * ray.d.z;
        Ray rr(o, d, ray.time);
        HitRecord in;
        if (!obj->try_hit(rr, tmin, tmax, cx, in)) return false;
        V3 p = in.p, n = in.normal;
        p.x = cos_t * in.p.x + sin_t * in.p.z; p.z = -sin_t * in.p.x + cos_t * in.p.z;
        n.x = cos_t * in.normal.x + sin_t * in.normal.z; n.z = -sin_t * in.normal.x + cos_t * in.normal.z;
        // transrot.rs:114 -- the record is rebuilt against the OBJECT-space ray with the
        // world-space, already-flipped normal.
        rec = HitRecord::make(p, in.t, rr, n, in.mat, in.u, in.v, in.id);
        return true;
    }
    bool bbox(double, double, Aabb& out) const override { out = box; return has_box; }
};

struct ConstantMedium : Hittable {  // mediums.rs:14-92
    const Hittable* boundary; const Material* phase; double neg_inv_density; int id; int slot;
    bool try_hit(const Ray& ray, double tmin, double tmax, const Ctx& cx, HitRecord& rec) const override {
        if (cx.st) cx.st->med++;
        HitRecord r1, r2;
        if (!boundary->try_hit(ray, -kInf, kInf, cx, r1)) return false;
        if (!boundary->try_hit(ray, r1.t + 0.0001, kInf, cx, r2)) return false;
        if (r1.t < tmin) r1.t = tmin;
        if (r2.t > tmax) r2.t = tmax;
        if (r1.t >= r2.t) return false;
        if (r1.t < 0.0) r1.t = 0.0;
        double len = norm(ray.d);
        double inside = (r2.t - r1.t) * len;
        double hit_dist = neg_inv_density * std::log10(cx.xi(2 + slot));  // mediums.rs:64 (log base 10)
        if (hit_dist > inside) return false;
        double t = r1.t + hit_dist / len;
        rec = HitRecord::make(ray.at(t), t, ray, V3(1, 0, 0), phase, 0.0, 0.0, id);
        return true;
    }
    bool bbox(double t0, double t1, Aabb& out) const override { return boundary->bbox(t0, t1, out); }
};

struct BvhNode : Hittable {  // bvh.rs:82-192
    const Hittable* left; const Hittable* right; Aabb box;
    bool try_hit(const Ray& r, double tmin, double tmax, const Ctx& cx, HitRecord& rec) const override {
        if (cx.st) cx.st->box++;
        if (!box.hit(r, tmin, tmax)) return false;
        if (left->try_hit(r, tmin, tmax, cx, rec)) {
            if (right) { HitRecord rr; if (right->try_hit(r, tmin, rec.t, cx, rr)) rec = rr; }
            return true;
        }
        return right ? right->try_hit(r, tmin, tmax, cx, rec) : false;
    }
    bool bbox(double, double, Aabb& out) const override { out = box; return true; }
};

// ---------------------------------------------------------------- scene container
struct Scene {
    std::vector<std::unique_ptr<Texture>> texs;
    std::vector<std::unique_ptr<Material>> mats;
    std::vector<std::unique_ptr<Hittable>> objs;
    const Hittable* world = nullptr;
    int next_id = 0, n_media = 0;
    SceneRng bvh_rng{0};
    std::string err;

    int add_tex(Texture* t) { texs.emplace_back(t); return (int)texs.size() - 1; }
    int add_mat(Material* m) { mats.emplace_back(m); return (int)mats.size() - 1; }
    int add_obj(Hittable* o) { objs.emplace_back(o); return (int)objs.size() - 1; }

    int solid(V3 c) { return add_tex(new SolidColor(c)); }
    int checker(int even, int odd) { return add_tex(new Checker(texs[even].get(), texs[odd].get())); }
    int noise(double scale, std::shared_ptr<Perlin> p) { Noise* n = new Noise; n->perlin = p; n->scale = scale; return add_tex(n); }
    int lambertian(int tex) { return add_mat(new Lambertian(texs[tex].get())); }
    int metal(V3 a, double f) { return add_mat(new Metal(a, f)); }
    int dielectric(double ir) { return add_mat(new Dielectric(ir)); }
    int light(int tex) { return add_mat(new DiffuseLight(texs[tex].get())); }
    int isotropic(int tex) { return add_mat(new Isotropic(texs[tex].get())); }

    int sphere(V3 c, double r, int mat) {
        Sphere* s = new Sphere; s->c = c; s->r = r; s->m = mats[mat].get(); s->id = next_id++; return add_obj(s);
    }
    int moving_sphere(V3 c0, V3 c1, double t0, double t1, double r, int mat) {
        MovingSphere* s = new MovingSphere; s->c0 = c0; s->c1 = c1; s->t0 = t0; s->t1 = t1; s->r = r;
        s->m = mats[mat].get(); s->id = next_id++; return add_obj(s);
    }
    Rectangle* make_rect(int mat, double x0, double x1, double y0, double y1, double k, int type) {
        Rectangle* r = new Rectangle; r->m = mats[mat].get(); r->x0 = x0; r->x1 = x1; r->y0 = y0; r->y1 = y1;
        r->k = k; r->type = type; r->id = next_id++; return r;
    }
    int rectangle(int mat, double x0, double x1, double y0, double y1, double k, int type) {
        return add_obj(make_rect(mat, x0, x1, y0, y1, k, type));
    }
    int box(V3 mn, V3 mx, int mat) {  // objects.rs:442-507 (side order preserved)
        BoxObj* b = new BoxObj; b->mn = mn; b->mx = mx;
        int ids[6];
        ids[0] = rectangle(mat, mn.x, mx.x, mn.y, mx.y, mx.z, 0);
        ids[1] = rectangle(mat, mn.x, mx.x, mn.y, mx.y, mn.z, 0);
        ids[2] = rectangle(mat, mn.x, mx.x, mn.z, mx.z, mx.y, 1);
        ids[3] = rectangle(mat, mn.x, mx.x, mn.z, mx.z, mn.y, 1);
        ids[4] = rectangle(mat, mn.y, mx.y, mn.z, mx.z, mx.x, 2);
        ids[5] = rectangle(mat, mn.y, mx.y, mn.z, mx.z, mn.x, 2);
        for (int i = 0; i < 6; ++i) b->sides.items.push_back(objs[ids[i]].get());
        return add_obj(b);
    }
    int list(const int* children, int n) {
        HittableList* l = new HittableList;
        for (int i = 0; i < n; ++i) l->items.push_back(objs[children[i]].get());
        return add_obj(l);
    }
    int translate(int obj, V3 off) { Translate* t = new Translate; t->obj = objs[obj].get(); t->off = off; return add_obj(t); }
    int rotate_y(int obj, double deg) { RotateY* r = new RotateY; r->init(objs[obj].get(), deg); return add_obj(r); }
    int constant_medium(int boundary, double density, int tex) {  // mediums.rs:23-29
        ConstantMedium* m = new ConstantMedium; m->boundary = objs[boundary].get();
        m->phase = mats[isotropic(tex)].get(); m->neg_inv_density = -1.0 / density;
        m->id = next_id++; m->slot = n_media++; return add_obj(m);
    }
    // bvh.rs:89-156 -- random-axis median split; returns object index, or -1 (err set).
    const Hittable* split_tree(std::vector<const Hittable*> items, double t0, double t1) {
        int axis = (int)bvh_rng.upto(2);
        auto cmp = [axis, this](const Hittable* a, const Hittable* b) {  // bvh.rs:195-203 (boxes at time 0,0)
            Aabb ba, bb;
            if (!a->bbox(0, 0, ba) || !b->bbox(0, 0, bb)) { err = "No bounding box in bvh constructor"; return false; }
            return ba.mn[axis] < bb.mn[axis];
        };
        BvhNode* n = new BvhNode; add_obj(n);
        if (items.size() == 1) { n->left = items[0]; n->right = nullptr; }
        else if (items.size() == 2) {
            const Hittable* b = items[1]; const Hittable* a = items[0];
            if (cmp(a, b)) { n->left = a; n->right = b; } else { n->left = b; n->right = a; }
        } else {
            std::stable_sort(items.begin(), items.end(), cmp);
            size_t mid = items.size() / 2;
            std::vector<const Hittable*> l(items.begin(), items.begin() + mid), r(items.begin() + mid, items.end());
            n->left = split_tree(l, t0, t1); n->right = split_tree(r, t0, t1);
        }
        Aabb bl, br;
        if (!n->left->bbox(t0, t1, bl)) { err = "No bounding box in bvh constructor"; return n; }
        if (n->right) {
            if (!n->right->bbox(t0, t1, br)) { err = "No bounding box in bvh constructor"; return n; }
            n->box = Aabb::surrounding(bl, br);
        } else n->box = bl;
        return n;
    }
    int bvh_node(int list_obj, double t0, double t1) {
        const HittableList* l = dynamic_cast<const HittableList*>(objs[list_obj].get());
        if (!l || l->items.empty()) { err = "bvh_node needs a non-empty HittableList"; return -1; }
        const Hittable* n = split_tree(l->items, t0, t1);
        for (size_t i = 0; i < objs.size(); ++i) if (objs[i].get() == n) return (int)i;
        return -1;
    }
};

// ---------------------------------------------------------------- canned scenes (main.rs:13-451)
static int lambert_color(Scene& s, V3 c) { return s.lambertian(s.solid(c)); }

static int random_scene(Scene& s, SceneRng& rng) {  // main.rs:13-79
    std::vector<int> w;
    int checker = s.checker(s.solid(V3(0.2, 0.3, 0.1)), s.solid(V3(0.9, 0.9, 0.9)));
    w.push_back(s.sphere(V3(0, -1000, 0), 1000, s.lambertian(checker)));
    for (int a = -11; a < 11; ++a) for (int b = -11; b < 11; ++b) {
        double choose = rng.gen();
        double cx = a + 0.9 * rng.gen(); double cz = b + 0.9 * rng.gen();
        V3 center(cx, 0.2, cz);
        if (norm(center - V3(4, 0.2, 0)) > 0.9) {
            if (choose < 0.8) {
                double g[6];
                for (int q = 0; q < 6; ++q) g[q] = rng.gen();  // two gen_random(3) vectors, in draw order
                V3 a1(g[0], g[1], g[2]), a2(g[3], g[4], g[5]);
                int m = lambert_color(s, cmul(a1, a2));
                V3 c2 = center + V3(0, rng.range(0.0, 0.5), 0);
                w.push_back(s.moving_sphere(center, c2, 0.0, 1.0, 0.2, m));
            } else if (choose < 0.95) {
                double r = rng.range(0.5, 1.0), g = rng.range(0.5, 1.0), bl = rng.range(0.5, 1.0);
                double fuzz = rng.range(0.0, 0.5);
                w.push_back(s.sphere(center, 0.2, s.metal(V3(r, g, bl), fuzz)));
            } else w.push_back(s.sphere(center, 0.2, s.dielectric(1.5)));
        }
    }
    w.push_back(s.sphere(V3(0, 1, 0), 1.0, s.dielectric(1.5)));
    w.push_back(s.sphere(V3(-4, 1, 0), 1.0, lambert_color(s, V3(0.4, 0.2, 0.1))));
    w.push_back(s.sphere(V3(4, 1, 0), 1.0, s.metal(V3(0.7, 0.6, 0.5), 0.0)));
    return s.list(w.data(), (int)w.size());
}
// "Synthetic:<n>" -- the library's traversal stress scene (no counterpart in the reference): same draws, same order.
static int synthetic_scene(Scene& s, int n, SceneRng& rng) {
    std::vector<int> w;
    w.push_back(s.sphere(V3(0, -100000, 0), 100000, lambert_color(s, V3(0.5, 0.5, 0.5))));
    for (int i = 0; i < n; ++i) {
        double choose = rng.gen();
        double cx = rng.range(-1000.0, 1000.0), cy = rng.range(0.0, 200.0), cz = rng.range(-1000.0, 1000.0);
        double rad = rng.range(0.2, 2.0);
        int m;
        if (choose < 0.8) {
            double g[6];
            for (int q = 0; q < 6; ++q) g[q] = rng.gen();
            m = lambert_color(s, cmul(V3(g[0], g[1], g[2]), V3(g[3], g[4], g[5])));
        } else if (choose < 0.95) {
            double r = rng.range(0.5, 1.0), g = rng.range(0.5, 1.0), bl = rng.range(0.5, 1.0);
            double fuzz = rng.range(0.0, 0.5);
            m = s.metal(V3(r, g, bl), fuzz);
        } else m = s.dielectric(1.5);
        w.push_back(s.sphere(V3(cx, cy, cz), rad, m));
    }
    return s.list(w.data(), (int)w.size());
}
static int two_spheres(Scene& s) {  // main.rs:81-100
    int checker = s.checker(s.solid(V3(0.2, 0.3, 0.1)), s.solid(V3(0.9, 0.9, 0.9)));
    int w[2] = {s.sphere(V3(0, -10, 0), 10, s.lambertian(checker)), s.sphere(V3(0, 10, 0), 10, s.lambertian(checker))};
    return s.list(w, 2);
}
static int perlin_pair(Scene& s, SceneRng& rng, std::vector<int>& w) {  // main.rs:105-115 / 123-133
    std::shared_ptr<Perlin> p(new Perlin); p->generate(rng);
    int t1 = s.noise(4.0, p), t2 = s.noise(4.0, p);  // pertext.clone() and pertext share table values
    w.push_back(s.sphere(V3(0, -1000, 0), 1000, s.lambertian(t1)));
    w.push_back(s.sphere(V3(0, 2, 0), 2, s.lambertian(t2)));
    return 0;
}
static int two_perlin_spheres(Scene& s, SceneRng& rng) {  // main.rs:102-118
    std::vector<int> w; perlin_pair(s, rng, w); return s.list(w.data(), (int)w.size());
}
static int simple_light(Scene& s, SceneRng& rng) {  // main.rs:120-147
    std::vector<int> w; perlin_pair(s, rng, w);
    w.push_back(s.rectangle(s.light(s.solid(V3(4, 4, 4))), 3, 5, 1, 3, -2, 0));
    return s.list(w.data(), (int)w.size());
}
static void cornell_walls(Scene& s, std::vector<int>& w, int& white, V3 light_c, double lx0, double lx1, double lz0, double lz1) {
    int red = lambert_color(s, V3(0.65, 0.05, 0.05));
    white = lambert_color(s, V3(0.73, 0.73, 0.73));
    int green = lambert_color(s, V3(0.12, 0.45, 0.15));
    int light = s.light(s.solid(light_c));
    w.push_back(s.rectangle(green, 0, 555, 0, 555, 555, 2));
    w.push_back(s.rectangle(red, 0, 555, 0, 555, 0, 2));
    w.push_back(s.rectangle(light, lx0, lx1, lz0, lz1, 554, 1));
    w.push_back(s.rectangle(white, 0, 555, 0, 555, 0, 1));
    w.push_back(s.rectangle(white, 0, 555, 0, 555, 555, 1));
    w.push_back(s.rectangle(white, 0, 555, 0, 555, 555, 0));
}
static int cornell_box(Scene& s) {  // main.rs:149-239
    std::vector<int> w; int white;
    cornell_walls(s, w, white, V3(15, 15, 15), 213, 343, 227, 332);
    int b1 = s.box(V3(0, 0, 0), V3(165, 330, 165), white);
    w.push_back(s.translate(s.rotate_y(b1, 15.0), V3(265, 0, 295)));
    int b2 = s.box(V3(0, 0, 0), V3(165, 165, 165), white);
    w.push_back(s.translate(s.rotate_y(b2, -18.0), V3(130, 0, 65)));
    return s.list(w.data(), (int)w.size());
}
static int cornell_smoke(Scene& s) {  // main.rs:241-330
    std::vector<int> w; int white;
    cornell_walls(s, w, white, V3(7, 7, 7), 113, 443, 127, 432);
    int b1 = s.translate(s.rotate_y(s.box(V3(0, 0, 0), V3(165, 330, 165), white), 15.0), V3(265, 0, 295));
    int b2 = s.translate(s.rotate_y(s.box(V3(0, 0, 0), V3(165, 165, 165), white), -18.0), V3(130, 0, 65));
    w.push_back(s.constant_medium(b1, 0.01, s.solid(V3(0, 0, 0))));
    w.push_back(s.constant_medium(b2, 0.01, s.solid(V3(1, 1, 1))));
    return s.list(w.data(), (int)w.size());
}
static int final_scene(Scene& s, SceneRng& rng) {  // main.rs:332-451
    std::vector<int> boxes1;
    int ground = lambert_color(s, V3(0.48, 0.83, 0.53));
    for (int i = 0; i < 20; ++i) for (int j = 0; j < 20; ++j) {
        double w = 100.0, x0 = -1000.0 + i * w, z0 = -1000.0 + j * w, y0 = 0.0;
        double x1 = x0 + w, y1 = rng.range(1.0, 101.0), z1 = z0 + w;
        boxes1.push_back(s.box(V3(x0, y0, z0), V3(x1, y1, z1), ground));
    }
    std::vector<int> o;
    o.push_back(s.bvh_node(s.list(boxes1.data(), (int)boxes1.size()), 0.0, 1.0));
    o.push_back(s.rectangle(s.light(s.solid(V3(7, 7, 7))), 123, 423, 147, 412, 554, 1));
    V3 c1(400, 400, 200), c2 = c1 + V3(30, 0, 0);
    o.push_back(s.moving_sphere(c1, c2, 0.0, 1.0, 50.0, lambert_color(s, V3(0.7, 0.3, 0.1))));
    o.push_back(s.sphere(V3(260, 150, 45), 50.0, s.dielectric(1.5)));
    o.push_back(s.sphere(V3(0, 150, 145), 50.0, s.metal(V3(0.8, 0.8, 0.9), 1.0)));
    o.push_back(s.sphere(V3(360, 150, 145), 70.0, s.dielectric(1.5)));                 // boundary.clone()
    int bnd = s.sphere(V3(360, 150, 145), 70.0, s.dielectric(1.5));
    o.push_back(s.constant_medium(bnd, 0.2, s.solid(V3(0.2, 0.4, 0.9))));
    int fog = s.sphere(V3(0, 0, 0), 5000.0, s.dielectric(1.5));
    o.push_back(s.constant_medium(fog, 0.0001, s.solid(V3(1, 1, 1))));
    o.push_back(s.sphere(V3(400, 200, 400), 100.0, lambert_color(s, V3(0, 1, 0))));
    std::shared_ptr<Perlin> p(new Perlin); p->generate(rng);
    o.push_back(s.sphere(V3(220, 280, 300), 80.0, s.lambertian(s.noise(0.1, p))));
    std::vector<int> boxes2;
    int white = lambert_color(s, V3(0.73, 0.73, 0.73));
    for (int i = 0; i < 1000; ++i) {
        double x = rng.range(0.0, 165.0), y = rng.range(0.0, 165.0), z = rng.range(0.0, 165.0);
        boxes2.push_back(s.sphere(V3(x, y, z), 10.0, white));
    }
    int cluster = s.bvh_node(s.list(boxes2.data(), (int)boxes2.size()), 0.0, 1.0);
    o.push_back(s.translate(s.rotate_y(cluster, 15.0), V3(-100, 270, 395)));
    return s.list(o.data(), (int)o.size());
}

// ---------------------------------------------------------------- camera (cameras.rs:23-98)
struct Camera {
    V3 origin, llc, horizontal, vertical, u, v, w; double lens_radius, shutter_open, shutter_close;
    void init(V3 from, V3 at, V3 vup, double vfov, double aspect, double aperture, double focus, double so, double sc) {
        double theta = vfov * (kPi / 180.0);
        double h = std::tan(theta / 2.0);
        double vh = 2.0 * h, vw = aspect * vh;
        w = normalize(from - at); u = normalize(cross(vup, w)); v = cross(w, u);
        origin = from; horizontal = focus * vw * u; vertical = focus * vh * v;
        llc = origin - horizontal / 2.0 - vertical / 2.0 - focus * w;
        lens_radius = aperture / 2.0; shutter_open = so; shutter_close = sc;
    }
    Ray ray(double s, double t, double l1, double l2, double tm) const {  // cameras.rs:82-98
        double dx, dy; unit_disk(l1, l2, dx, dy);
        V3 offset = u * (lens_radius * dx) + v * (lens_radius * dy);
        double time = (shutter_open != 0.0 || shutter_close != 0.0) ? shutter_open + (shutter_close - shutter_open) * tm : 0.0;
        return Ray(origin + offset, llc + s * horizontal + t * vertical - origin - offset, time);
    }
};

// ---------------------------------------------------------------- integrator (lib.rs:56-77)
static V3 get_color(const Ray& ray, V3 bg, const Hittable* world, int depth, int max_depth, Ctx& cx) {
    if (depth == 0) return V3(0, 0, 0);
    cx.bounce = max_depth - depth + 1;
    if (cx.st) cx.st->rays++;
    HitRecord hr;
    if (world->try_hit(ray, 0.001, std::numeric_limits<double>::max(), cx, hr)) {
        V3 emitted = hr.mat->emitted(hr.u, hr.v, hr.p, cx.st);
        V3 atten; Ray scattered;
        if (cx.st) cx.st->shade[hr.mat->kind]++;
        if (hr.mat->scatter(ray, hr, cx, atten, scattered))
            return emitted + cmul(atten, get_color(scattered, bg, world, depth - 1, max_depth, cx));
        return emitted;
    }
    return bg;
}

struct RenderParams {
    int width, height, spp, sample_offset, max_depth;
    double background[3];
    u64 seed;
};

static Ray camera_ray(const Camera& cam, const RenderParams& rp, int px, int py, Ctx& cx) {
    // main.rs:567-574: j counts rows bottom-up, the image row is H-1-j.
    cx.bounce = 0;
    int j = rp.height - 1 - py;
    double u = (px + cx.xi(0)) / (double)(rp.width - 1);
    double v = (j + cx.xi(1)) / (double)(rp.height - 1);
    return cam.ray(u, v, cx.xi(2), cx.xi(3), cx.xi(4));
}

static void render_rows(const Scene& sc, const Camera& cam, const RenderParams& rp, int row0, int row1,
                        double* accum, double* accum_sq, Stats* st) {
    V3 bg(rp.background[0], rp.background[1], rp.background[2]);
    for (int py = row0; py < row1; ++py) for (int px = 0; px < rp.width; ++px) {
        V3 sum(0, 0, 0), sumsq(0, 0, 0);
        for (int s = 0; s < rp.spp; ++s) {
            Ctx cx; cx.pixel = (u32)(py * rp.width + px); cx.sample = (u32)(rp.sample_offset + s);
            cx.key[0] = (u32)rp.seed; cx.key[1] = (u32)(rp.seed >> 32); cx.st = st; cx.bounce = 0;
            Ray r = camera_ray(cam, rp, px, py, cx);
            V3 c = get_color(r, bg, sc.world, rp.max_depth, rp.max_depth, cx);
            sum = sum + c; sumsq = sumsq + cmul(c, c);
        }
        size_t o = ((size_t)py * rp.width + px) * 3;
        accum[o] = sum.x; accum[o + 1] = sum.y; accum[o + 2] = sum.z;
        if (accum_sq) { accum_sq[o] = sumsq.x; accum_sq[o + 1] = sumsq.y; accum_sq[o + 2] = sumsq.z; }
    }
}

}  // namespace orc

// ==================================================================== C ABI for ctypes
using namespace orc;

struct OrcScene { Scene s; };
struct OrcCamera { Camera c; };

extern "C" {

void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { philox4x32_10(ctr, key, out); }
double orc_xi(uint32_t pixel, uint32_t sample, uint64_t seed, int bounce, int dim) {
    Ctx c; c.pixel = pixel; c.sample = sample; c.key[0] = (u32)seed; c.key[1] = (u32)(seed >> 32); c.bounce = bounce; c.st = nullptr;
    return c.xi(dim);
}
uint64_t orc_scene_rng(uint64_t seed, int n, double* out) {  // first n gen() values of the scene stream
    SceneRng r(seed); for (int i = 0; i < n; ++i) out[i] = r.gen(); return r.s;
}

OrcScene* orc_scene_new(uint64_t bvh_seed) { OrcScene* s = new OrcScene; s->s.bvh_rng = SceneRng(bvh_seed ^ 0xB5AD4ECEDA1CE2A9ull); return s; }
void orc_scene_free(OrcScene* s) { delete s; }
const char* orc_scene_error(OrcScene* s) { return s->s.err.c_str(); }
int orc_scene_prim_count(OrcScene* s) { return s->s.next_id; }

int orc_solid_color(OrcScene* s, double r, double g, double b) { return s->s.solid(V3(r, g, b)); }
int orc_checker(OrcScene* s, int even, int odd) { return s->s.checker(even, odd); }
int orc_noise(OrcScene* s, double scale, const double* ranvec768, const int32_t* perm768) {
    std::shared_ptr<Perlin> p(new Perlin);
    for (int i = 0; i < 256; ++i) p->ranvec[i] = V3(ranvec768[3 * i], ranvec768[3 * i + 1], ranvec768[3 * i + 2]);
    for (int k = 0; k < 3; ++k) for (int i = 0; i < 256; ++i) p->perm[k][i] = perm768[k * 256 + i];
    return s->s.noise(scale, p);
}
int orc_lambertian(OrcScene* s, int tex) { return s->s.lambertian(tex); }
int orc_metal(OrcScene* s, double r, double g, double b, double fuzz) { return s->s.metal(V3(r, g, b), fuzz); }
int orc_dielectric(OrcScene* s, double ir) { return s->s.dielectric(ir); }
int orc_diffuse_light(OrcScene* s, int tex) { return s->s.light(tex); }
int orc_isotropic(OrcScene* s, int tex) { return s->s.isotropic(tex); }
int orc_sphere(OrcScene* s, const double c[3], double r, int mat) { return s->s.sphere(V3(c[0], c[1], c[2]), r, mat); }
int orc_moving_sphere(OrcScene* s, const double c0[3], const double c1[3], double t0, double t1, double r, int mat) {
    return s->s.moving_sphere(V3(c0[0], c0[1], c0[2]), V3(c1[0], c1[1], c1[2]), t0, t1, r, mat);
}
int orc_rectangle(OrcScene* s, int mat, double x0, double x1, double y0, double y1, double k, int type) {
    return s->s.rectangle(mat, x0, x1, y0, y1, k, type);
}
int orc_box(OrcScene* s, const double mn[3], const double mx[3], int mat) {
    return s->s.box(V3(mn[0], mn[1], mn[2]), V3(mx[0], mx[1], mx[2]), mat);
}
int orc_list(OrcScene* s, const int* children, int n) { return s->s.list(children, n); }
int orc_bvh_node(OrcScene* s, int list_obj, double t0, double t1) { return s->s.bvh_node(list_obj, t0, t1); }
int orc_translate(OrcScene* s, int obj, const double off[3]) { return s->s.translate(obj, V3(off[0], off[1], off[2])); }
int orc_rotate_y(OrcScene* s, int obj, double deg) { return s->s.rotate_y(obj, deg); }
int orc_constant_medium(OrcScene* s, int boundary, double density, int tex) { return s->s.constant_medium(boundary, density, tex); }

// Canned scenes (main.rs:520-528).  Returns the HittableList object, or -1 for an unknown name.
int orc_canned(OrcScene* s, const char* type, uint64_t scene_seed) {
    SceneRng rng(scene_seed);
    std::string t(type);
    if (t == "Random") return random_scene(s->s, rng);
    if (t == "TwoSpheres") return two_spheres(s->s);
    if (t == "TwoPerlinSpheres") return two_perlin_spheres(s->s, rng);
    if (t == "SimpleLight") return simple_light(s->s, rng);
    if (t == "Cornell") return cornell_box(s->s);
    if (t == "CornellSmoke") return cornell_smoke(s->s);
    if (t == "FinalScene") return final_scene(s->s, rng);
    if (t.compare(0, 10, "Synthetic:") == 0) return synthetic_scene(s->s, std::atoi(t.c_str() + 10), rng);
    return -1;
}
// main.rs:531 -- world = BvhNode::new(list, 0, 1).  use_bvh = 0 keeps the plain list (brute force).
int orc_set_world(OrcScene* s, int list_obj, int use_bvh) {
    int w = list_obj;
    if (use_bvh) { w = s->s.bvh_node(list_obj, 0.0, 1.0); if (w < 0) return -1; }
    s->s.world = s->s.objs[w].get();
    return s->s.err.empty() ? 0 : -1;
}
int orc_world_bbox(OrcScene* s, double out[6]) {
    Aabb b; if (!s->s.world->bbox(0.0, 1.0, b)) return -1;
    out[0] = b.mn.x; out[1] = b.mn.y; out[2] = b.mn.z; out[3] = b.mx.x; out[4] = b.mx.y; out[5] = b.mx.z; return 0;
}
int orc_obj_bbox(OrcScene* s, int obj, double t0, double t1, double out[6]) {
    Aabb b; if (!s->s.objs[obj]->bbox(t0, t1, b)) return -1;
    out[0] = b.mn.x; out[1] = b.mn.y; out[2] = b.mn.z; out[3] = b.mx.x; out[4] = b.mx.y; out[5] = b.mx.z; return 0;
}

OrcCamera* orc_camera_new(const double from[3], const double at[3], const double vup[3], double vfov, double aspect,
                          double aperture, double focus, double shutter_open, double shutter_close) {
    OrcCamera* c = new OrcCamera;
    c->c.init(V3(from[0], from[1], from[2]), V3(at[0], at[1], at[2]), V3(vup[0], vup[1], vup[2]), vfov, aspect, aperture, focus,
              shutter_open, shutter_close);
    return c;
}
void orc_camera_free(OrcCamera* c) { delete c; }
// out[0..20]: origin, llc, horizontal, vertical, u, v, w ; out[21] lens_radius
void orc_camera_get(OrcCamera* c, double out[22]) {
    const V3* vs[7] = {&c->c.origin, &c->c.llc, &c->c.horizontal, &c->c.vertical, &c->c.u, &c->c.v, &c->c.w};
    for (int i = 0; i < 7; ++i) { out[3 * i] = vs[i]->x; out[3 * i + 1] = vs[i]->y; out[3 * i + 2] = vs[i]->z; }
    out[21] = c->c.lens_radius;
}
// Camera ray for (pixel px,py ; sample) under the fixed schedule: out = o(3), d(3), time.
void orc_camera_ray(OrcCamera* c, int width, int height, int px, int py, uint32_t sample, uint64_t seed, double out[7]) {
    RenderParams rp; rp.width = width; rp.height = height; rp.seed = seed;
    Ctx cx; cx.pixel = (u32)(py * width + px); cx.sample = sample; cx.key[0] = (u32)seed; cx.key[1] = (u32)(seed >> 32); cx.st = nullptr;
    Ray r = camera_ray(c->c, rp, px, py, cx);
    out[0] = r.o.x; out[1] = r.o.y; out[2] = r.o.z; out[3] = r.d.x; out[4] = r.d.y; out[5] = r.d.z; out[6] = r.time;
}

// Single closest-hit query (lib.rs:63 when tmin=0.001, tmax=f64::MAX).  `pixel,sample,bounce,seed`
// select the medium free-flight draws.  out: t, p(3), normal(3), front_face, u, v ; returns id or -1.
int orc_hit(OrcScene* s, const double o[3], const double d[3], double time, double tmin, double tmax,
            uint32_t pixel, uint32_t sample, int bounce, uint64_t seed, double out[10]) {
    Ctx cx; cx.pixel = pixel; cx.sample = sample; cx.key[0] = (u32)seed; cx.key[1] = (u32)(seed >> 32); cx.bounce = bounce; cx.st = nullptr;
    Ray r(V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), time);
    HitRecord h;
    if (!s->s.world->try_hit(r, tmin, tmax, cx, h)) return -1;
    out[0] = h.t; out[1] = h.p.x; out[2] = h.p.y; out[3] = h.p.z; out[4] = h.normal.x; out[5] = h.normal.y; out[6] = h.normal.z;
    out[7] = h.front_face ? 1.0 : 0.0; out[8] = h.u; out[9] = h.v;
    return h.id;
}

// One scatter event at the closest hit of the given ray (lib.rs:63-65).  Returns -1 miss, 0 absorbed
// (no scatter), 1 scattered.  out: emitted(3), attenuation(3), scattered o(3), d(3), time.
int orc_scatter(OrcScene* s, const double o[3], const double d[3], double time, uint32_t pixel, uint32_t sample, int bounce,
                uint64_t seed, double out[13]) {
    Ctx cx; cx.pixel = pixel; cx.sample = sample; cx.key[0] = (u32)seed; cx.key[1] = (u32)(seed >> 32); cx.bounce = bounce; cx.st = nullptr;
    Ray r(V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), time);
    HitRecord h;
    if (!s->s.world->try_hit(r, 0.001, std::numeric_limits<double>::max(), cx, h)) return -1;
    V3 e = h.mat->emitted(h.u, h.v, h.p, nullptr), a(0, 0, 0); Ray sc;
    bool ok = h.mat->scatter(r, h, cx, a, sc);
    out[0] = e.x; out[1] = e.y; out[2] = e.z; out[3] = a.x; out[4] = a.y; out[5] = a.z;
    out[6] = sc.o.x; out[7] = sc.o.y; out[8] = sc.o.z; out[9] = sc.d.x; out[10] = sc.d.y; out[11] = sc.d.z; out[12] = sc.time;
    return ok ? 1 : 0;
}

void orc_texture_value(OrcScene* s, int tex, double u, double v, const double p[3], double out[3]) {
    V3 c = s->s.texs[tex]->value(u, v, V3(p[0], p[1], p[2]), nullptr);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}
void orc_sphere_uv(const double p[3], double out[2]) { sphere_uv(V3(p[0], p[1], p[2]), out[0], out[1]); }
double orc_reflectance(double cosine, double ref_idx) { return Dielectric::reflectance(cosine, ref_idx); }
void orc_refract(const double uv[3], const double n[3], double eta, double out[3]) {
    V3 r = refract(V3(uv[0], uv[1], uv[2]), V3(n[0], n[1], n[2]), eta); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_reflect(const double v[3], const double n[3], double out[3]) {
    V3 r = reflect(V3(v[0], v[1], v[2]), V3(n[0], n[1], n[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_unit_vector(double u1, double u2, double out[3]) { V3 r = unit_vector(u1, u2); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
int orc_image_height(int width, double aspect) { return (int)std::round((double)width / aspect); }  // main.rs:515-516

// utils.rs:49-63 -- 256*clamp(sqrt(c/spp),0,0.999) -> u8
void orc_tonemap(const double* accum, int n_pixels, int spp, uint8_t* rgb) {
    double scale = 1.0 / (double)spp;
    for (int i = 0; i < n_pixels * 3; ++i) {
        double v = std::sqrt(scale * accum[i]);
        rgb[i] = (uint8_t)(256.0 * std::max(std::min(v, 0.999), 0.0));
    }
}

// Primary hits for sample `sample` of every pixel (camera ray + lib.rs:63).  id = -1 / t = inf on a miss.
void orc_trace_primary(OrcScene* s, OrcCamera* c, int width, int height, uint32_t sample, uint64_t seed,
                       int32_t* id, double* t, double* normal /*nullable, 3 per pixel*/) {
    RenderParams rp; rp.width = width; rp.height = height; rp.seed = seed;
    for (int py = 0; py < height; ++py) for (int px = 0; px < width; ++px) {
        Ctx cx; cx.pixel = (u32)(py * width + px); cx.sample = sample; cx.key[0] = (u32)seed; cx.key[1] = (u32)(seed >> 32); cx.st = nullptr;
        Ray r = camera_ray(c->c, rp, px, py, cx);
        cx.bounce = 1;
        HitRecord h; size_t i = (size_t)py * width + px;
        if (s->s.world->try_hit(r, 0.001, std::numeric_limits<double>::max(), cx, h)) {
            id[i] = h.id; t[i] = h.t;
            if (normal) { normal[3 * i] = h.normal.x; normal[3 * i + 1] = h.normal.y; normal[3 * i + 2] = h.normal.z; }
        } else {
            id[i] = -1; t[i] = kInf;
            if (normal) { normal[3 * i] = normal[3 * i + 1] = normal[3 * i + 2] = 0; }
        }
    }
}

// main.rs:567-599 render loop.  accum / accum_sq: [H][W][3] f64 sums over samples (accum_sq nullable).
// stats_out (nullable, 14 u64): rays, box, sph, msph, rect, inst, med, perlin, shade[5], 0.
void orc_render(OrcScene* s, OrcCamera* c, int width, int height, int spp, int sample_offset, int max_depth,
                const double background[3], uint64_t seed, int n_threads, double* accum, double* accum_sq, uint64_t* stats_out) {
    RenderParams rp; rp.width = width; rp.height = height; rp.spp = spp; rp.sample_offset = sample_offset; rp.max_depth = max_depth;
    rp.background[0] = background[0]; rp.background[1] = background[1]; rp.background[2] = background[2]; rp.seed = seed;
    if (n_threads < 1) n_threads = 1;
    std::vector<Stats> st((size_t)n_threads);
    std::vector<std::thread> th;
    // rows are dealt round-robin in small strips so that threads finish together
    auto worker = [&](int tid) {
        for (int row = tid; row < height; row += n_threads)
            render_rows(s->s, c->c, rp, row, row + 1, accum, accum_sq, &st[(size_t)tid]);
    };
    for (int i = 1; i < n_threads; ++i) th.emplace_back(worker, i);
    worker(0);
    for (auto& t : th) t.join();
    if (stats_out) {
        Stats tot; for (auto& x : st) tot.add(x);
        uint64_t v[14] = {tot.rays, tot.box, tot.sph, tot.msph, tot.rect, tot.inst, tot.med, tot.perlin,
                          tot.shade[0], tot.shade[1], tot.shade[2], tot.shade[3], tot.shade[4], 0};
        std::memcpy(stats_out, v, sizeof(v));
    }
}

}  // extern "C"

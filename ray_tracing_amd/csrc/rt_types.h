// rt_types.h -- flattened-scene layout shared by the host flattener and the gfx950 kernels.
//
// Everything the kernels read is a flat array of 16-byte records (so that every fetch is one
// global_load_dwordx4) or a small int array.  The host flattener (flatten.cpp) turns the
// reference's pointer tree (objects.rs / bvh.rs / transrot.rs / mediums.rs) into these arrays.
#pragma once
#include <stdint.h>

struct alignas(16) rt_f4 { float x, y, z, w; };
struct alignas(8) rt_i2 { int32_t x, y; };

// ---- primitive kinds (also the top bits of a hit code)
enum { RT_KIND_SPHERE = 0, RT_KIND_MSPHERE = 1, RT_KIND_RECT = 2, RT_KIND_BOX = 3, RT_KIND_MEDIUM = 4 };
#define RT_HIT_NONE 0xFFFFFFFFu
#define RT_SRC_RESOLVED 0xFFFFFFFEu   /* in a queued ray's source-primitive word: its closest surface hit is already in the hit stream */
#define RT_HIT_CODE(kind, index) (((uint32_t)(kind) << 28) | (uint32_t)(index))
#define RT_HIT_KIND(code) ((code) >> 28)
#define RT_HIT_INDEX(code) ((code) & 0x0FFFFFFFu)

// ---- BVH2 node, 64 B = 4 x rt_f4; both child boxes live in the parent:
//   q0 = (L.min.x, L.min.y, L.min.z, L.max.x)
//   q1 = (L.max.y, L.max.z, R.min.x, R.min.y)
//   q2 = (R.min.z, R.max.x, R.max.y, R.max.z)
//   q3 = (left, right, 0, 0) as int32 bit patterns
// child >= 0 : index of an inner node.   child < 0 : leaf, x = ~child:
//   bit 30 set : instance leaf, index = x & 0x3FFFFFFF
//   else       : kind = (x >> 28) & 3, count = ((x >> 26) & 3) + 1, first = x & 0x03FFFFFF
// An absent child has an empty box (min = +inf, max = -inf) and is never entered.
#define RT_LEAF_INSTANCE_BIT 0x40000000
#ifndef RT_LEAF_MAX_PRIMS
#define RT_LEAF_MAX_PRIMS 4
#endif
static inline int32_t rt_leaf_prims(int kind, int first, int count) {
    return ~(int32_t)(((uint32_t)kind << 28) | ((uint32_t)(count - 1) << 26) | (uint32_t)first);
}
static inline int32_t rt_leaf_instance(int index) { return ~(int32_t)(RT_LEAF_INSTANCE_BIT | (uint32_t)index); }

// ---- primitives
//   sphere   : 1 x f4 = (cx, cy, cz, r)                               objects.rs:155-159
//   msphere  : 2 x f4 = (c0x, c0y, c0z, r), (c1-c0 xyz, t0) ; meta.z below holds 1/(t1-t0)
//              stored as a third f4 = (1/(t1-t0), 0, 0, 0)             objects.rs:257-264
//   rect     : 2 x f4 = (x0, x1, y0, y1), (k, type as int bits, 0, 0)  objects.rs:346-361
//              type 0 = Xy, 1 = Xz, 2 = Yz                            objects.rs:339-343
//   box      : 2 x f4 = (min xyz, material as int bits), (max xyz, object id of side 0 as int bits)
//              a BoxObj (objects.rs:433-507): its six rectangles as ONE primitive; side s (order of
//              objects.rs:445-500: z max, z min, y max, y min, x max, x min) has object id = id0 + s and
//              hit code index = box index * 8 + s
//   *_meta   : rt_i2 = (material index, object id in creation order)
#define RT_MSPHERE_F4 3
#define RT_RECT_F4 2

// ---- instance (a collapsed chain of Translate / RotateY, transrot.rs), 2 x f4:
//   i0 = (cos, sin, cx, cy)   object-space point = R * world + c ; R: x' = cos x - sin z, z' = sin x + cos z
//   i1 = (cz, root node as int, first op as int, op count as int)
// ops (outermost first), 1 x f4 each: Translate = (ox, oy, oz, type 0), RotateY = (cos, sin, 0, type 1);
// the type is an int bit pattern in .w.
#define RT_INSTANCE_F4 2
enum { RT_OP_TRANSLATE = 0, RT_OP_ROTATE_Y = 1 };

// ---- constant medium (mediums.rs:14-21), 10 x f4:
//   m0 = (box.min xyz, -1/density)      m1 = (box.max xyz, material index as int)
//   m2 = int (instance or -1, sphere first, sphere count, msphere first)
//   m3 = int (msphere count, rect first, rect count, object id)
//   m4 = int (box first, box count, local surface 0, local surface 1)
//   m5 = int (fast kind: 0 = scan the ranges above, 1 = the boundary is the one sphere in m6, 2 = the one box in m6, m7;
//             bit 0: m8, m9 hold the instance record of m2.x, bit 1: the boundary encloses the whole scene with room to spare
//             (flatten.cpp, emit_medium_locals); n | mask << 8: number of local surfaces 0..3 and, bit k, whether
//             local surface k lies wholly inside the boundary; local surface 2)
//   local surfaces: the only primitives (as one-primitive leaf codes) a ray can hit while inside the boundary, see flatten.cpp
//   m6, m7 = that primitive's record, m8, m9 = that instance's record
#define RT_MEDIUM_F4 10

// ---- material (materials.rs), 2 x f4:
//   a0 = int/float (type, texture index or -1, fuzz, ir)     a1 = (rgb, 0): Metal albedo, or the colour of a
//   SolidColor texture baked in (texture index -1) so that the common case needs no texture fetch; a1.w = int 1 when the
//   texture tree holds a Noise (costly to evaluate)
enum { RT_MAT_LAMBERTIAN = 0, RT_MAT_METAL = 1, RT_MAT_DIELECTRIC = 2, RT_MAT_LIGHT = 3, RT_MAT_ISOTROPIC = 4 };
#define RT_MATERIAL_F4 2

// ---- texture (textures.rs), 2 x f4:
//   t0 = int (type, even child | perlin table, odd child, 0)    t1 = (rgb | scale, 0, 0, 0)
enum { RT_TEX_SOLID = 0, RT_TEX_CHECKER = 1, RT_TEX_NOISE = 2 };
#define RT_TEXTURE_F4 2

// ---- Perlin table (textures.rs:99-109): 256 x f4 gradient (xyz, 0) then 768 bytes of permutations
#define RT_PERLIN_VEC_F4 256
#define RT_PERLIN_PERM_BYTES 768

// Device-side view of a flattened scene (all pointers are device pointers inside the library;
// the same struct with host pointers is what the flattener produces).
struct RtSceneView {
    const rt_f4* nodes;
    const rt_f4* spheres;   const rt_i2* sphere_meta;
    const rt_f4* mspheres;  const rt_i2* msphere_meta;
    const rt_f4* rects;     const rt_i2* rect_meta;
    const rt_f4* instances; const rt_f4* ops;
    const rt_f4* media;
    const rt_f4* materials; const rt_f4* textures;
    const rt_f4* perlin_vec; const uint8_t* perlin_perm;
    const rt_f4* boxes;
    int32_t root;           // TLAS root node, -1 for an empty world
    int32_t n_nodes, n_spheres, n_mspheres, n_rects, n_instances, n_ops, n_media, n_materials, n_textures, n_perlin;
    int32_t max_stack;      // deepest traversal stack the BVH can need (TLAS depth + sentinel + BLAS depth)
    int32_t n_boxes;
    // The same tree collapsed to 4-wide nodes with quantised child boxes (layout: wide_bvh.cpp), 4 x rt_f4 per node.
    const rt_f4* wnodes;
    const int32_t* winst_roots;   // per instance: the wide node that roots its subtree
    int32_t n_wnodes, wroot, wmax_stack;
};

// Camera in device form (cameras.rs:23-34); dir0 = lower_left_corner - origin is precomputed in
// f64 so that the fp32 direction does not lose digits to cancellation.
struct RtCameraDev {
    float origin[3], dir0[3], horizontal[3], vertical[3], u[3], v[3];
    float lens_radius, shutter_open, shutter_close;
};

struct RtFrameDev {
    int32_t width, height, n_pixels;
    int32_t spp_chunk;       // samples of every pixel traced by this pass
    int32_t sample_base;     // absolute index of the pass's first sample (Philox counter)
    int32_t max_depth;
    uint32_t key0, key1;     // Philox key = render seed
    float inv_wm1, inv_hm1;  // 1/(W-1), 1/(H-1)   main.rs:572-573
    double inv_n_set;        // for path id -> (sample, set index) without an integer divide
    float background[3];
    // The pixels this pass renders ("the set"): n_set of them, numbered row by row.  subset = 0: the whole image (set index =
    // pixel index).  subset = 1: the rectangle (x0, y0, w) of which only the row bands of one shard are kept -- bands of
    // `band` rows dealt round-robin to `n_shards` shards, this one being `shard` (band = 0: every row).
    int32_t n_set, subset;
    int32_t set_x0, set_y0, set_w, band, n_shards, shard;
    double inv_set_w;
};

#define RT_TMIN 0.001f           /* lib.rs:63 */
#define RT_STACK_DEPTH 32        /* LDS traversal stack entries per ray */
#define RT_EMPTY_STACK 0x7FFFFFFF
#define RT_SENTINEL 0x7FFFFFFE

// lbvh.h -- linear BVH over a list of boxed items, as per-element steps that run either as HIP kernels (lbvh.hip) or in
// plain loops on the host (lbvh_host.cpp).  It replaces the host SAH builder for the top level of large scenes, where the
// reference's BvhNode::new (bvh.rs:89-156, O(n log^2 n) on one core) and even a binned SAH take seconds.
//
// Pipeline: Morton code of every item's box centre inside the box of all centres -> sort of (code, item) pairs -> radix tree
// over the sorted keys (Karras, "Maximizing parallelism in the construction of BVHs, octrees and k-d trees", HPG 2012) ->
// fill passes that write the 64-byte nodes of rt_types.h (both child boxes live in the parent) and measure the depth.
// Equal codes are told apart by their sorted position, so the tree is fully determined by the sorted order.
// The fill goes bottom-up in whole passes, one launch each: a node is filled in pass p once both children were filled in
// passes < p.  Nothing is handed from one workgroup to another inside a launch (on this GPU that needs an L2 write-back and
// an L1 invalidate per hand-off); the number of passes is the height of the tree.
#pragma once
#include "rt_device.h"

struct RtLbvhFrame { float lo[3]; float inv_ext[3]; };   // box of the item centres: code = quantise((centre - lo) * inv_ext)

RT_HD uint64_t rt_expand_bits21(uint64_t x) {              // 21 bits -> every third bit
    x &= 0x1FFFFFull;
    x = (x | x << 32) & 0x1F00000000FFFFull;
    x = (x | x << 16) & 0x1F0000FF0000FFull;
    x = (x | x << 8) & 0x100F00F00F00F00Full;
    x = (x | x << 4) & 0x10C30C30C30C30C3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
// An item = two records: (box.min xyz, child code as int bits), (box.max xyz, 0).  63-bit Morton code of its box centre,
// 21 bits per axis: one huge item (a ground sphere) stretches the frame a thousandfold and must not leave the rest of
// the scene in a handful of cells.
RT_HD uint64_t rt_lbvh_key(const RtLbvhFrame& fr, rt_f4 lo, rt_f4 hi) {
    float c[3] = {0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z)};
    uint64_t q[3];
    for (int a = 0; a < 3; ++a) {
        float u = (c[a] - fr.lo[a]) * fr.inv_ext[a] * 2097152.0f;
        q[a] = u >= 2097151.0f ? 2097151ull : (u > 0.0f ? (uint64_t)u : 0ull);   // also sends NaN to 0
    }
    return (rt_expand_bits21(q[0]) << 2) | (rt_expand_bits21(q[1]) << 1) | rt_expand_bits21(q[2]);
}

RT_HD int rt_clz64(uint64_t x) {
#if RT_ON_DEVICE
    return __clzll((long long)x);
#else
    return x ? __builtin_clzll(x) : 64;
#endif
}
// Length of the common prefix of the sorted keys at positions i and j, with the position as tie-breaker for equal codes
// (so that the tree is a function of the sorted order alone); -1 when j is outside [0, n).
RT_HD int rt_lbvh_delta(const uint64_t* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    uint64_t x = keys[i] ^ keys[j];
    return x ? rt_clz64(x) : 64 + (rt_clz64((uint64_t)(uint32_t)(i ^ j)) - 32);
}

// Inner node i (0 <= i < n - 1) of the radix tree over n >= 2 sorted keys.  A child reference is an inner node index
// (>= 0) or ~leaf position (< 0).
RT_HD void rt_lbvh_node(const uint64_t* keys, int n, int i, int* left, int* right) {
    int d = rt_lbvh_delta(keys, n, i, i + 1) - rt_lbvh_delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    int dmin = rt_lbvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (rt_lbvh_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (rt_lbvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = rt_lbvh_delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {          // t = ceil(l / 2), ceil(l / 4), ..., 1
        if (rt_lbvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    int gamma = i + s * d + (d < 0 ? -1 : 0);
    int lo = i < j ? i : j, hi = i < j ? j : i;
    left[i] = lo == gamma ? ~gamma : gamma;
    right[i] = hi == gamma + 1 ? ~(gamma + 1) : gamma + 1;
}

struct RtLbvhBox { float lo[3], hi[3]; };

// Pass `pass` (1, 2, ...) of the fill for node i.  filled[i] = pass in which node i was filled (0 = not yet).  Child boxes:
// of a leaf, its item's box; of an inner node, box[] / depth[] written when that node was filled.  `node_base` is added to
// inner child indices (the nodes are appended to an array that already holds the bottom-level trees).
RT_HD void rt_lbvh_fill(int i, int pass, const int* left, const int* right, const uint32_t* order, const rt_f4* items, RtLbvhBox* box,
                        int* depth, int* filled, rt_f4* nodes, int node_base) {
    if (filled[i] != 0) return;
    RtLbvhBox cb[2]; int code[2], dep[2];
    for (int c = 0; c < 2; ++c) {
        int ref = c == 0 ? left[i] : right[i];
        if (ref < 0) {
            uint32_t item = order[~ref];                   // the item at that sorted position
            rt_f4 lo = items[2 * (size_t)item], hi = items[2 * (size_t)item + 1];
            cb[c].lo[0] = lo.x; cb[c].lo[1] = lo.y; cb[c].lo[2] = lo.z; cb[c].hi[0] = hi.x; cb[c].hi[1] = hi.y; cb[c].hi[2] = hi.z;
            code[c] = f2i(lo.w); dep[c] = 0;
        } else {
            int f = filled[ref];
            if (f == 0 || f >= pass) return;               // that child belongs to this pass or a later one
            cb[c] = box[ref]; code[c] = ref + node_base; dep[c] = depth[ref];
        }
    }
    RtLbvhBox u;
    for (int a = 0; a < 3; ++a) { u.lo[a] = fminf(cb[0].lo[a], cb[1].lo[a]); u.hi[a] = fmaxf(cb[0].hi[a], cb[1].hi[a]); }
    box[i] = u; depth[i] = 1 + (dep[0] > dep[1] ? dep[0] : dep[1]);
    rt_f4* q = nodes + 4 * (size_t)i;
    q[0].x = cb[0].lo[0]; q[0].y = cb[0].lo[1]; q[0].z = cb[0].lo[2]; q[0].w = cb[0].hi[0];
    q[1].x = cb[0].hi[1]; q[1].y = cb[0].hi[2]; q[1].z = cb[1].lo[0]; q[1].w = cb[1].lo[1];
    q[2].x = cb[1].lo[2]; q[2].y = cb[1].hi[0]; q[2].z = cb[1].hi[1]; q[2].w = cb[1].hi[2];
    q[3].x = i2f(code[0]); q[3].y = i2f(code[1]); q[3].z = 0.0f; q[3].w = 0.0f;
    filled[i] = pass;
}

// refit.h -- boxes of an existing tree recomputed from the primitives as they are now, as per-element steps that run either
// as HIP kernels (refit.hip) or in plain loops on the host (refit_host.cpp).  The reference builds a new BvhNode for a changed
// world (src/main.rs:531, src/bvh.rs:89-156); a tree whose primitives only moved keeps its shape here, and its boxes follow
// the rules of the reference's bounding boxes: Sphere src/objects.rs:241-244, MovingSphere src/objects.rs:328-335 at
// (time0, time1) = (0, 1), Rectangle src/objects.rs:413-429, BoxObj src/objects.rs:514-516, RotateY src/transrot.rs:60-82,
// Translate src/transrot.rs:34-38.
//
// The binary nodes are refitted bottom-up in whole passes, one launch each, like the fill of the linear builder (lbvh.h): a
// node is done in pass p once the nodes its boxes are made from were done in passes < p.  The 4-wide quantised nodes
// (wide_bvh.cpp) are then re-encoded from the binary child boxes each of their children came from.
#pragma once
#include "rt_device.h"

struct RtBox6 { float lo[3], hi[3]; };

RT_HD float rt_f_down(double v) {
#if RT_ON_DEVICE
    return __double2float_rd(v);
#else
    float f = (float)v; if ((double)f > v) f = nextafterf(f, -INFINITY); return f;
#endif
}
RT_HD float rt_f_up(double v) {
#if RT_ON_DEVICE
    return __double2float_ru(v);
#else
    float f = (float)v; if ((double)f < v) f = nextafterf(f, INFINITY); return f;
#endif
}
RT_HD RtBox6 rt_box_empty() { RtBox6 b; for (int a = 0; a < 3; ++a) { b.lo[a] = rt_inf(); b.hi[a] = -rt_inf(); } return b; }
RT_HD void rt_box_grow(RtBox6& b, const RtBox6& o) { for (int a = 0; a < 3; ++a) { b.lo[a] = fminf(b.lo[a], o.lo[a]); b.hi[a] = fmaxf(b.hi[a], o.hi[a]); } }
RT_HD void rt_box_grow_pt(RtBox6& b, double x, double y, double z) {
    b.lo[0] = fminf(b.lo[0], rt_f_down(x)); b.lo[1] = fminf(b.lo[1], rt_f_down(y)); b.lo[2] = fminf(b.lo[2], rt_f_down(z));
    b.hi[0] = fmaxf(b.hi[0], rt_f_up(x)); b.hi[1] = fmaxf(b.hi[1], rt_f_up(y)); b.hi[2] = fmaxf(b.hi[2], rt_f_up(z));
}
RT_HD bool rt_box_is_empty(const RtBox6& b) { return !(b.lo[0] <= b.hi[0]); }

// The two child boxes of binary node n.
RT_HD void rt_node_boxes(const rt_f4* nodes, int n, RtBox6& l, RtBox6& r) {
    const rt_f4* q = nodes + 4 * (size_t)n;
    rt_f4 q0 = q[0], q1 = q[1], q2 = q[2];
    l.lo[0] = q0.x; l.lo[1] = q0.y; l.lo[2] = q0.z; l.hi[0] = q0.w; l.hi[1] = q1.x; l.hi[2] = q1.y;
    r.lo[0] = q1.z; r.lo[1] = q1.w; r.lo[2] = q2.x; r.hi[0] = q2.y; r.hi[1] = q2.z; r.hi[2] = q2.w;
}
RT_HD RtBox6 rt_node_union(const rt_f4* nodes, int n) {
    RtBox6 l, r; rt_node_boxes(nodes, n, l, r);
    RtBox6 u = rt_box_empty();
    if (!rt_box_is_empty(l)) rt_box_grow(u, l);
    if (!rt_box_is_empty(r)) rt_box_grow(u, r);
    return u;
}

// Box of the primitives of a leaf (code < 0, not an instance), rounded outward, from the records the kernels intersect.
RT_HD RtBox6 rt_leaf_box(const RtSceneView& sc, int32_t leaf) {
    uint32_t x = ~(uint32_t)leaf;
    uint32_t kind = (x >> 28) & 3u, cnt = ((x >> 26) & 3u) + 1u, first = x & 0x03FFFFFFu;
    RtBox6 b = rt_box_empty();
    for (uint32_t k = 0; k < cnt; ++k) {
        uint32_t i = first + k;
        if (kind == RT_KIND_SPHERE) {                       // objects.rs:241-244, |r| so that a negative radius keeps a valid box
            rt_f4 s = sc.spheres[i];
            double r = fabs((double)s.w);
            rt_box_grow_pt(b, s.x - r, s.y - r, s.z - r); rt_box_grow_pt(b, s.x + r, s.y + r, s.z + r);
        } else if (kind == RT_KIND_MSPHERE) {               // objects.rs:328-335 over the times 0 and 1 (main.rs:531)
            rt_f4 m0 = sc.mspheres[3 * (size_t)i], m1 = sc.mspheres[3 * (size_t)i + 1], m2 = sc.mspheres[3 * (size_t)i + 2];
            double r = fabs((double)m0.w);
            for (int t = 0; t <= 1; ++t) {
                double s = ((double)t - (double)m1.w) * (double)m2.x;
                double c[3] = {m0.x + s * m1.x, m0.y + s * m1.y, m0.z + s * m1.z};
                // the kernels place the centre with fp32 multiply-adds: a few ulps of slack cover their rounding
                double e = 4e-7 * (fabs(c[0]) + fabs(c[1]) + fabs(c[2]) + r);
                rt_box_grow_pt(b, c[0] - r - e, c[1] - r - e, c[2] - r - e); rt_box_grow_pt(b, c[0] + r + e, c[1] + r + e, c[2] + r + e);
            }
        } else if (kind == RT_KIND_BOX) {                   // objects.rs:514-516
            rt_f4 b0 = sc.boxes[2 * (size_t)i], b1 = sc.boxes[2 * (size_t)i + 1];
            rt_box_grow_pt(b, b0.x, b0.y, b0.z); rt_box_grow_pt(b, b1.x, b1.y, b1.z);
        } else {                                            // objects.rs:413-429
            rt_f4 r0 = sc.rects[2 * (size_t)i], r1 = sc.rects[2 * (size_t)i + 1];
            const double off = 0.0001, k = r1.x;
            int type = f2i(r1.y);
            if (type == 0) { rt_box_grow_pt(b, r0.x, r0.z, k - off); rt_box_grow_pt(b, r0.y, r0.w, k + off); }
            else if (type == 1) { rt_box_grow_pt(b, r0.x, k - off, r0.z); rt_box_grow_pt(b, r0.y, k + off, r0.w); }
            else { rt_box_grow_pt(b, k - off, r0.x, r0.z); rt_box_grow_pt(b, k + off, r0.y, r0.w); }
        }
    }
    return b;
}
// World box of instance `ii` whose subtree has the object-space box `bx`: through the op chain, innermost first.
RT_HD RtBox6 rt_instance_world_box(const RtSceneView& sc, int ii, RtBox6 bx) {
    rt_f4 i1 = sc.instances[2 * (size_t)ii + 1];
    int first = f2i(i1.z), count = f2i(i1.w);
    for (int k = count - 1; k >= 0; --k) {
        rt_f4 op = sc.ops[first + k];
        RtBox6 r = rt_box_empty();
        if (f2i(op.w) == RT_OP_TRANSLATE) {
            rt_box_grow_pt(r, (double)bx.lo[0] + op.x, (double)bx.lo[1] + op.y, (double)bx.lo[2] + op.z);
            rt_box_grow_pt(r, (double)bx.hi[0] + op.x, (double)bx.hi[1] + op.y, (double)bx.hi[2] + op.z);
        } else {
            for (int c = 0; c < 8; ++c) {
                double x = (c & 1) ? bx.hi[0] : bx.lo[0], y = (c & 2) ? bx.hi[1] : bx.lo[1], z = (c & 4) ? bx.hi[2] : bx.lo[2];
                rt_box_grow_pt(r, (double)op.x * x + (double)op.y * z, y, -(double)op.y * x + (double)op.x * z);   // transrot.rs:72-73
            }
        }
        bx = r;
    }
    return bx;
}

// Pass `pass` (1, 2, ...) of the refit for binary node i.  done[i] = the pass in which node i was refitted (0 = not yet).
RT_HD void rt_refit_node(const RtSceneView& sc, rt_f4* nodes, int i, int pass, int* done) {
    if (done[i] != 0) return;
    RtBox6 old[2]; rt_node_boxes(nodes, i, old[0], old[1]);
    rt_f4 q3 = nodes[4 * (size_t)i + 3];
    int32_t code[2] = {f2i(q3.x), f2i(q3.y)};
    RtBox6 nb[2];
    for (int c = 0; c < 2; ++c) {
        if (rt_box_is_empty(old[c])) { nb[c] = old[c]; continue; }          // an absent child stays absent
        if (code[c] >= 0) {
            int f = done[code[c]];
            if (f == 0 || f >= pass) return;
            nb[c] = rt_node_union(nodes, code[c]);
        } else if (~(uint32_t)code[c] & RT_LEAF_INSTANCE_BIT) {
            int ii = (int)(~(uint32_t)code[c] & 0x3FFFFFFFu);
            int root = f2i(sc.instances[2 * (size_t)ii + 1].y);
            int f = done[root];
            if (f == 0 || f >= pass) return;
            nb[c] = rt_instance_world_box(sc, ii, rt_node_union(nodes, root));
        } else nb[c] = rt_leaf_box(sc, code[c]);
    }
    rt_f4* q = nodes + 4 * (size_t)i;
    rt_f4 q0, q1, q2;
    q0.x = nb[0].lo[0]; q0.y = nb[0].lo[1]; q0.z = nb[0].lo[2]; q0.w = nb[0].hi[0];
    q1.x = nb[0].hi[1]; q1.y = nb[0].hi[2]; q1.z = nb[1].lo[0]; q1.w = nb[1].lo[1];
    q2.x = nb[1].lo[2]; q2.y = nb[1].hi[0]; q2.z = nb[1].hi[1]; q2.w = nb[1].hi[2];
    q[0] = q0; q[1] = q1; q[2] = q2;
    done[i] = pass;
}

// ---- 4-wide quantised node: encoding of n <= 4 children (boxes + codes) into the four records described in wide_bvh.cpp.
RT_HD void rt_wide_encode(const RtBox6* c, const int32_t* code_in, int n, rt_f4 w[4]) {
    // (every loop over children or axes has a constant trip count and is unrolled: indexed by a variable, the arrays would live in
    // scratch memory on the GPU)
    float org[3]; double step[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float lo = rt_inf(), hi = -rt_inf();
#pragma unroll
        for (int k = 0; k < 4; ++k) if (k < n) { lo = fminf(lo, c[k].lo[a]); hi = fmaxf(hi, c[k].hi[a]); }
        if (n == 0) { lo = hi = 0.0f; }
        org[a] = lo;
        double ext = (double)hi - (double)lo;
        int e = ext > 0 ? (int)ceil(log2(ext / 255.0)) : -126;
        e = e < -126 ? -126 : (e > 127 ? 127 : e);
        while (ldexp(255.0, e) < ext && e < 127) ++e;      // log2's rounding must not leave the far plane outside the grid
        step[a] = ldexp(1.0, e);
    }
    uint32_t q[6] = {0, 0, 0, 0, 0, 0};                     // lo.x lo.y lo.z hi.x hi.y hi.z, child k in byte k
    int32_t code[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k >= n) {                                       // absent: an inverted box and a code no traversal enters
#pragma unroll
            for (int a = 0; a < 3; ++a) q[a] |= 255u << (8 * k);
            code[k] = RT_EMPTY_STACK;
            continue;
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            double lo = floor(((double)c[k].lo[a] - (double)org[a]) / step[a]);
            double hi = ceil(((double)c[k].hi[a] - (double)org[a]) / step[a]);
            lo = lo < 0.0 ? 0.0 : (lo > 255.0 ? 255.0 : lo); hi = hi < 0.0 ? 0.0 : (hi > 255.0 ? 255.0 : hi);
            q[a] |= (uint32_t)lo << (8 * k); q[3 + a] |= (uint32_t)hi << (8 * k);
        }
        code[k] = code_in[k];
    }
    w[0].x = org[0]; w[0].y = org[1]; w[0].z = org[2]; w[0].w = (float)step[0];
    w[1].x = i2f((int)q[0]); w[1].y = i2f((int)q[1]); w[1].z = i2f((int)q[2]); w[1].w = i2f((int)q[3]);
    w[2].x = i2f((int)q[4]); w[2].y = i2f((int)q[5]); w[2].z = i2f(code[0]); w[2].w = i2f(code[1]);
    w[3].x = i2f(code[2]); w[3].y = i2f(code[3]); w[3].z = (float)step[1]; w[3].w = (float)step[2];
}
// Re-encodes wide node i from the binary child boxes its children were taken from: src[4 i + k] = binary node << 1 | which
// child, or -1 for an absent child; the child codes stay as they are.
RT_HD void rt_wide_refit_node(const rt_f4* nodes, const int32_t* src, rt_f4* wnodes, int i) {
    rt_f4* w = wnodes + 4 * (size_t)i;
    rt_f4 w2 = w[2], w3 = w[3];
    int32_t old_code[4] = {f2i(w2.z), f2i(w2.w), f2i(w3.x), f2i(w3.y)};
    RtBox6 c[4]; int n = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {                                // present children are the first n slots (wide_bvh.cpp), so slot k stays slot k
        int32_t s = src[4 * (size_t)i + k];
        if (s < 0) continue;
        RtBox6 l, r; rt_node_boxes(nodes, s >> 1, l, r);
        c[k] = (s & 1) ? r : l; n = k + 1;
    }
    rt_f4 out[4];
    rt_wide_encode(c, old_code, n, out);
    w[0] = out[0]; w[1] = out[1]; w[2] = out[2]; w[3] = out[3];
}

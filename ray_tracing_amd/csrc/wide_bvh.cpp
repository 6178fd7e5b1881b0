// wide_bvh.cpp -- collapses the flattened scene's binary BVH (rt_types.h: 64-byte nodes holding two child boxes) into a
// 4-wide BVH with 8-bit quantised child boxes, also 64 bytes per node.  It replaces the traversal of src/bvh.rs:159-187 for
// scenes whose node array does not fit a CU's LDS (the 1 000 000-sphere stress scene): a ray then makes half as many
// dependent node fetches, each of the same size, and closest hits do not depend on the tree (ties aside).
//
// Wide node (4 x rt_f4):
//   w0 = (origin.x, origin.y, origin.z, step.x)        origin = min corner of the children's boxes; step = the grid step per
//                                                      axis, a power of two
//   w1 = 16 bytes: lo.x[4] lo.y[4] lo.z[4] hi.x[4]     child i's byte at bit 8 i of each word; plane = origin + byte * step,
//   w2 = (hi.y[4], hi.z[4], code 0, code 1)            lo rounded down and hi rounded up, so the box only grows
//   w3 = (code 2, code 3, step.y, step.z)
// A child code >= 0 is a wide node, < 0 a leaf code of rt_types.h (primitives or an instance), RT_EMPTY_STACK an absent child.
#if defined(__HIP__)
#include <hip/hip_runtime.h>   // hipcc compiles every source of its command line as HIP
#endif
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include "scene_builder.h"

namespace rt {
namespace {

struct ChildRef { float lo[3], hi[3]; int32_t code; };   // code: binary-tree child code (>= 0 inner node, < 0 leaf)

int32_t as_int(float f) { int32_t i; std::memcpy(&i, &f, 4); return i; }
float as_float(int32_t i) { float f; std::memcpy(&f, &i, 4); return f; }

void children_of(const std::vector<rt_f4>& nodes, int n, ChildRef out[2]) {
    const rt_f4* q = &nodes[(size_t)n * 4];
    out[0] = ChildRef{{q[0].x, q[0].y, q[0].z}, {q[0].w, q[1].x, q[1].y}, as_int(q[3].x)};
    out[1] = ChildRef{{q[1].z, q[1].w, q[2].x}, {q[2].y, q[2].z, q[2].w}, as_int(q[3].y)};
}
double area(const ChildRef& c) {
    double d[3] = {(double)c.hi[0] - c.lo[0], (double)c.hi[1] - c.lo[1], (double)c.hi[2] - c.lo[2]};
    if (!(d[0] >= 0 && d[1] >= 0 && d[2] >= 0)) return -1.0;   // an absent child (empty box)
    return d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
}

}  // namespace

// Appends the wide tree under binary node `root2` to f.wnodes (breadth-first, so that the top of the tree is contiguous);
// returns its root's wide index.  *stack_need = the deepest the traversal stack can get inside this tree when the nearest hit
// child is entered and the others are pushed: the largest sum over a root-to-leaf path of (children - 1).
static int collapse(FlatScene& f, int root2, int* stack_need) {
    struct Pending { int wide, node2; };
    std::vector<Pending> queue;
    const int first = (int)f.wnodes.size() / 4;
    auto alloc = [&](int node2) { int w = (int)f.wnodes.size() / 4; f.wnodes.resize(f.wnodes.size() + 4, rt_f4{0, 0, 0, 0}); queue.push_back({w, node2}); return w; };
    alloc(root2);
    std::vector<int> need_above;   // per wide node of this tree: sum of (children - 1) over its ancestors
    need_above.push_back(0);
    int need = 0;
    for (size_t head = 0; head < queue.size(); ++head) {
        const Pending cur = queue[head];
        ChildRef c[4]; int n = 0;
        {
            ChildRef two[2]; children_of(f.nodes, cur.node2, two);
            for (int k = 0; k < 2; ++k) if (area(two[k]) >= 0.0) c[n++] = two[k];
            if (n == 2 && c[0].code == c[1].code && c[0].code < 0) n = 1;   // a lone leaf referenced by both children of a root
        }
        while (n < 4) {   // open the inner child with the largest box until there are four children or only leaves
            int best = -1; double best_area = -1.0;
            for (int k = 0; k < n; ++k) if (c[k].code >= 0 && area(c[k]) > best_area) { best_area = area(c[k]); best = k; }
            if (best < 0) break;
            ChildRef two[2]; children_of(f.nodes, c[best].code, two);
            int m = 0; ChildRef keep[2];
            for (int k = 0; k < 2; ++k) if (area(two[k]) >= 0.0) keep[m++] = two[k];
            if (m == 0) { c[best] = c[--n]; continue; }
            c[best] = keep[0];
            if (m == 2) c[n++] = keep[1];
        }
        // quantisation frame
        float org[3]; double step[3];
        for (int a = 0; a < 3; ++a) {
            float lo = std::numeric_limits<float>::infinity(), hi = -lo;
            for (int k = 0; k < n; ++k) { lo = std::min(lo, c[k].lo[a]); hi = std::max(hi, c[k].hi[a]); }
            if (n == 0) { lo = hi = 0.0f; }
            org[a] = lo;
            double ext = (double)hi - (double)lo;
            int e = ext > 0 ? (int)std::ceil(std::log2(ext / 255.0)) : -126;
            e = std::max(-126, std::min(127, e));
            while (std::ldexp(255.0, e) < ext && e < 127) ++e;     // log2's rounding must not leave the far plane outside the grid
            step[a] = std::ldexp(1.0, e);
        }
        uint32_t q[6] = {0, 0, 0, 0, 0, 0};   // lo.x lo.y lo.z hi.x hi.y hi.z, child k in byte k
        int32_t code[4];
        for (int k = 0; k < 4; ++k) {
            if (k >= n) {   // absent: an inverted box and a code no traversal enters
                for (int a = 0; a < 3; ++a) q[a] |= 255u << (8 * k);
                code[k] = RT_EMPTY_STACK;
                continue;
            }
            for (int a = 0; a < 3; ++a) {
                double lo = std::floor(((double)c[k].lo[a] - (double)org[a]) / step[a]);
                double hi = std::ceil(((double)c[k].hi[a] - (double)org[a]) / step[a]);
                uint32_t ql = (uint32_t)std::max(0.0, std::min(255.0, lo)), qh = (uint32_t)std::max(0.0, std::min(255.0, hi));
                q[a] |= ql << (8 * k); q[3 + a] |= qh << (8 * k);
            }
            code[k] = c[k].code;
        }
        const int here = need_above[(size_t)(cur.wide - first)] + std::max(0, n - 1);
        need = std::max(need, here);
        for (int k = 0; k < n; ++k) if (code[k] >= 0) { code[k] = alloc(code[k]); need_above.push_back(here); }
        rt_f4* w = &f.wnodes[(size_t)cur.wide * 4];
        w[0] = rt_f4{org[0], org[1], org[2], (float)step[0]};
        w[1] = rt_f4{as_float((int32_t)q[0]), as_float((int32_t)q[1]), as_float((int32_t)q[2]), as_float((int32_t)q[3])};
        w[2] = rt_f4{as_float((int32_t)q[4]), as_float((int32_t)q[5]), as_float(code[0]), as_float(code[1])};
        w[3] = rt_f4{as_float(code[2]), as_float(code[3]), (float)step[1], (float)step[2]};
    }
    *stack_need = need;
    return first;
}

// Builds f.wnodes / f.winst_roots / f.wroot / f.wmax_stack from the finished binary tree (top level first, so that its top
// levels are the first nodes of the array; then every instance's tree).
void build_wide_bvh(FlatScene& f) {
    f.wnodes.clear(); f.winst_roots.clear(); f.wroot = -1; f.wmax_stack = 0;
    if (f.root < 0) return;
    int top_need = 0, blas_need = 0;
    f.wroot = collapse(f, f.root, &top_need);
    const size_t n_inst = f.instances.size() / RT_INSTANCE_F4;
    for (size_t i = 0; i < n_inst; ++i) {
        int32_t root2 = as_int(f.instances[i * RT_INSTANCE_F4 + 1].y);
        int need = 0;
        f.winst_roots.push_back(root2 >= 0 ? collapse(f, root2, &need) : -1);
        blas_need = std::max(blas_need, need);
    }
    f.wmax_stack = top_need + (n_inst ? 1 + blas_need : 0);   // + the sentinel pushed on entering an instance
}

}  // namespace rt

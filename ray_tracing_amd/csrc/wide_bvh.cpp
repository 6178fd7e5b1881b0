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
#include "refit.h"

namespace rt {
namespace {

struct ChildRef { float lo[3], hi[3]; int32_t code; int32_t src; };   // code: binary-tree child code (>= 0 inner node, < 0 leaf); src: binary node << 1 | which child

int32_t as_int(float f) { int32_t i; std::memcpy(&i, &f, 4); return i; }
float as_float(int32_t i) { float f; std::memcpy(&f, &i, 4); return f; }

void children_of(const std::vector<rt_f4>& nodes, int n, ChildRef out[2]) {
    const rt_f4* q = &nodes[(size_t)n * 4];
    out[0] = ChildRef{{q[0].x, q[0].y, q[0].z}, {q[0].w, q[1].x, q[1].y}, as_int(q[3].x), n << 1};
    out[1] = ChildRef{{q[1].z, q[1].w, q[2].x}, {q[2].y, q[2].z, q[2].w}, as_int(q[3].y), n << 1 | 1};
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
    auto alloc = [&](int node2) {
        int w = (int)f.wnodes.size() / 4;
        f.wnodes.resize(f.wnodes.size() + 4, rt_f4{0, 0, 0, 0}); f.wsrc.resize(f.wsrc.size() + 4, -1);
        queue.push_back({w, node2});
        return w;
    };
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
        RtBox6 boxes[4]; int32_t code[4];
        for (int k = 0; k < n; ++k) {
            for (int a = 0; a < 3; ++a) { boxes[k].lo[a] = c[k].lo[a]; boxes[k].hi[a] = c[k].hi[a]; }
            code[k] = c[k].code;
        }
        const int here = need_above[(size_t)(cur.wide - first)] + std::max(0, n - 1);
        need = std::max(need, here);
        for (int k = 0; k < n; ++k) if (code[k] >= 0) { code[k] = alloc(code[k]); need_above.push_back(here); }
        rt_f4 w[4];
        rt_wide_encode(boxes, code, n, w);                  // the same encoder the refit kernels use
        for (int k = 0; k < 4; ++k) { f.wnodes[(size_t)cur.wide * 4 + k] = w[k]; f.wsrc[(size_t)cur.wide * 4 + k] = k < n ? c[k].src : -1; }
    }
    *stack_need = need;
    return first;
}

// Builds f.wnodes / f.winst_roots / f.wroot / f.wmax_stack from the finished binary tree (top level first, so that its top
// levels are the first nodes of the array; then every instance's tree).
void build_wide_bvh(FlatScene& f) {
    f.wnodes.clear(); f.wsrc.clear(); f.winst_roots.clear(); f.wroot = -1; f.wmax_stack = 0;
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

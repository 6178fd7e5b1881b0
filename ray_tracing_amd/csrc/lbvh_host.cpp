// lbvh_host.cpp -- the linear top-level BVH on the host: the very steps the GPU builder runs (lbvh.h), in plain loops.
// Used by rt_flat_create_ex (inspection and CPU-side checks of the builder) and as the reference for the device build.
#if defined(__HIP__)
#include <hip/hip_runtime.h>   // hipcc compiles every source of its command line as HIP
#endif
#include <algorithm>
#include <cmath>
#include <cstring>
#include "scene_builder.h"
#include "lbvh.h"

namespace rt {

// Frame of the Morton codes: per axis the range of the item centres, cut to mean +- 3 standard deviations.  One far-away
// centre (a ground sphere of radius 1e5) would otherwise stretch an axis until the codes stop resolving the rest of the
// scene along it; centres outside the frame fall into the first or last cell.
LbvhFrameHost lbvh_frame(const FlatScene& f) {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, sum[3] = {0, 0, 0}, sq[3] = {0, 0, 0};
    const size_t n = f.tlas_items.size() / 2;
    for (size_t i = 0; i < n; ++i) {
        const rt_f4 &a = f.tlas_items[2 * i], &b = f.tlas_items[2 * i + 1];
        double c[3] = {0.5 * ((double)a.x + b.x), 0.5 * ((double)a.y + b.y), 0.5 * ((double)a.z + b.z)};
        for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], c[k]); hi[k] = std::max(hi[k], c[k]); sum[k] += c[k]; sq[k] += c[k] * c[k]; }
    }
    LbvhFrameHost fr;
    for (int k = 0; k < 3; ++k) {
        double mean = n ? sum[k] / (double)n : 0.0, var = n ? std::max(0.0, sq[k] / (double)n - mean * mean) : 0.0, sd = std::sqrt(var);
        double l = std::max(lo[k], mean - 3.0 * sd), h = std::min(hi[k], mean + 3.0 * sd);
        fr.lo[k] = (float)l; fr.inv_ext[k] = h > l ? (float)(1.0 / (h - l)) : 0.0f;
    }
    return fr;
}

bool lbvh_finish_host(FlatScene& f, std::string& err) {
    const int n = (int)f.tlas_items.size() / 2;
    if (n < 2) { err = "lbvh: nothing deferred"; return false; }
    LbvhFrameHost h = lbvh_frame(f);
    RtLbvhFrame fr;
    for (int a = 0; a < 3; ++a) { fr.lo[a] = h.lo[a]; fr.inv_ext[a] = h.inv_ext[a]; }
    std::vector<uint64_t> code((size_t)n), keys((size_t)n);
    std::vector<uint32_t> order((size_t)n);
    for (int i = 0; i < n; ++i) { code[i] = rt_lbvh_key(fr, f.tlas_items[2 * (size_t)i], f.tlas_items[2 * (size_t)i + 1]); order[i] = (uint32_t)i; }
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return code[a] < code[b]; });   // as a radix sort: stable
    for (int i = 0; i < n; ++i) keys[i] = code[order[i]];
    std::vector<int> left((size_t)n - 1), right((size_t)n - 1), depth((size_t)n - 1, 0), filled((size_t)n - 1, 0);
    std::vector<RtLbvhBox> box((size_t)n - 1);
    for (int i = 0; i < n - 1; ++i) rt_lbvh_node(keys.data(), n, i, left.data(), right.data());
    const int base = (int)f.nodes.size() / 4;
    f.nodes.resize(f.nodes.size() + 4 * ((size_t)n - 1));
    rt_f4* tail = f.nodes.data() + 4 * (size_t)base;
    for (int pass = 1; filled[0] == 0; ++pass) {
        if (pass > 104) { err = "lbvh: fill did not reach the root"; return false; }
        for (int i = 0; i < n - 1; ++i)
            rt_lbvh_fill(i, pass, left.data(), right.data(), order.data(), f.tlas_items.data(), box.data(), depth.data(), filled.data(), tail, base);
    }
    f.root = base; f.tlas_depth = depth[0];
    f.max_stack += f.tlas_depth;
    if (f.max_stack > RT_STACK_DEPTH) { err = "flatten: linear BVH deeper than the traversal stack"; return false; }
    return true;
}

}  // namespace rt

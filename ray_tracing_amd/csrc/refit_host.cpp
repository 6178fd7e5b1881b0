// refit_host.cpp -- the refit of refit.h in plain loops on the host (rt_flat_refit: inspection and CPU-side checks; the GPU
// runs the same per-element steps as kernels, refit.hip), and the lookup of a primitive record by object id.
#if defined(__HIP__)
#include <hip/hip_runtime.h>   // hipcc compiles every source of its command line as HIP
#endif
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>
#include "scene_builder.h"
#include "refit.h"

namespace rt {

// of_id[kind][object id] = index of that sphere (kind 0) / moving sphere (kind 1) in its record array; -1: no such primitive;
// -2: the primitive only bounds a constant medium (such primitives are in no tree: the medium record holds a copy).
void build_prim_index(const FlatScene& f, std::vector<int32_t> of_id[2]) {
    const std::vector<rt_i2>* meta[2] = {&f.sphere_meta, &f.msphere_meta};
    std::vector<char> in_tree[2];
    for (int k = 0; k < 2; ++k) { of_id[k].assign((size_t)std::max(0, f.n_ids), -1); in_tree[k].assign(meta[k]->size(), 0); }
    auto mark = [&](int32_t code) {
        if (code >= 0) return;
        uint32_t x = ~(uint32_t)code;
        if (x & RT_LEAF_INSTANCE_BIT) return;
        int k = (int)((x >> 28) & 3u), cnt = (int)((x >> 26) & 3u) + 1, first = (int)(x & 0x03FFFFFFu);
        if (k > 1) return;
        for (int i = first; i < first + cnt && i < (int)in_tree[k].size(); ++i) in_tree[k][(size_t)i] = 1;
    };
    const size_t n_nodes = f.nodes.size() / 4;
    for (size_t n = 0; n < n_nodes; ++n) {
        int32_t kids[2]; std::memcpy(&kids[0], &f.nodes[n * 4 + 3].x, 4); std::memcpy(&kids[1], &f.nodes[n * 4 + 3].y, 4);
        mark(kids[0]); mark(kids[1]);
    }
    for (size_t i = 0; i + 1 < f.tlas_items.size(); i += 2) { int32_t code; std::memcpy(&code, &f.tlas_items[i].w, 4); mark(code); }
    for (int k = 0; k < 2; ++k)
        for (size_t i = 0; i < meta[k]->size(); ++i) {
            int id = (*meta[k])[i].y;
            if (id >= 0 && id < (int)of_id[k].size()) of_id[k][(size_t)id] = in_tree[k][i] ? (int32_t)i : -2;
        }
}

// Writes new parameters into the host records of spheres (kind 0: xyzr = centre, radius) or moving spheres (kind 1: 9 values =
// centre0, centre1, time0, time1, radius, as MovingSphere::new takes them, src/objects.rs:266).  touched = the record indices.
bool set_prims_host(FlatScene& f, const std::vector<int32_t> of_id[2], int kind, const int32_t* ids, int n, const double* values,
                    std::vector<int32_t>& touched, std::string& err) {
    touched.clear();
    // every id and value is checked before any record is written, so that a failing call changes nothing
    std::vector<int32_t> recs((size_t)n);
    for (int j = 0; j < n; ++j) {
        int id = ids[j];
        int32_t rec = (id >= 0 && id < (int)of_id[kind].size()) ? of_id[kind][(size_t)id] : -1;
        if (rec == -2) { err = "flatten: object " + std::to_string(id) + " bounds a constant medium and cannot be moved"; return false; }
        if (rec < 0) { err = "object " + std::to_string(id) + " is not a " + (kind == 0 ? "sphere" : "moving sphere") + " of this scene"; return false; }
        if (kind == 1 && !(values[9 * (size_t)j + 7] != values[9 * (size_t)j + 6])) { err = "moving sphere with time0 == time1"; return false; }
        recs[(size_t)j] = rec;
    }
    for (int j = 0; j < n; ++j) {
        const int32_t rec = recs[(size_t)j];
        if (kind == 0) {
            const double* v = values + 4 * (size_t)j;
            f.spheres[(size_t)rec] = rt_f4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        } else {
            const double* v = values + 9 * (size_t)j;
            f.mspheres[3 * (size_t)rec] = rt_f4{(float)v[0], (float)v[1], (float)v[2], (float)v[8]};
            f.mspheres[3 * (size_t)rec + 1] = rt_f4{(float)(v[3] - v[0]), (float)(v[4] - v[1]), (float)(v[5] - v[2]), (float)v[6]};
            f.mspheres[3 * (size_t)rec + 2] = rt_f4{(float)(1.0 / (v[7] - v[6])), 0, 0, 0};
        }
        touched.push_back(rec);
    }
    // A moved primitive may enter or leave a medium's boundary: the lists of surfaces inside the boundaries (flatten.cpp,
    // emit_medium_locals) no longer hold, and every ray goes through the tree again.
    for (size_t m = 0; m * RT_MEDIUM_F4 < f.media.size(); ++m) {
        f.media[m * RT_MEDIUM_F4 + 5].z = 0.0f;   // count | mask << 8
        int32_t flags; std::memcpy(&flags, &f.media[m * RT_MEDIUM_F4 + 5].y, 4);
        flags &= 1;                               // ... nor the claim that a boundary encloses everything else
        std::memcpy(&f.media[m * RT_MEDIUM_F4 + 5].y, &flags, 4);
    }
    return true;
}

bool refit_host(FlatScene& f, std::string& err) {
    const int n = (int)f.nodes.size() / 4;
    if (n == 0 || f.root < 0) return true;
    RtSceneView v = f.view();
    std::vector<int> done((size_t)n, 0);
    const int passes = f.tlas_depth + f.blas_depth + 2;
    for (int pass = 1; pass <= passes; ++pass)
        for (int i = 0; i < n; ++i) rt_refit_node(v, f.nodes.data(), i, pass, done.data());
    if (!done[f.root]) { err = "refit: the passes did not reach the root"; return false; }
    const int nw = (int)f.wnodes.size() / 4;
    for (int i = 0; i < nw; ++i) rt_wide_refit_node(f.nodes.data(), f.wsrc.data(), f.wnodes.data(), i);
    return true;
}

}  // namespace rt

// capi_internal.h -- definitions of the opaque handles of include/rt_hip.h shared by capi.cpp and render.hip.
#pragma once
#include <string>
#include <vector>
#include "../../include/rt_hip.h"
#include "scene_builder.h"

struct RtBuilder { rt::Builder b; };
struct RtFlat { rt::FlatScene flat; RtSceneView view; std::vector<int32_t> of_id[2]; bool indexed = false; };

extern thread_local std::string g_rt_error;
int rt_set_error(int code, const std::string& msg);

void rt_fill_info(const rt::FlatScene& f, RtSceneInfo* out);   // capi.cpp

// Status code of a flattener message.
inline int rt_status_of(const std::string& msg) {
    if (msg.find("No bounding box") != std::string::npos) return RT_ERR_NO_BBOX;
    if (msg.find("flatten:") != std::string::npos) return RT_ERR_UNSUPPORTED;
    return RT_ERR_INVALID;
}

namespace rt {
// lbvh.hip: the n - 1 nodes over n >= 2 deferred top-level items, built on the current device (see lbvh.h).
bool lbvh_build_device(const float frame_lo[3], const float frame_inv_ext[3], const rt_f4* d_items, int n, rt_f4* d_nodes_tail, int node_base,
                       int* depth_out, double* seconds_out, std::string& err);
// refit.hip: recomputes the boxes of the binary and the wide tree of a scene resident on the current device (see refit.h).
bool refit_device(const RtSceneView& view, const int32_t* d_wsrc, int root, int passes, double* seconds_out, std::string& err);
}

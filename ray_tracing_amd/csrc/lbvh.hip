// lbvh.hip -- the linear top-level BVH built on the GPU (gfx950): keys, radix sort (rocPRIM), radix-tree nodes, fill passes.
// See lbvh.h for the algorithm; every kernel here is one of its per-element steps over a grid.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <string>
#include "capi_internal.h"
#include "lbvh.h"

#define LB_BLOCK 256
#define LB_TRY(expr)                                                                                           \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); goto fail; }          \
    } while (0)

__global__ __launch_bounds__(LB_BLOCK) void k_lbvh_keys(RtLbvhFrame fr, const rt_f4* items, int n, uint64_t* keys, uint32_t* order) {
    int i = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (i < n) { keys[i] = rt_lbvh_key(fr, items[2 * (size_t)i], items[2 * (size_t)i + 1]); order[i] = (uint32_t)i; }
}
__global__ __launch_bounds__(LB_BLOCK) void k_lbvh_nodes(const uint64_t* keys, int n, int* left, int* right) {
    int i = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (i < n - 1) rt_lbvh_node(keys, n, i, left, right);
}
__global__ __launch_bounds__(LB_BLOCK) void k_lbvh_fill(int n, int pass, const int* left, const int* right, const uint32_t* order,
                                                        const rt_f4* items, RtLbvhBox* box, int* depth, int* filled, rt_f4* nodes, int node_base) {
    int i = blockIdx.x * LB_BLOCK + threadIdx.x;
    if (i < n - 1) rt_lbvh_fill(i, pass, left, right, order, items, box, depth, filled, nodes, node_base);
}

namespace rt {

// Builds the n - 1 nodes over the n >= 2 items (device pointer, 2 records each) into nodes_tail (device, 4 records per
// node); inner child indices are offset by node_base.  Returns false with a message on failure; *depth_out = height.
bool lbvh_build_device(const float frame_lo[3], const float frame_inv_ext[3], const rt_f4* d_items, int n, rt_f4* d_nodes_tail, int node_base,
                       int* depth_out, double* seconds_out, std::string& err) {
    RtLbvhFrame fr;
    for (int a = 0; a < 3; ++a) { fr.lo[a] = frame_lo[a]; fr.inv_ext[a] = frame_inv_ext[a]; }
    uint64_t *keys_in = nullptr, *keys = nullptr;
    uint32_t *order_in = nullptr, *order = nullptr;
    int *left = nullptr, *right = nullptr, *depth = nullptr, *filled = nullptr;
    RtLbvhBox* box = nullptr;
    void* temp = nullptr; size_t temp_bytes = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const int grid = (n + LB_BLOCK - 1) / LB_BLOCK;
    int root_filled = 0, root_depth = 0;
    bool ok = false;
    LB_TRY(hipMalloc((void**)&keys_in, (size_t)n * sizeof(uint64_t)));
    LB_TRY(hipMalloc((void**)&keys, (size_t)n * sizeof(uint64_t)));
    LB_TRY(hipMalloc((void**)&order_in, (size_t)n * sizeof(uint32_t)));
    LB_TRY(hipMalloc((void**)&order, (size_t)n * sizeof(uint32_t)));
    LB_TRY(hipMalloc((void**)&left, (size_t)n * sizeof(int)));
    LB_TRY(hipMalloc((void**)&right, (size_t)n * sizeof(int)));
    LB_TRY(hipMalloc((void**)&depth, (size_t)n * sizeof(int)));
    LB_TRY(hipMalloc((void**)&filled, (size_t)n * sizeof(int)));
    LB_TRY(hipMalloc((void**)&box, (size_t)n * sizeof(RtLbvhBox)));
    LB_TRY(rocprim::radix_sort_pairs(nullptr, temp_bytes, keys_in, keys, order_in, order, (size_t)n, 0, 63, (hipStream_t)0));
    LB_TRY(hipMalloc(&temp, temp_bytes > 0 ? temp_bytes : 16));
    LB_TRY(hipEventCreate(&e0)); LB_TRY(hipEventCreate(&e1));
    LB_TRY(hipMemsetAsync(filled, 0, (size_t)n * sizeof(int), 0));
    LB_TRY(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_lbvh_keys, dim3(grid), dim3(LB_BLOCK), 0, 0, fr, d_items, n, keys_in, order_in);
    LB_TRY(rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys, order_in, order, (size_t)n, 0, 63, (hipStream_t)0));
    hipLaunchKernelGGL(k_lbvh_nodes, dim3(grid), dim3(LB_BLOCK), 0, 0, keys, n, left, right);
    LB_TRY(hipGetLastError());
    // one launch per level of the tree, checked for the root every 8 passes (63 code bits + up to 32 tie-break bits)
    for (int pass = 1; pass <= 104 && !root_filled; ++pass) {
        hipLaunchKernelGGL(k_lbvh_fill, dim3(grid), dim3(LB_BLOCK), 0, 0, n, pass, left, right, order, d_items, box, depth, filled, d_nodes_tail, node_base);
        if (pass % 8 == 0) { LB_TRY(hipGetLastError()); LB_TRY(hipMemcpy(&root_filled, filled, sizeof(int), hipMemcpyDeviceToHost)); }
    }
    LB_TRY(hipMemcpy(&root_filled, filled, sizeof(int), hipMemcpyDeviceToHost));
    if (!root_filled) { err = "lbvh: fill did not reach the root"; goto fail; }
    LB_TRY(hipEventRecord(e1, 0));
    LB_TRY(hipEventSynchronize(e1));
    LB_TRY(hipMemcpy(&root_depth, depth, sizeof(int), hipMemcpyDeviceToHost));
    { float ms = 0.0f; LB_TRY(hipEventElapsedTime(&ms, e0, e1)); if (seconds_out) *seconds_out = ms * 1e-3; }
    *depth_out = root_depth;
    ok = true;
fail:
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    void* bufs[] = {keys_in, keys, order_in, order, left, right, depth, filled, box, temp};
    for (void* b : bufs) if (b) (void)hipFree(b);
    return ok;
}

}  // namespace rt

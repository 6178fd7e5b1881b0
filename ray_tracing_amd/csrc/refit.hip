// refit.hip -- the refit of refit.h as gfx950 kernels: one launch per pass over the binary nodes, then one over the wide nodes.
#include <hip/hip_runtime.h>
#include <string>
#include "capi_internal.h"
#include "refit.h"

#define RF_BLOCK 256

__global__ __launch_bounds__(RF_BLOCK) void k_refit_pass(RtSceneView sc, rt_f4* nodes, int n, int pass, int* done) {
    int i = blockIdx.x * RF_BLOCK + threadIdx.x;
    if (i < n) rt_refit_node(sc, nodes, i, pass, done);
}
__global__ __launch_bounds__(RF_BLOCK) void k_refit_wide(const rt_f4* nodes, const int32_t* src, rt_f4* wnodes, int n) {
    int i = blockIdx.x * RF_BLOCK + threadIdx.x;
    if (i < n) rt_wide_refit_node(nodes, src, wnodes, i);
}

namespace rt {

// Refits the trees of a scene resident on the current device.  `view` holds device pointers; d_wsrc = the wide nodes' sources
// (device).  `passes` = height of the deepest dependency chain (top-level depth + instance subtree depth + 2).
bool refit_device(const RtSceneView& view, const int32_t* d_wsrc, int root, int passes, double* seconds_out, std::string& err) {
    const int n = view.n_nodes;
    if (n == 0 || root < 0) return true;
    int* done = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ok = false;
    int root_done = 0;
    hipError_t e;
#define RF_TRY(expr) do { e = (expr); if (e != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e); goto fail; } } while (0)
    RF_TRY(hipMalloc((void**)&done, (size_t)n * sizeof(int)));
    RF_TRY(hipEventCreate(&e0)); RF_TRY(hipEventCreate(&e1));
    RF_TRY(hipMemsetAsync(done, 0, (size_t)n * sizeof(int), 0));
    RF_TRY(hipEventRecord(e0, 0));
    for (int pass = 1; pass <= passes; ++pass)
        hipLaunchKernelGGL(k_refit_pass, dim3((n + RF_BLOCK - 1) / RF_BLOCK), dim3(RF_BLOCK), 0, 0, view, const_cast<rt_f4*>(view.nodes), n, pass, done);
    if (view.n_wnodes > 0)
        hipLaunchKernelGGL(k_refit_wide, dim3((view.n_wnodes + RF_BLOCK - 1) / RF_BLOCK), dim3(RF_BLOCK), 0, 0, view.nodes, d_wsrc,
                           const_cast<rt_f4*>(view.wnodes), view.n_wnodes);
    RF_TRY(hipGetLastError());
    RF_TRY(hipEventRecord(e1, 0));
    RF_TRY(hipEventSynchronize(e1));
    RF_TRY(hipMemcpy(&root_done, done + root, sizeof(int), hipMemcpyDeviceToHost));
    if (!root_done) { err = "refit: the passes did not reach the root"; goto fail; }
    { float ms = 0.0f; RF_TRY(hipEventElapsedTime(&ms, e0, e1)); if (seconds_out) *seconds_out = ms * 1e-3; }
    ok = true;
fail:
#undef RF_TRY
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (done) (void)hipFree(done);
    return ok;
}

}  // namespace rt

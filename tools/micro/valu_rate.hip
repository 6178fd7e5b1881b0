// valu_rate.hip -- issue cost of some gfx950 vector instructions: every wave runs a long run of independent instructions
// of one kind; 4 waves per SIMD on every CU.  Prints SIMD cycles per wave-instruction.
// build: hipcc -O2 --offload-arch=gfx950 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(S) S S S S S S S S
#define BODY(NAME, ASM, TYPE, CONS)                                                                      \
    __global__ void __launch_bounds__(256) NAME(float* out, int iters) {                                  \
        TYPE a0 = (TYPE)threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = (TYPE)1.0001f;             \
        for (int i = 0; i < iters; ++i) {                                                                 \
            REP8(asm volatile(ASM : "+" CONS(a0) : CONS(b));                                               \
                 asm volatile(ASM : "+" CONS(a1) : CONS(b));                                               \
                 asm volatile(ASM : "+" CONS(a2) : CONS(b));                                               \
                 asm volatile(ASM : "+" CONS(a3) : CONS(b));)                                              \
        }                                                                                                 \
        if (iters < 0) out[threadIdx.x] = (float)((double)a0 + (double)a1 + (double)a2 + (double)a3);     \
    }
#define V(x) "v"(x)
typedef double f64;
__device__ inline float tof(f2 v) { return v.x + v.y; }
BODY(k_mul, "v_mul_f32 %0, %0, %1", float, V)
BODY(k_fma, "v_fma_f32 %0, %0, %1, %1", float, V)
BODY(k_min, "v_min_f32 %0, %0, %1", float, V)
BODY(k_max3, "v_max3_f32 %0, %0, %1, %1", float, V)
BODY(k_fma64, "v_fma_f64 %0, %0, %1, %1", f64, V)
BODY(k_add64, "v_add_f64 %0, %0, %1", f64, V)
BODY(k_mul64, "v_mul_f64 %0, %0, %1", f64, V)

#define BODY2(NAME, ASM)                                                                                  \
    __global__ void __launch_bounds__(256) NAME(float* out, int iters) {                                  \
        f2 a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, b = {1.0001f, 0.9999f}; \
        for (int i = 0; i < iters; ++i) {                                                                 \
            REP8(asm volatile(ASM : "+v"(a0) : "v"(b));                                                    \
                 asm volatile(ASM : "+v"(a1) : "v"(b));                                                    \
                 asm volatile(ASM : "+v"(a2) : "v"(b));                                                    \
                 asm volatile(ASM : "+v"(a3) : "v"(b));)                                                   \
        }                                                                                                 \
        if (iters < 0) out[threadIdx.x] = tof(a0) + tof(a1) + tof(a2) + tof(a3);                           \
    }
BODY2(k_pkmul, "v_pk_mul_f32 %0, %0, %1")
BODY2(k_pkadd, "v_pk_add_f32 %0, %0, %1")
BODY2(k_pkfma, "v_pk_fma_f32 %0, %0, %1, %1")

// integer and transcendental forms (Philox rounds, reciprocals, square roots, logarithms of the shading kernel)
#define BODYU(NAME, ASM)                                                                                  \
    __global__ void __launch_bounds__(256) NAME(float* out, int iters) {                                  \
        unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 0xD2511F53u;                 \
        for (int i = 0; i < iters; ++i) {                                                                 \
            REP8(asm volatile(ASM : "+v"(a0) : "v"(b));                                                    \
                 asm volatile(ASM : "+v"(a1) : "v"(b));                                                    \
                 asm volatile(ASM : "+v"(a2) : "v"(b));                                                    \
                 asm volatile(ASM : "+v"(a3) : "v"(b));)                                                   \
        }                                                                                                 \
        if (iters < 0) out[threadIdx.x] = (float)(a0 + a1 + a2 + a3);                                     \
    }
BODYU(k_mulhi, "v_mul_hi_u32 %0, %0, %1")
BODYU(k_mullo, "v_mul_lo_u32 %0, %0, %1")
BODYU(k_xor, "v_xor_b32 %0, %0, %1")
BODYU(k_rcp, "v_rcp_f32 %0, %0")
BODYU(k_sqrt, "v_sqrt_f32 %0, %0")
BODYU(k_log, "v_log_f32 %0, %0")
BODYU(k_sin, "v_sin_f32 %0, %0")
__global__ void __launch_bounds__(256) k_mad64(float* out, int iters) {
    unsigned long long a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, c;
    unsigned b = 0xD2511F53u, x = threadIdx.x | 1u;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(a0), "=s"(c) : "v"(b), "v"(x));
             asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(a1), "=s"(c) : "v"(b), "v"(x));
             asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(a2), "=s"(c) : "v"(b), "v"(x));
             asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(a3), "=s"(c) : "v"(b), "v"(x));)
    }
    if (iters < 0) out[threadIdx.x] = (float)(a0 + a1 + a2 + a3);
}
__global__ void __launch_bounds__(256) k_cvt64(float* out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3; double d;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(a0));
             asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(a1));
             asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(a2));
             asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(a3));)
    }
    if (iters < 0) out[threadIdx.x] = (float)d;
}

template <class K> int run(const char* name, K k, float* out, double ghz) {
    const int iters = 65536, per_iter = 32, grid = 256 * 4;   // 4 blocks of 4 waves per CU = 4 waves per SIMD
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, iters);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, iters);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double insts_per_simd = 4.0 * iters * per_iter;           // wave-instructions issued on one SIMD
    printf("%-12s %8.3f ms  %.2f cycles per wave-instruction (at %.2f GHz)\n", name, ms, ms * 1e-3 * ghz * 1e9 / insts_per_simd, ghz);
    return 0;
}
int main() {
    float* out; CHECK(hipMalloc(&out, 4096));
    int khz = 0; CHECK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    double ghz = khz * 1e-6;
    for (int rep = 0; rep < 2; ++rep) {
    run("mul", k_mul, out, ghz); run("fma", k_fma, out, ghz); run("min", k_min, out, ghz); run("max3", k_max3, out, ghz);
    run("pk_mul", k_pkmul, out, ghz); run("pk_add", k_pkadd, out, ghz); run("pk_fma", k_pkfma, out, ghz);
    run("fma_f64", k_fma64, out, ghz); run("add_f64", k_add64, out, ghz); run("mul_f64", k_mul64, out, ghz);
    run("mul_hi_u32", k_mulhi, out, ghz); run("mul_lo_u32", k_mullo, out, ghz); run("mad_u64_u32", k_mad64, out, ghz); run("xor", k_xor, out, ghz);
    run("rcp", k_rcp, out, ghz); run("sqrt", k_sqrt, out, ghz); run("log", k_log, out, ghz); run("sin", k_sin, out, ghz); run("cvt_f64_f32", k_cvt64, out, ghz); }
    return 0;
}

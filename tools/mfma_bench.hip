// Dev probe: f64 MFMA issue rate on gfx950 (v_mfma_f64_16x16x4_f64), 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void k(double* out, int iters) {
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ __launch_bounds__(1024) void kf(double* out, int iters) {
    double c[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int q = 0; q < 8; ++q) c[q] = __builtin_fma(a, b, c[q]);
    double s = 0;
    for (int q = 0; q < 8; ++q) s += c[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    double* d; hipMalloc(&d, 256 * 1024 * 8 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int threads : {256, 512, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double nm = 256.0 * (threads / 64) * iters * 4;
            if (rep) printf("mfma f64: %d waves/SIMD: %.3f ms, %.1f TFLOP/s, %.1f ns per MFMA per SIMD\n", threads / 256, ms, nm * 2048 / ms / 1e9, ms * 1e6 / (nm / 1024));
        }
    }
    for (int threads : {256, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kf, dim3(256), dim3(threads), 0, 0, d, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double nf = 256.0 * threads * iters * 8;
            if (rep) printf("valu fma f64: %d waves/SIMD: %.3f ms, %.1f TFLOP/s\n", threads / 256, ms, nf * 2 / ms / 1e9);
        }
    }
    return 0;
}

// panel_bench.hip -- developer probe: cost of one 16-column block column factorisation by one wave
// (bcr_panel_factor and experimental alternatives), alone on its CU.  s_memrealtime (100 MHz) around R repeats.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I bpldenoising_amd/csrc tools/panel_bench.hip -o tools/_bin/panel_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "adjoint_bcr_kernels.hpp"
using namespace bpltv;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

template <int VAR>
__global__ __launch_bounds__(64) void bench(const double* __restrict__ A, int MP, int R, long long* __restrict__ out,
                                            double* __restrict__ sink) {
    extern __shared__ double S[];
    const int ld = MP + 1, lane = threadIdx.x;
    double* dinv = S + (size_t)ld * MP;
    double* S2 = dinv + MP;   // pristine copy
    for (int e = lane; e < MP * MP; e += 64) { S2[(e % MP) + ld * (e / MP)] = A[e]; }
    __syncthreads();
    long long t = 0;
    bool bad = false;
    for (int rep = 0; rep < R; ++rep) {
        for (int e = lane; e < MP * 16; e += 64) S[(e % MP) + ld * (e / MP)] = S2[(e % MP) + ld * (e / MP)];
        __syncthreads();
        const long long t0 = wall_clock64();
        if (VAR == 0) bad |= bcr_panel_factor<false>(S, ld, MP, 0, lane, dinv);
        if (VAR == 1) bad |= bcr_panel_factor<true>(S, ld, MP, 0, lane, dinv);
        __syncthreads();
        t += wall_clock64() - t0;
    }
    if (lane == 0) { out[0] = t; out[1] = bad; }
    sink[lane] = S[lane + ld * 3] + dinv[lane & 15];
}

int main() {
    const int MP = 64 + 64;
    std::vector<double> A((size_t)MP * MP);
    for (int r = 0; r < MP; ++r)
        for (int c = 0; c < MP; ++c) A[r + (size_t)MP * c] = (r == c) ? 40.0 : 1.0 / (1.0 + abs(r - c));
    double *dA, *sink; long long* dout;
    CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&sink, 64 * 8)); CK(hipMalloc(&dout, 16));
    CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
    const int R = 200;
    const size_t lds = ((size_t)(MP + 1) * MP + MP + (size_t)(MP + 1) * 16) * 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&bench<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&bench<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int var = 0; var < 2; ++var) {
        for (int rep = 0; rep < 2; ++rep) {
            if (var == 0) hipLaunchKernelGGL(bench<0>, dim3(1), dim3(64), lds, 0, dA, MP, R, dout, sink);
            if (var == 1) hipLaunchKernelGGL(bench<1>, dim3(1), dim3(64), lds, 0, dA, MP, R, dout, sink);
            CK(hipDeviceSynchronize());
        }
        long long h[2]; CK(hipMemcpy(h, dout, 16, hipMemcpyDeviceToHost));
        printf("variant %d: %.3f us per block column (bad %lld)\n", var, h[0] * 0.01 / R, h[1]);
    }
    return 0;
}

// dpp_probe.hip -- does v_fmac_f64 take a DPP row_newbcast operand on gfx950, what does it select, what does it cost?
// (round 4: the wave-per-front factorisation of nd_kernels.hpp gets its multipliers this way.)  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define FMAC_BC(acc, src, mul, N) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #N " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul))

__global__ void sel_kernel(double* out) {
    const int lane = threadIdx.x;
    double src = (double)lane, one = 1.0, a5 = 0.0, a11 = 100.0;
    FMAC_BC(a5, src, one, 5);
    FMAC_BC(a11, src, one, 11);
    out[lane] = a5;
    out[64 + lane] = a11;
}

template <bool DPP>
__global__ void rate_kernel(double* out, long long* cyc, int reps) {
    const int lane = threadIdx.x & 63;
    double acc[16], src = 1.0 + lane * 1e-9, mul = 1e-9 * (lane + 1);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = i;
    const long long t0 = clock64();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (DPP) FMAC_BC(acc[i], src, mul, 3);
            else asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(acc[i]) : "v"(src), "v"(mul));
        }
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    double* d; long long* c;
    hipMalloc(&d, sizeof(double) * 1 << 20); hipMalloc(&c, 8);
    sel_kernel<<<1, 64>>>(d);
    std::vector<double> h(128);
    hipMemcpy(h.data(), d, 128 * 8, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        if (h[l] != 16 * (l / 16) + 5) ++bad;
        if (h[64 + l] != 100 + 16 * (l / 16) + 11) ++bad;
    }
    printf("row_newbcast selects lane 16*(lane/16)+N: %s (lane 0: %g %g, lane 37: %g %g)\n", bad ? "NO" : "yes", h[0], h[64], h[37], h[64 + 37]);
    const int reps = 4096;
    for (int waves = 1; waves <= 4; waves *= 2)
        for (int dpp = 0; dpp < 2; ++dpp) {
            long long cy = 0;
            for (int it = 0; it < 2; ++it) {
                if (dpp) rate_kernel<true><<<1, 256 * waves>>>(d, c, reps);   // `waves` waves per SIMD of one CU
                else rate_kernel<false><<<1, 256 * waves>>>(d, c, reps);
                hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
            }
            printf("%s, %d waves per SIMD: %.2f shader cycles per instruction and wave\n", dpp ? "v_fmac_f64_dpp" : "v_fmac_f64    ", waves, (double)cy / (16.0 * reps));
        }
    return bad != 0;
}

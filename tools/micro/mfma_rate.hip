// dev micro-benchmark: issue rate of v_mfma_f32_32x32x16_f16 in the correlation kernel's pattern (8 accumulator tiles, 24 MFMAs
// per step) at 1 and 2 waves per SIMD.   hipcc -O3 --offload-arch=gfx950 mfma_rate.hip -o mfma_rate && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int WPS>
__global__ void __launch_bounds__(256, WPS) burn(const _Float16* in, float* out, int steps) {
    f16x8 a[2][2], b[4][2];
    const int lane = threadIdx.x;
    for (int h = 0; h < 2; ++h) for (int t = 0; t < 2; ++t) a[h][t] = *reinterpret_cast<const f16x8*>(in + ((h * 2 + t) * 256 + lane) * 8);
    for (int h = 0; h < 4; ++h) for (int t = 0; t < 2; ++t) b[h][t] = *reinterpret_cast<const f16x8*>(in + ((4 + h * 2 + t) * 256 + lane) * 8);
    f32x16 acc[2][4];
    for (int h = 0; h < 2; ++h) for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[h][t][r] = 0.f;
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[h][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[h][p == 0], b[t][p == 1], acc[h][t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0.f;
    for (int h = 0; h < 2; ++h) for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[h][t][r];
    out[blockIdx.x * 256 + lane] = s;
}

int main() {
    _Float16* in; float* out;
    hipMalloc(&in, 12 * 256 * 8 * 2); hipMemset(in, 0x3c, 12 * 256 * 8 * 2);
    hipMalloc(&out, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int steps = 20000;
    for (int wps = 1; wps <= 2; ++wps) {
        const int grid = 256 * wps;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (wps == 1) hipLaunchKernelGGL(burn<1>, dim3(grid), dim3(256), 0, 0, in, out, steps);
            else hipLaunchKernelGGL(burn<2>, dim3(grid), dim3(256), 0, 0, in, out, steps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double mfma_per_simd = (double)steps * 24 * wps;
            printf("waves/SIMD %d: %.3f ms, %.1f ns per MFMA per SIMD (= %.1f clk at 2.4 GHz), %.0f TFLOP/s\n", wps, ms,
                   ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4, mfma_per_simd * 1024 * 32768 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}

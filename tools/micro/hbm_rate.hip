// dev micro-benchmark: what this chip's memory system sustains for streams (read / write / copy, 16 bytes per lane) and for
// 64-byte random sectors - the calibration of "8 TB/s" that SURVEY 8(d) asks for.
//   hipcc -O3 --offload-arch=gfx950 hbm_rate.hip -o hbm_rate && ./hbm_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_read(const f32x4* __restrict__ in, float* __restrict__ out, size_t n) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) acc += in[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1234.5f) out[0] = 1.f;
}
__global__ void __launch_bounds__(256) k_write(f32x4* __restrict__ out, size_t n, float v) {
    const f32x4 x = {v, v, v, v};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) __builtin_nontemporal_store(x, out + i);
}
__global__ void __launch_bounds__(256) k_copy(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) out[i] = in[i];
}
// every lane reads one 16-byte piece of its own pseudo-random 64-byte sector
__global__ void __launch_bounds__(256) k_gather64(const f32x4* __restrict__ in, float* __restrict__ out, size_t sectors, int per_thread) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    uint64_t s = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    for (int k = 0; k < per_thread; ++k) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        acc += in[((s >> 20) % sectors) * 4];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1234.5f) out[0] = 1.f;
}

int main() {
    const size_t bytes = 2ull << 30, n = bytes / 16;
    f32x4 *a, *b; float* flag;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&flag, 4);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, double moved, auto launch) {
        launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        printf("%-28s %8.3f ms  %7.0f GB/s\n", name, ms, moved / (ms * 1e-3) / 1e9);
    };
    const int grid = 256 * 16;
    timeit("read  (2 GiB)", (double)bytes, [&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, flag, n); });
    timeit("write (2 GiB, non-temporal)", (double)bytes, [&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, b, n, 1.0f); });
    timeit("copy  (2 + 2 GiB)", 2.0 * bytes, [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); });
    const int per_thread = 16;
    const double pieces = (double)grid * 256 * per_thread;
    timeit("random 64-B sectors (2 GiB)", pieces * 64.0, [&] { hipLaunchKernelGGL(k_gather64, dim3(grid), dim3(256), 0, 0, a, flag, bytes / 64, per_thread); });
    return 0;
}

// dev micro-benchmark: calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 in the ACCESS SHAPES of the MSDA forward (the guide,
// MI355X_MICROARCH.md "HBM": FETCH_SIZE reports half of a wide coalesced streaming read; other widths and WRITE_SIZE are uncalibrated —
// "calibrate on a known byte count in your own access pattern").  Every kernel below moves a KNOWN number of bytes exactly once
// (buffers of 1 GiB: far beyond the 256 MiB Infinity Cache, so nothing is served on-die):
//   stream16   16 bytes per lane, coalesced (offsets, the LDS image copy)            1 GiB read
//   stream8     8 bytes per lane, coalesced (attention logits)                       1 GiB read
//   rows64     64-byte rows, 4 lanes x 16 B each, every row once in a scattered order (bf16 head rows)   1 GiB read
//   rows128   128-byte rows, 8 lanes x 16 B each, every row once in a scattered order (fp32 head rows)   1 GiB read
//   write16    16 bytes per lane, coalesced plain stores (the output)                1 GiB written
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/cal -- tools/micro/fetch_calib   (WRITE_SIZE in a pass of its own)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) stream16(const f32x4* __restrict__ in, float* __restrict__ out, size_t n) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) acc += in[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1234.5f) out[0] = 1.f;
}
__global__ void __launch_bounds__(256) stream8(const f32x2* __restrict__ in, float* __restrict__ out, size_t n) {
    f32x2 acc = {0.f, 0.f};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) acc += in[i];
    if (acc[0] + acc[1] == 1234.5f) out[0] = 1.f;
}
// LANES lanes x 16 B cover one row; row index = a bijection of the visit index (odd multiplier modulo a power of two)
template <int LANES>
__global__ void __launch_bounds__(256) rows(const f32x4* __restrict__ in, float* __restrict__ out, size_t nrows) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t t = blockIdx.x * 256ull + threadIdx.x;
    const size_t part = t % LANES;
    for (size_t v = t / LANES; v < nrows; v += (size_t)gridDim.x * 256ull / LANES) {
        const size_t r = (v * 0x9E3779B1ull + 12345ull) & (nrows - 1);
        acc += in[r * LANES + part];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1234.5f) out[0] = 1.f;
}
__global__ void __launch_bounds__(256) write16(f32x4* __restrict__ out, size_t n, float v) {
    const f32x4 x = {v, v, v, v};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) out[i] = x;
}

int main() {
    const size_t bytes = 1ull << 30;
    f32x4* a; float* flag;
    hipMalloc(&a, bytes); hipMalloc(&flag, 4);
    hipMemset(a, 0, bytes);
    const int grid = 256 * 16;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream16, dim3(grid), dim3(256), 0, 0, a, flag, bytes / 16);
        hipLaunchKernelGGL(stream8, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const f32x2*>(a), flag, bytes / 8);
        hipLaunchKernelGGL(rows<4>, dim3(grid), dim3(256), 0, 0, a, flag, bytes / 64);
        hipLaunchKernelGGL(rows<8>, dim3(grid), dim3(256), 0, 0, a, flag, bytes / 128);
        hipLaunchKernelGGL(write16, dim3(grid), dim3(256), 0, 0, a, bytes / 16, 1.0f);
    }
    hipDeviceSynchronize();
    printf("every kernel moved %zu bytes\n", bytes);
    return 0;
}

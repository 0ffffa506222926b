// dev micro-benchmark: what does it cost to flush a 128-byte fp32 row with MORE THAN ONE atomic instruction?  The wide MSDA backward
// flushes a row with one instruction of a half wave (32 lanes x 4 B); letting a quarter wave own a short row would need two
// instructions per row.  Rows as in atomic_width.hip (pseudo-random inside a 2048-row window per wave).
//   A  1 instruction / row: 2 rows x 32 lanes                          (the kernel today)
//   B  2 instructions / row: 4 rows x 16 lanes, each a contiguous 64-byte half row
//   C  2 instructions / row: 4 rows x 16 lanes, every other dword of the whole row
//   D  4 instructions / row: 8 rows x 8 lanes, every fourth dword (the first version of the wide kernel)
//   E  4 instructions / row: 8 rows x 8 lanes, each a contiguous 32-byte quarter row
//   hipcc --offload-arch=gfx950 -O2 -munsafe-fp-atomics -w -o tools/micro/atomic_split tools/micro/atomic_split.hip && tools/micro/atomic_split
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(64) k(float* buf, unsigned nrows, int rows_per_wave) {
    const unsigned lane = threadIdx.x;
    unsigned s = blockIdx.x * 2654435761u + 12345u;
    const unsigned win0 = (blockIdx.x * 97u) % (nrows - 4096u);
    constexpr int LANES = MODE == 0 ? 32 : (MODE >= 3 ? 8 : 16), ROWS = 64 / LANES, INSTR = 32 / LANES;
    const unsigned sub = lane / LANES, l = lane % LANES;
    for (int i = 0; i < rows_per_wave; i += ROWS) {
        s = s * 1664525u + 1013904223u;
        const unsigned r = win0 + ((s >> 8) % 2048u) + sub * 7u;
        float* row = buf + (size_t)r * 32;
#pragma unroll
        for (int j = 0; j < INSTR; ++j) {
            const unsigned ch = MODE == 0 ? l : (MODE == 1 ? 16u * j + l : (MODE == 2 ? 2u * l + j : (MODE == 3 ? 4u * l + j : 8u * j + l)));
            atomicAdd(row + ch, 1.0f);
        }
    }
}

int main() {
    const unsigned nrows = 4u * 22223u * 8u;
    float* buf;
    hipMalloc(&buf, (size_t)nrows * 128);
    hipMemset(buf, 0, (size_t)nrows * 128);
    const int waves = 45568, rows_per_wave = 144;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](auto kern, const char* name) {
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(waves), dim3(64), 0, 0, buf, nrows, rows_per_wave);
        hipEventRecord(a);
        for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(kern, dim3(waves), dim3(64), 0, 0, buf, nrows, rows_per_wave);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        ms /= 20;
        printf("%-72s %.4f ms  %.1f G rows/s\n", name, ms, waves * (double)rows_per_wave / ms / 1e6);
    };
    run(k<0>, "A  1 instruction / row  (2 rows x 32 lanes)");
    run(k<1>, "B  2 instructions / row (4 rows x 16 lanes, contiguous 64-byte halves)");
    run(k<2>, "C  2 instructions / row (4 rows x 16 lanes, every other dword)");
    run(k<3>, "D  4 instructions / row (8 rows x 8 lanes, every fourth dword)");
    run(k<4>, "E  4 instructions / row (8 rows x 8 lanes, contiguous 32-byte quarters)");
    return 0;
}

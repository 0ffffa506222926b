// dev micro-benchmark: does the width of a global atomic change what a 128-byte row costs?  tools/micro/atomic_scope.hip found 10.5 G
// rows/s for fp32 row atomics (32 dword adds per row) whatever the pattern: "one 4-byte add per clock per L2 channel".  If an 8-byte
// atomic costs the channel one operation as well, a row written as 16 x 64-bit adds costs half — which is what two channels packed
// as 32-bit fixed point into one u64 add would buy the MSDA backward (and make its sums order-independent).
//   hipcc --offload-arch=gfx950 -O2 -munsafe-fp-atomics -w -o tools/micro/atomic_width tools/micro/atomic_width.hip && tools/micro/atomic_width
#include <hip/hip_runtime.h>
#include <cstdio>

// same row pattern as atomic_scope's k<>: pseudo-random rows inside a 2048-row window per wave
template <int MODE>
__global__ void __launch_bounds__(64) k(void* buf, unsigned nrows, int rows_per_wave) {
    const unsigned lane = threadIdx.x;
    unsigned s = blockIdx.x * 2654435761u + 12345u;
    const unsigned win0 = (blockIdx.x * 97u) % (nrows - 4096u);
    if (MODE == 0) {            // 2 rows per instruction, 32 lanes x 4 B each
        const unsigned half = lane >> 5, ch = lane & 31;
        for (int i = 0; i < rows_per_wave; i += 2) {
            s = s * 1664525u + 1013904223u;
            const unsigned r = win0 + ((s >> 8) % 2048u) + half * 7u;
            atomicAdd(reinterpret_cast<float*>(buf) + (size_t)r * 32 + ch, 1.0f);
        }
    } else {                    // 4 rows per instruction, 16 lanes x 8 B each
        const unsigned quarter = lane >> 4, ch = lane & 15;
        for (int i = 0; i < rows_per_wave; i += 4) {
            s = s * 1664525u + 1013904223u;
            const unsigned r = win0 + ((s >> 8) % 2048u) + quarter * 7u;
            if (MODE == 1) atomicAdd(reinterpret_cast<unsigned long long*>(buf) + (size_t)r * 16 + ch, 0x0000000100000001ull);
            else atomicAdd(reinterpret_cast<double*>(buf) + (size_t)r * 16 + ch, 1.0);
        }
    }
}

int main() {
    const unsigned nrows = 4u * 22223u * 8u;
    void* buf;
    hipMalloc(&buf, (size_t)nrows * 128);
    hipMemset(buf, 0, (size_t)nrows * 128);
    const int waves = 45568, rows_per_wave = 140;
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
        printf("%-46s %.4f ms  %.1f G rows/s\n", name, ms, waves * (double)rows_per_wave / ms / 1e6);
    };
    run(k<0>, "fp32 adds, 32 x 4 B per row");
    run(k<1>, "u64 adds, 16 x 8 B per row");
    run(k<2>, "f64 adds, 16 x 8 B per row");
    return 0;
}

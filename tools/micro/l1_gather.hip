// Micro-benchmark: how fast does one CU's vector L1 serve gathers of 64-byte pieces (4 lanes x 16 B) compared with 128-byte
// pieces (8 lanes x 16 B) when every piece is a random, naturally aligned chunk of an L2-resident array?
// Build: hipcc -O3 --offload-arch=gfx950 l1_gather.hip -o l1_gather ; run: ./l1_gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int GROUP>   // lanes per contiguous piece: 4 -> 64 B, 8 -> 128 B, 2 -> 32 B
__global__ void __launch_bounds__(256) gather(const u32x4* __restrict__ data, const unsigned* __restrict__ idx, unsigned* out, int iters,
                                              unsigned mask) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    unsigned piece = idx[(tid / GROUP) & 0xffff];
    u32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned p = (piece * 2654435761u + j * 40503u + i * 7919u) & mask;   // pseudo-random piece index
            v[j] = data[(size_t)p * GROUP + (lane % GROUP)];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc += v[j]; }
        piece += acc.x & 1;
    }
    if (acc.x == 0x12345678u) out[tid] = acc.y;
}

template <int GROUP>
void run(const u32x4* d, const unsigned* idx, unsigned* out, size_t bytes) {
    const unsigned pieces = (unsigned)(bytes / (16 * GROUP));
    unsigned mask = 1; while (mask * 2 <= pieces) mask *= 2; mask -= 1;
    const int iters = 200, blocks = 256 * 8;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    gather<GROUP><<<blocks, 256>>>(d, idx, out, 10, mask);
    hipDeviceSynchronize();
    hipEventRecord(a);
    gather<GROUP><<<blocks, 256>>>(d, idx, out, iters, mask);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double lane_loads = (double)blocks * 256 * iters * 8;
    const double gb = lane_loads * 16 / 1e9;
    printf("piece %3d B over %6.1f MB: %.3f ms  %.1f TB/s  %.2f G pieces/s\n", 16 * GROUP, bytes / 1e6, ms, gb / ms, lane_loads / GROUP / ms / 1e6);
}

int main() {
    for (size_t bytes : {(size_t)2 << 20, (size_t)24 << 20, (size_t)96 << 20}) {
        u32x4* d; unsigned* idx; unsigned* out;
        hipMalloc(&d, bytes); hipMemset(d, 1, bytes);
        hipMalloc(&idx, 65536 * 4); hipMalloc(&out, 256 * 8 * 256 * 4);
        std::vector<unsigned> h(65536); for (auto& x : h) x = rand();
        hipMemcpy(idx, h.data(), 65536 * 4, hipMemcpyHostToDevice);
        run<2>(d, idx, out, bytes); run<4>(d, idx, out, bytes); run<8>(d, idx, out, bytes); run<16>(d, idx, out, bytes);
        hipFree(d); hipFree(idx); hipFree(out);
    }
    return 0;
}

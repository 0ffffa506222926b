// dev micro-benchmark: what an LDS accumulation costs on gfx950 — the primitive an output-stationary MSDA backward would stand on
// (a workgroup owns a tile of grad_value in LDS and adds 128-byte rows to it).  Per wave-instruction, two 32-channel rows:
//   ds_add_f32 (no return)          the LDS's own float add
//   ds_read_b32 + v_add + ds_write  read-modify-write by a wave that owns its rows (no atomicity needed)
//   ds_add_rtn_u32                  slot allocation for a bucketed (sorted) formulation
// Rows are pseudo-random inside a 512-row tile (conflict-free inside an instruction: 64 lanes = 2 rows x 32 consecutive floats).
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/lds_atomic tools/micro/lds_atomic.hip && tools/micro/lds_atomic
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int kRows = 512;

template <int MODE, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k(float* out, int iters) {
    __shared__ float tile[kRows * 32];
    __shared__ unsigned counters[kRows];
    for (int i = threadIdx.x; i < kRows * 32; i += blockDim.x) tile[i] = 0.f;
    for (int i = threadIdx.x; i < kRows; i += blockDim.x) counters[i] = 0;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, half = lane >> 5, ch = lane & 31, wave = threadIdx.x >> 6;
    unsigned s = (blockIdx.x * WAVES + wave) * 2654435761u + 12345u;
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        s = s * 1664525u + 1013904223u;
        // MODE 1: every wave owns its own quarter of the rows (plain read-modify-write is race-free)
        unsigned r = MODE == 1 ? (wave * (kRows / WAVES) + ((s >> 9) % (kRows / WAVES / 2)) * 2 + half) : (((s >> 9) % (kRows / 2)) * 2 + half);
        float v = __uint_as_float(0x3f800000u | (s & 0x7fffff)) - 1.0f;
        if (MODE == 0) {
            __hip_atomic_fetch_add(&tile[r * 32 + ch], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_add_f32
        } else if (MODE == 1) {
            tile[r * 32 + ch] += v;                                                                         // read, add, write
        } else {
            acc += (float)__hip_atomic_fetch_add(&counters[(r * 7 + lane) % kRows], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    float sum = acc;
    for (int i = threadIdx.x; i < kRows * 32; i += blockDim.x) sum += tile[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int MODE, int WAVES>
void run(const char* name, float* out) {
    const int blocks = 256 * 2, iters = 4096;   // two workgroups per CU
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, out, iters);
    hipEventRecord(a);
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((k<MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    ms /= 10;
    const double instr = (double)blocks * WAVES * iters;
    // cycles per wave-instruction per CU at 2.4 GHz, 256 CUs
    printf("%-52s %8.4f ms  %7.2f G wave-instr/s  = %5.2f clk per wave-instr per CU  (%.1f G rows/s)\n", name, ms, instr / ms / 1e6,
           ms * 1e-3 * 2.4e9 * 256 / instr, instr * 2 / ms / 1e6);
}

int main() {
    float* out;
    hipMalloc(&out, 512 * 1024 * 4);
    run<0, 4>("ds_add_f32, 4 waves/WG, 2 WG/CU", out);
    run<0, 8>("ds_add_f32, 8 waves/WG, 2 WG/CU", out);
    run<1, 4>("read + add + write (wave-owned rows), 4 waves", out);
    run<1, 8>("read + add + write (wave-owned rows), 8 waves", out);
    run<2, 4>("ds_add_rtn_u32 (scattered counters), 4 waves", out);
    run<2, 8>("ds_add_rtn_u32 (scattered counters), 8 waves", out);
    return 0;
}

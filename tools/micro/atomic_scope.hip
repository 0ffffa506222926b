// How fast are fp32 row atomics (128-byte rows, 32 lanes x 4 B) on MI355X when they execute in the XCD's own L2 (workgroup scope: no sc1)
// compared with device scope (sc1: performed at the memory side, the only place that is coherent across the 8 XCD L2s)?
//   hipcc --offload-arch=gfx950 -O2 -munsafe-fp-atomics -o tools/micro/atomic_scope tools/micro/atomic_scope.hip && tools/micro/atomic_scope
// Pattern: like msda_bwd_tiled_kernel's stage 2 — every wave adds `rows_per_wave` rows (two rows per instruction) at pseudo-random row
// indices inside a window that moves with the wave id; buffer = 4 x 22223 x 8 rows of 128 B (91 MB).  Variant "xcd": the rows a wave touches
// are owned by the XCD it runs on (row % 8 == XCC_ID), so L2-local atomics would be CORRECT there.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// window-like pattern: a wave adds to `rows_per_wave` CONSECUTIVE pixel rows of one head (row stride = 8 heads x 128 B, as in grad_value),
// neighbouring waves' windows overlap by half
__global__ void __launch_bounds__(64) kwin(float* buf, unsigned nrows, int rows_per_wave, int stride_rows) {
    const unsigned lane = threadIdx.x, half = lane >> 5, ch = lane & 31;
    const unsigned head = blockIdx.x & 7, tile = blockIdx.x >> 3;
    const unsigned px0 = (tile * (unsigned)(rows_per_wave / 2)) % (nrows / 8 - 1024u);
    for (int i = 0; i < rows_per_wave; i += 2) {
        const unsigned r = (px0 + (unsigned)(i + half) * stride_rows) * 8u + head;
        atomicAdd(buf + (size_t)r * 32 + ch, 1.0f);
    }
}

template <int SCOPE, bool XCD_OWNED>
__global__ void __launch_bounds__(64) k(float* buf, unsigned nrows, int rows_per_wave) {
    const unsigned lane = threadIdx.x, half = lane >> 5, ch = lane & 31;
    unsigned xcc = 0;
    if (XCD_OWNED) xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 7;   // HW_REG_XCC_ID (id 20), bits 3:0
    unsigned s = blockIdx.x * 2654435761u + 12345u;
    const unsigned win0 = (blockIdx.x * 97u) % (nrows - 4096u);
    for (int i = 0; i < rows_per_wave; i += 2) {
        s = s * 1664525u + 1013904223u;
        unsigned r = win0 + ((s >> 8) % 2048u) + half * 7u;
        if (XCD_OWNED) r = (r & ~7u) | xcc;
        float* p = buf + (size_t)r * 32 + ch;
        if (SCOPE == 0) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

int main() {
    const unsigned nrows = 4u * 22223u * 8u;
    float* buf;
    hipMalloc(&buf, (size_t)nrows * 128);
    hipMemset(buf, 0, (size_t)nrows * 128);
    const int waves = 45568, rows_per_wave = 140;   // 6.4 M rows
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
        printf("%-44s %.4f ms  %.1f G rows/s  %.2f TB/s of payload\n", name, ms, waves * (double)rows_per_wave / ms / 1e6,
               waves * (double)rows_per_wave * 128 / ms / 1e9);
    };
    auto runw = [&](int stride, const char* name) {
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kwin, dim3(waves), dim3(64), 0, 0, buf, nrows, rows_per_wave, stride);
        hipEventRecord(a);
        for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(kwin, dim3(waves), dim3(64), 0, 0, buf, nrows, rows_per_wave, stride);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        ms /= 20;
        printf("%-44s %.4f ms  %.1f G rows/s  %.2f TB/s of payload\n", name, ms, waves * (double)rows_per_wave / ms / 1e6,
               waves * (double)rows_per_wave * 128 / ms / 1e9);
    };
    runw(1, "window: consecutive pixels of one head");
    runw(3, "window: every 3rd pixel of one head");
    run(k<0, false>, "agent scope (sc1), any rows");
    run(k<1, false>, "workgroup scope (L2), any rows [incoherent]");
    run(k<0, true>, "agent scope (sc1), rows owned by the XCD");
    run(k<1, true>, "workgroup scope (L2), rows owned by the XCD");
    // correctness of the owned variant: every touched word must hold an integer count and the total must match
    hipMemset(buf, 0, (size_t)nrows * 128);
    hipLaunchKernelGGL((k<1, true>), dim3(waves), dim3(64), 0, 0, buf, nrows, rows_per_wave);
    hipDeviceSynchronize();
    float* h = (float*)malloc((size_t)nrows * 128);
    hipMemcpy(h, buf, (size_t)nrows * 128, hipMemcpyDeviceToHost);
    double total = 0;
    for (size_t i = 0; i < (size_t)nrows * 32; ++i) total += h[i];
    printf("owned + L2 atomics: sum %.0f, expected %.0f -> %s\n", total, (double)waves * rows_per_wave * 32,
           total == (double)waves * rows_per_wave * 32 ? "EXACT" : "LOST UPDATES");
    hipMemset(buf, 0, (size_t)nrows * 128);
    hipLaunchKernelGGL((k<1, false>), dim3(waves), dim3(64), 0, 0, buf, nrows, rows_per_wave);
    hipDeviceSynchronize();
    hipMemcpy(h, buf, (size_t)nrows * 128, hipMemcpyDeviceToHost);
    total = 0;
    for (size_t i = 0; i < (size_t)nrows * 32; ++i) total += h[i];
    printf("any rows + L2 atomics: sum %.0f, expected %.0f -> %s\n", total, (double)waves * rows_per_wave * 32,
           total == (double)waves * rows_per_wave * 32 ? "EXACT" : "LOST UPDATES (as expected: 8 incoherent L2s)");
    return 0;
}

// Which LDS element lands where with ds_read_b64_tr_b16?  Every lane supplies its own 8-byte address; LDS holds, at 16-bit element e of
// lane i's 8 bytes, the tag (i << 2) | e.  Prints, for every output lane and element, the (source lane, source element) it received.
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/tr_read tools/micro/tr_read.hip && tools/micro/tr_read
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4i16 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 4 * 3];
    const int l = threadIdx.x;
    // lane l's 8 bytes live at a scattered place: stride 24 bytes (= 12 elements), so addresses are neither contiguous nor ordered
    const int base = ((l * 37) % 64) * 12;
    for (int e = 0; e < 4; ++e) lds[base + e] = (unsigned short)((l << 2) | e);
    __syncthreads();
    v4i16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(lds + base));
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = (unsigned short)r[e];
}
int main() {
    unsigned short* d; unsigned short h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) {
            const int sl = h[l * 4 + e] >> 2, se = h[l * 4 + e] & 3;
            printf("  (%2d,%d)", sl, se);
            const int g = l & ~15, c = l & 15;
            if (sl != g + 4 * e + c / 4 || se != c % 4) ok = 0;
        }
        printf("\n");
    }
    printf("matches out[c][e] = in[4 e + c / 4][c %% 4] within each 16-lane group: %s\n", ok ? "YES" : "NO");
    return 0;
}

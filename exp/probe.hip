#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ uint16_t bf(float f) { return (uint16_t)(__float_as_uint(f) >> 16); }
__global__ void k(float* out, unsigned* permout) {
    int lane = threadIdx.x;
    int b = lane / 4, i = lane % 4;
    // A_b[i][k] = (i+1) * 2^k ; B_b[k][j] = 16^k... use small ints exactly representable in bf16
    s16x4 a, bb;
    for (int kk = 0; kk < 4; ++kk) {
        a[kk] = (short)bf((float)((i + 1) * (1 << kk)));          // row i, col k
        bb[kk] = (short)bf((float)((kk + 1) + 8 * i) + 0.0f * b);   // B[k][j=i] = (k+1) + 8*j
    }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, bb, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
    unsigned x = 0x33221100u, y = 0x77665544u;
    permout[0] = __builtin_amdgcn_perm(y, x, 0x05040100u);
    permout[1] = __builtin_amdgcn_perm(y, x, 0x07060302u);
}
int main() {
    float* d; unsigned* p; hipMalloc(&d, 64 * 4 * 4); hipMalloc(&p, 8);
    k<<<1, 64>>>(d, p);
    float h[256]; unsigned hp[2];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hp, p, 8, hipMemcpyDeviceToHost);
    for (int lane = 0; lane < 8; ++lane) printf("lane %d: %g %g %g %g\n", lane, h[lane*4], h[lane*4+1], h[lane*4+2], h[lane*4+3]);
    // expected if D[i][j] (vgpr i, lane j) = sum_k A[i][k] B[k][j]: = sum_k (i+1) 2^k ((k+1)+8j)
    for (int j = 0; j < 4; ++j) { printf("expect lane %d:", j); for (int i = 0; i < 4; ++i) { float s = 0; for (int kk = 0; kk < 4; ++kk) s += (i+1)*(1<<kk)*((kk+1)+8*j); printf(" %g", s);} printf("\n"); }
    printf("perm lo %08x hi %08x\n", hp[0], hp[1]);
    return 0;
}

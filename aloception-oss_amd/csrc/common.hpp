// Shared helpers for the gfx950 kernels: error reporting, buffer-resource loads, element conversion.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/alo_hotpath.h"

namespace alo {

// ---- host side: thread-local error string -------------------------------------------------------------------------
char* error_buffer();  // api.hip
inline int fail(alo_status_t code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return (int)code;
}
#define ALO_REQUIRE(cond, code, ...) \
    do {                             \
        if (!(cond)) return ::alo::fail(code, __VA_ARGS__); \
    } while (0)

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ALO_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return ALO_OK;
}

// Raise a kernel's dynamic-LDS limit once per (kernel, device): a function attribute is per device, so a process-wide flag
// would leave the second GPU of a single-process multi-GPU job at the 48/64 KB default.  `done` is one atomic bit mask per
// call site (up to 64 devices); the attribute call itself is idempotent, so a race only repeats it.
inline hipError_t ensure_dynamic_lds(const void* kernel, int bytes, unsigned long long* done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (__atomic_load_n(done, __ATOMIC_ACQUIRE) & bit) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) __atomic_fetch_or(done, bit, __ATOMIC_RELEASE);
    return e;
}

constexpr int kNumXcd = 8;  // MI355X: 8 XCDs, block b is dispatched to XCD b % 8 (speed only, never correctness)

// Remap a launch-order block id so that each XCD (private 4 MiB L2) receives one CONTIGUOUS range of logical work
// items instead of every 8th one.  Bijective for any grid size.
__device__ __forceinline__ unsigned xcd_contiguous_block(unsigned bid, unsigned nblocks) {
    const unsigned q = nblocks / kNumXcd, r = nblocks % kNumXcd;
    const unsigned xcd = bid % kNumXcd, k = bid / kNumXcd;
    const unsigned start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + k;
}

// msda_bwd_wide.hip: the 16x16-block backward (fp32 / bf16, D = 32 or 64, L = P = 4, Lq == S).  ALO_ERR_UNSUPPORTED = nothing enqueued.
int msda_backward_wide(const void* value, const int32_t* shapes, const int32_t* lstart, const void* loc, const void* attn,
                       const void* grad_out, void* grad_value, void* grad_loc, void* grad_attn, int N, int S, int M, int D, int Lq,
                       int value_dtype, const int32_t* host_shapes, hipStream_t stream, bool plan_only = false);

// ---- device side ---------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct bf16_t {
    uint16_t bits;
};

__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float(((unsigned)b) << 16); }
// fp32 -> bf16, round to nearest even: gfx950 converts in hardware (v_cvt_pk_bf16_f32, two values per instruction)
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    const __bf16 b = static_cast<__bf16>(f);
    return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {  // lo in bits 0-15
    typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
    typedef float f32x2_hw __attribute__((ext_vector_type(2)));
    const bf16x2_hw r = __builtin_convertvector(f32x2_hw{lo, hi}, bf16x2_hw);
    return __builtin_bit_cast(unsigned, r);
}

// A raw (stride-0) buffer resource over [base, base + bytes).  Loads whose byte offset falls outside return 0 and
// touch no memory: the hardware's bounds check is how out-of-map bilinear corners get their zero padding.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
constexpr unsigned kOutOfRange = 0xC0000000u;  // byte offset guaranteed past any slab (slabs are < 3 GiB)

// Load VEC elements of storage type T at byte offset `off` (raw registers), widen them to the compute type CT later:
// keeping the two steps apart lets a kernel put a whole batch of loads in flight before the first conversion.
template <typename T, typename CT, int VEC>
struct Loader;

template <>
struct Loader<float, float, 4> {
    using raw_t = u32x4;
    static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, unsigned off) {
        return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    }
    static __device__ __forceinline__ void widen(const raw_t& x, float (&v)[4]) {
        v[0] = __uint_as_float(x.x); v[1] = __uint_as_float(x.y); v[2] = __uint_as_float(x.z); v[3] = __uint_as_float(x.w);
    }
};
template <>
struct Loader<float, float, 1> {
    using raw_t = unsigned;
    static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, unsigned off) {
        return __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0);
    }
    static __device__ __forceinline__ void widen(const raw_t& x, float (&v)[1]) { v[0] = __uint_as_float(x); }
};
template <>
struct Loader<double, double, 2> {
    using raw_t = u32x4;
    static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, unsigned off) {
        return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    }
    static __device__ __forceinline__ void widen(const raw_t& x, double (&v)[2]) {
        v[0] = __hiloint2double((int)x.y, (int)x.x);
        v[1] = __hiloint2double((int)x.w, (int)x.z);
    }
};
template <>
struct Loader<double, double, 1> {
    using raw_t = u32x2;
    static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, unsigned off) {
        return __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
    }
    static __device__ __forceinline__ void widen(const raw_t& x, double (&v)[1]) {
        v[0] = __hiloint2double((int)x.y, (int)x.x);
    }
};
template <>
struct Loader<bf16_t, float, 8> {
    using raw_t = u32x4;
    static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, unsigned off) {
        return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    }
    static __device__ __forceinline__ void widen(const raw_t& x, float (&v)[8]) {
        const unsigned w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
};
template <>
struct Loader<bf16_t, float, 1> {
    using raw_t = unsigned short;
    static __device__ __forceinline__ raw_t load(__amdgpu_buffer_rsrc_t r, unsigned off) {
        return __builtin_amdgcn_raw_buffer_load_b16(r, off, 0, 0);
    }
    static __device__ __forceinline__ void widen(const raw_t& x, float (&v)[1]) { v[0] = bf16_to_f32(x); }
};

// Scalar global load/store with widening / narrowing (sampling locations, attention weights, outputs).
__device__ __forceinline__ float ld(const float* p) { return *p; }
__device__ __forceinline__ double ld(const double* p) { return *p; }
__device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(p->bits); }
__device__ __forceinline__ void st(float* p, float v) { *p = v; }
__device__ __forceinline__ void st(double* p, double v) { *p = v; }
__device__ __forceinline__ void st(bf16_t* p, float v) { p->bits = f32_to_bf16(v); }

// Store VEC consecutive outputs (16-byte vector store when VEC * sizeof(T) == 16).
template <typename T, typename CT, int VEC>
__device__ __forceinline__ void store_vec(T* p, const CT (&v)[VEC]) {
    if constexpr (sizeof(T) == 4 && VEC == 4) {
        *reinterpret_cast<f32x4*>(p) = f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    } else if constexpr (sizeof(T) == 8 && VEC == 2) {
        *reinterpret_cast<double2*>(p) = double2{(double)v[0], (double)v[1]};
    } else if constexpr (sizeof(T) == 2 && VEC == 8) {
        u32x4 o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        o.z = pack_bf16x2(v[4], v[5]);
        o.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<u32x4*>(p) = o;
    } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) st(p + i, v[i]);
    }
}

}  // namespace alo

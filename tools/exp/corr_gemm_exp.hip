// EXPERIMENT (round 4): where does corr_gemm3_kernel<POOLED> spend its 1.8 ms, and what does a software-pipelined main loop /
// a leaner epilogue buy?  Stand-alone copy of the library kernel's structure (csrc/corr.hip) with compile-time variants:
//
//   LOOP 0  the library's loop: DMA of slice k+1 at the top, operand reads just in front of their MFMAs, __syncthreads per slice
//   LOOP 1  operands of slice k+1 read into a second register set while the 24 MFMAs of slice k run; 2 LDS stages
//   LOOP 2  the same with 3 LDS stages: the DMA of a slice has two iterations to land
//   LOOP 3  LOOP 0 with the A operand loaded from memory straight into the MFMA operand registers (a wave's 64 rows of A are
//           its own: only B is shared by the four waves and goes through LDS), one slice ahead
//   LOOP 9  no loop at all (accumulators = lane id): the epilogue alone
//   EPI 0   the library's epilogue (per-element predicates, 64-bit addresses, non-temporal dword stores)
//   EPI 1   no epilogue (one conditional store that keeps the accumulators alive): the loop alone
//   EPI 2   buffer stores: the row tile is a buffer resource whose range check drops rows past the edge, scalar row offsets
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o corr_gemm_exp corr_gemm_exp.hip && ./corr_gemm_exp
// One JSON line per variant (ms over 20 launches after 5 warm-ups; max |difference| of the three levels against LOOP 0 / EPI 0).
// Not part of the library; nothing imports this.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef _Float16 __attribute__((ext_vector_type(8))) f16x8_t;

#define CK(x)                                                                                          \
    do {                                                                                               \
        hipError_t e_ = (x);                                                                           \
        if (e_ != hipSuccess) {                                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                  \
            exit(1);                                                                                   \
        }                                                                                              \
    } while (0)

constexpr unsigned kNumXcd = 8;
__device__ __forceinline__ unsigned xcd_contiguous_block(unsigned bid, unsigned nblocks) {
    const unsigned q = nblocks / kNumXcd, r = nblocks % kNumXcd;
    const unsigned xcd = bid % kNumXcd, k = bid / kNumXcd;
    const unsigned start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + k;
}
__device__ __forceinline__ f16x8_t as_f16x8(const u32x4& v) {
    union { u32x4 u; f16x8_t h; } x;
    x.u = v;
    return x.h;
}

struct Args {
    const uint16_t* a;
    const uint16_t* b;
    float* out0;
    float* out1;
    float* out2;
    int B, KC, HW;
    int H, W, n;
    int h1, w1, h2, w2;
    int tiles_m, tiles_r, tiles_c;
    float scale;
    unsigned nblocks;
    unsigned long long* trace;   // per workgroup: {hw id, start, loop end, end} (s_memrealtime, 100 MHz) or null
    unsigned* tickets;           // per CU arrival counter (zeroed before the launch) or null
    int stagger;   // first-generation workgroups 256..511 (the second slot of every CU) sleep this many x 8128 cycles before they start
};

constexpr int kTM = 256, kTN = 128, kThreads = 256, kRowGroup = 4;
constexpr int kGranA = 2 * kTM * 2, kGranB = 2 * kTN * 2;
constexpr int kStageGran = kGranA + kGranB;   // 1536 granules = 24 KiB

template <int ROWS>
__device__ __forceinline__ int slot(int term, int row, int g) { return (term * ROWS + row) * 2 + (g ^ ((row >> 3) & 1)); }

__device__ __forceinline__ void dma16(const uint16_t* src, u32x4* lds_dst) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(src, lds_dst, 16, 0, 0);
#endif
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}

template <int LOOP, int EPI, int PRIO = 0, int AUX = 2>
__global__ void __launch_bounds__(kThreads, 2) gemm3(const Args g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* const lds0 = reinterpret_cast<u32x4*>(smem_raw);
    constexpr int STAGE = LOOP == 3 ? kGranB : kStageGran;
    auto ldsA = [&](int stage) { return lds0 + stage * STAGE; };
    auto ldsB = [&](int stage) { return lds0 + stage * STAGE + (LOOP == 3 ? 0 : kGranA); };

    const unsigned lb = xcd_contiguous_block(blockIdx.x, g.nblocks);
    const int tiles_n = g.tiles_r * g.tiles_c, row_groups = (g.tiles_m + kRowGroup - 1) / kRowGroup;
    const int tm = ((lb / (kRowGroup * tiles_n)) % row_groups) * kRowGroup + lb % kRowGroup;
    const int tn = (lb / kRowGroup) % tiles_n;
    const int b = lb / (kRowGroup * tiles_n * row_groups);
    if (tm >= g.tiles_m) return;
    unsigned long long t_start = 0, t_loop = 0, r_start = 0;
    unsigned hwid = 0;
    if (g.trace || g.tickets) {
        const unsigned id = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID, all 32 bits
        const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID[3:0]
        hwid = (xcc << 16) | ((id >> 8) & 0xff) | (((id >> 13) & 7) << 8);                  // xcc | se | sh, cu
        t_start = __builtin_readcyclecounter();
        r_start = __builtin_amdgcn_s_memrealtime();
    }
    if (g.stagger > 0 && g.tickets && blockIdx.x < 512) {
        // the SECOND workgroup to arrive on a CU waits: the two then alternate between loop and epilogue instead of doing both together
        const unsigned cu = ((hwid >> 16) & 7) * 256 + (hwid & 0xff) + ((hwid >> 8) & 7) * 32;
        unsigned tk = 0;
        if (threadIdx.x == 0) tk = atomicAdd(&g.tickets[cu & 2047], 1u);
        tk = __builtin_amdgcn_readfirstlane(tk);
        if (threadIdx.x < 64 && (tk & 1))
            for (int i = 0; i < g.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    } else if (g.stagger > 0 && !g.tickets && blockIdx.x >= 256 && blockIdx.x < 512) {
        for (int i = 0; i < g.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }
    const int tr = tn / g.tiles_c, tc = tn % g.tiles_c;
    const int i0 = tm * kTM;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const int drow = lane >> 1, dhalf = lane & 1;
    const uint16_t* asrc[4];
    const uint16_t* bsrc[2];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int ins = wave + 4 * e, term = ins >> 3, row = 32 * (ins & 7) + drow;
        const int half = dhalf ^ ((row >> 3) & 1);
        asrc[e] = g.a + ((((long)b * g.KC) * 2 + term) * g.HW + (i0 + row < g.HW ? i0 + row : 0)) * 16 + half * 8;
    }
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int ins = wave + 4 * e, term = ins >> 2, row = 32 * (ins & 3) + drow;
        const int half = dhalf ^ ((row >> 3) & 1);
        const int y = 4 * tr + (row >> 5), x = 32 * tc + (row & 31);
        const bool ok = y < g.H && x < g.W;
        const long col = (long)y * g.W + x;
        bsrc[e] = g.b + ((((long)b * g.KC) * 2 + term) * g.n + (ok ? col : 0)) * 16 + half * 8;
    }
    const long a_step = 2L * g.HW * 16, b_step = 2L * g.n * 16;
    auto stage_in = [&](int kc, int stage) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dma16(asrc[e] + kc * a_step, ldsA(stage) + (wave + 4 * e) * 64);
#pragma unroll
        for (int e = 0; e < 2; ++e) dma16(bsrc[e] + kc * b_step, ldsB(stage) + (wave + 4 * e) * 64);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[h][t][r] = LOOP == 9 ? (float)lane : 0.f;

    const int kg = lane >> 5, li = lane & 31;

    struct Ops {
        u32x4 af[2][2], bf[4][2];
    };
    auto read_ops = [&](int stage, Ops& o) {
        const u32x4* As = ldsA(stage);
        const u32x4* Bs = ldsB(stage);
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int term = 0; term < 2; ++term) o.af[h][term] = As[slot<kTM>(term, 64 * wave + 32 * h + li, kg)];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int term = 0; term < 2; ++term) o.bf[t][term] = Bs[slot<kTN>(term, 32 * t + li, kg)];
    };
    auto multiply = [&](const Ops& o) {
#define STEP(AT, BT)                                                                                                             \
        _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                                            \
        _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                                            \
            acc[h][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(o.af[h][AT]), as_f16x8(o.bf[t][BT]), acc[h][t], 0, 0, 0);
        STEP(1, 0) STEP(0, 1) STEP(0, 0)
#undef STEP
    };

    if (LOOP == 0) {
        stage_in(0, 0);
        __syncthreads();
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
        for (int kc = 0; kc < g.KC; ++kc) {
            const int cur = kc & 1;
            if (kc + 1 < g.KC) stage_in(kc + 1, cur ^ 1);
            Ops o;
            read_ops(cur, o);
            multiply(o);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
        }
    } else if (LOOP == 1 || LOOP == 2) {
        constexpr int NS = LOOP == 1 ? 2 : 3;     // LDS stages
        constexpr int AHEAD = NS - 1;             // the DMA of slice k + AHEAD is issued in iteration k - 1 ... see below
        // pipeline: DMA(s) issued in iteration s - NS, operands of slice s read in iteration s - 1, multiplied in iteration s
        // prologue = "iterations" -NS .. -1
#pragma unroll
        for (int s = 0; s < NS; ++s)
            if (s < g.KC) stage_in(s, s);
        // slice 0 landed?  (NS - 1 younger slices may stay in flight: 6 DMA instructions per wave and slice)
        if (NS == 2) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)\n\ts_barrier" ::: "memory");
        Ops o[2];
        read_ops(0, o[0]);
        // slice 1 landed for everybody before iteration 0 reads it
        if (NS == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        int st_read = 1 % NS, st_dma = 0;   // stage of slice k + 1; stage of slice k (free once its reads are done), gets slice k + NS
        auto advance = [&]() {
            st_dma = st_dma + 1 == NS ? 0 : st_dma + 1;
            st_read = st_read + 1 == NS ? 0 : st_read + 1;
        };
        int k = 0;
        // steady state, two iterations per trip (the register sets swap roles), no conditions inside: a branch around the reads makes
        // the compiler wait for them at the join, in front of the MFMAs they were meant to hide under
        for (; k + NS + 1 < g.KC; k += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                stage_in(k + u + NS, st_dma);      // the stage of slice k + u is free: everybody read it before the last barrier
                read_ops(st_read, o[u ^ 1]);       // slice k + u + 1 -> the other register set
                multiply(o[u]);
                __builtin_amdgcn_sched_barrier(0);
                // before the next iteration: slice k + u + 2 landed (it is read there), this wave's reads of slice k + u + 1 done
                if (NS == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                advance();
            }
        }
        for (; k < g.KC; k += 2) {   // the last slices: nothing (or not everything) left to request
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int kk = k + u;
                if (kk < g.KC) {   // wave-uniform
                    if (kk + NS < g.KC) stage_in(kk + NS, st_dma);
                    if (kk + 1 < g.KC) read_ops(st_read, o[u ^ 1]);
                    multiply(o[u]);
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    advance();
                }
            }
        }
        (void)AHEAD;
    }

    if (LOOP == 3) {
        const u32x4* aptr[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int term = 0; term < 2; ++term) {
                const int row = i0 + 64 * wave + 32 * h + li;
                aptr[h][term] = reinterpret_cast<const u32x4*>(g.a + ((((long)b * g.KC) * 2 + term) * g.HW + (row < g.HW ? row : 0)) * 16 + kg * 8);
            }
        const long a_step16 = a_step / 8;   // in 16-byte units
        auto stage_b = [&](int kc, int stage) {
#pragma unroll
            for (int e = 0; e < 2; ++e) dma16(bsrc[e] + kc * b_step, ldsB(stage) + (wave + 4 * e) * 64);
        };
        u32x4 aset[2][2][2];
        auto load_a = [&](int kc, u32x4 (&dst)[2][2]) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int term = 0; term < 2; ++term) dst[h][term] = aptr[h][term][kc * a_step16];
        };
        stage_b(0, 0);
        load_a(0, aset[0]);
        __syncthreads();
        for (int kc = 0; kc < g.KC; kc += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int k = kc + u;
                if (k < g.KC) {
                    const int cur = k & 1;
                    if (k + 1 < g.KC) {
                        stage_b(k + 1, cur ^ 1);
                        load_a(k + 1, aset[u ^ 1]);
                    }
                    const u32x4* Bs = ldsB(cur);
                    u32x4 bf[4][2];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int term = 0; term < 2; ++term) bf[t][term] = Bs[slot<kTN>(term, 32 * t + li, kg)];
#define STEP(AT, BT)                                                                                                             \
                    _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                                \
                    _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                                \
                        acc[h][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(aset[u][h][AT]), as_f16x8(bf[t][BT]), acc[h][t], 0, 0, 0);
                    STEP(1, 0) STEP(0, 1) STEP(0, 0)
#undef STEP
                    __builtin_amdgcn_sched_barrier(0);
                    __syncthreads();
                }
            }
        }
    }

    if (PRIO) __builtin_amdgcn_s_setprio(0);
    if (g.trace) t_loop = __builtin_readcyclecounter();
    auto finish = [&]() {
        if (g.trace && threadIdx.x == 0) {
            const unsigned long long t_issue = __builtin_readcyclecounter();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t_done = __builtin_readcyclecounter();
            unsigned long long* t = g.trace + (size_t)blockIdx.x * 7;
            t[0] = hwid; t[1] = t_start; t[2] = t_loop; t[3] = t_issue; t[4] = t_done; t[5] = r_start; t[6] = __builtin_amdgcn_s_memrealtime();
        }
    };
    // ---- epilogue ----------------------------------------------------------------------------------------------------------
    const long rowbase = (long)b * g.HW;
    const float up1 = 1.0f, up2 = g.scale;
    if (EPI == 1) {
        float s = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[h][t][r];
        if (s == 12345.678f) g.out0[lane] = s;
        finish();
        return;
    }
    if (EPI == 0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x16 (&acc_h)[4] = acc[h];
            const int wrow0 = i0 + 64 * wave + 32 * h;
            const int x = 32 * tc + li;
            const int yb = 4 * tr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = wrow0 + (r & 3) + 8 * (r >> 2) + 4 * kg;
                const bool iok = i < g.HW;
                float s4 = 0.f;
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const float v0 = acc_h[2 * p][r], v1 = acc_h[2 * p + 1][r];
                    const int y = yb + 2 * p;
                    if (iok && x < g.W) {
                        float* o = g.out0 + (rowbase + i) * g.n + (long)y * g.W + x;
                        if (y < g.H) __builtin_nontemporal_store(v0 * up1 * up2, o);
                        if (y + 1 < g.H) __builtin_nontemporal_store(v1 * up1 * up2, o + g.W);
                    }
                    float s2 = v0 + v1;
                    s2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s2), 0xB1, 0xf, 0xf, false));
                    s4 += s2;
                    if (g.out1 && iok && !(lane & 1)) {
                        const int y1 = (yb >> 1) + p, x1 = x >> 1;
                        if (y1 < g.h1 && x1 < g.w1)
                            __builtin_nontemporal_store(s2 * up1 * (0.25f * up2), g.out1 + (rowbase + i) * ((long)g.h1 * g.w1) + (long)y1 * g.w1 + x1);
                    }
                }
                s4 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s4), 0x4E, 0xf, 0xf, false));
                if (g.out2 && iok && !(lane & 3)) {
                    const int y2 = yb >> 2, x2 = x >> 2;
                    if (y2 < g.h2 && x2 < g.w2)
                        __builtin_nontemporal_store(s4 * up1 * (0.0625f * up2), g.out2 + (rowbase + i) * ((long)g.h2 * g.w2) + (long)y2 * g.w2 + x2);
                }
            }
        }
        finish();
        return;
    }
    if (EPI == 2) {
        // One buffer resource per level, based at the tile's first row; rows past HW fall outside num_records and are dropped by the
        // range check (which looks at the VGPR offset only, so the row part is added there: one v_add with a scalar operand per store).
        const int rows_left = g.HW - i0;   // > 0
        const long n1 = (long)g.h1 * g.w1, n2 = (long)g.h2 * g.w2;
        const unsigned kDrop = 0x80000000u;   // + any row offset of the tile stays past every range and below 2^32
        const __amdgpu_buffer_rsrc_t r0 = make_rsrc(g.out0 + (rowbase + i0) * g.n, (unsigned)min((long)rows_left, (long)kTM) * g.n * 4u);
        const __amdgpu_buffer_rsrc_t r1 = make_rsrc(g.out1 + (rowbase + i0) * n1, (unsigned)(min((long)rows_left, (long)kTM) * n1 * 4));
        const __amdgpu_buffer_rsrc_t r2 = make_rsrc(g.out2 + (rowbase + i0) * n2, (unsigned)(min((long)rows_left, (long)kTM) * n2 * 4));
        const int x = 32 * tc + li, yb = 4 * tr;
        const unsigned pitch0 = (unsigned)g.n * 4u, pitch1 = (unsigned)n1 * 4u, pitch2 = (unsigned)n2 * 4u;
        const int x1 = x >> 1, x2 = x >> 2, y2 = yb >> 2;
        // lane parts (dropped columns aim past every range)
        unsigned l0[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
            l0[t] = (x < g.W && yb + t < g.H) ? ((unsigned)(yb + t) * g.W + x) * 4u + (unsigned)(4 * kg) * pitch0 : kDrop;
        unsigned l1[2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
            l1[p] = (!(lane & 1) && x1 < g.w1 && (yb >> 1) + p < g.h1) ? ((unsigned)((yb >> 1) + p) * g.w1 + x1) * 4u + (unsigned)(4 * kg) * pitch1 : kDrop;
        const unsigned l2 = (!(lane & 3) && x2 < g.w2 && y2 < g.h2) ? ((unsigned)y2 * g.w2 + x2) * 4u + (unsigned)(4 * kg) * pitch2 : kDrop;
        const float c0 = up1 * up2, c1 = up1 * (0.25f * up2), c2 = up1 * (0.0625f * up2);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x16 (&acc_h)[4] = acc[h];
            const unsigned wrow = (unsigned)(64 * wave + 32 * h);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned row = wrow + (unsigned)((r & 3) + 8 * (r >> 2));   // + 4 kg is in the lane part
                float s4 = 0.f;
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const float v0 = acc_h[2 * p][r], v1 = acc_h[2 * p + 1][r];
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0 * c0), r0, l0[2 * p] + row * pitch0, 0, AUX);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1 * c0), r0, l0[2 * p + 1] + row * pitch0, 0, AUX);
                    float s2 = v0 + v1;
                    s2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s2), 0xB1, 0xf, 0xf, false));
                    s4 += s2;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s2 * c1), r1, l1[p] + row * pitch1, 0, AUX);
                }
                s4 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s4), 0x4E, 0xf, 0xf, false));
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s4 * c2), r2, l2 + row * pitch2, 0, AUX);
            }
        }
        finish();
    }
}

// ---- the same kernel with 128-row tiles: a wave owns 32 rows x 128 columns (64 accumulator registers), four workgroups per CU ----
constexpr int sTM = 128, sGranA = 2 * sTM * 2, sGranB = kGranB, sStage = sGranA + sGranB;   // 16 KiB per stage
template <int WGS>
__global__ void __launch_bounds__(kThreads, WGS) gemm3_small(const Args g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32x4* const lds0 = reinterpret_cast<u32x4*>(smem_raw);
    auto ldsA = [&](int stage) { return lds0 + stage * sStage; };
    auto ldsB = [&](int stage) { return lds0 + stage * sStage + sGranA; };
    const unsigned lb = xcd_contiguous_block(blockIdx.x, g.nblocks);
    const int tiles_m = (g.HW + sTM - 1) / sTM;
    const int tiles_n = g.tiles_r * g.tiles_c, row_groups = (tiles_m + kRowGroup - 1) / kRowGroup;
    const int tm = ((lb / (kRowGroup * tiles_n)) % row_groups) * kRowGroup + lb % kRowGroup;
    const int tn = (lb / kRowGroup) % tiles_n;
    const int b = lb / (kRowGroup * tiles_n * row_groups);
    if (tm >= tiles_m) return;
    const int tr = tn / g.tiles_c, tc = tn % g.tiles_c;
    const int i0 = tm * sTM;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int drow = lane >> 1, dhalf = lane & 1;
    const uint16_t* asrc[2];
    const uint16_t* bsrc[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int ins = wave + 4 * e, term = ins >> 2, row = 32 * (ins & 3) + drow;
        const int half = dhalf ^ ((row >> 3) & 1);
        asrc[e] = g.a + ((((long)b * g.KC) * 2 + term) * g.HW + (i0 + row < g.HW ? i0 + row : 0)) * 16 + half * 8;
        const int y = 4 * tr + (row >> 5), x = 32 * tc + (row & 31);
        const bool ok = y < g.H && x < g.W;
        const long col = (long)y * g.W + x;
        bsrc[e] = g.b + ((((long)b * g.KC) * 2 + term) * g.n + (ok ? col : 0)) * 16 + half * 8;
    }
    const long a_step = 2L * g.HW * 16, b_step = 2L * g.n * 16;
    auto stage_in = [&](int kc, int stage) {
#pragma unroll
        for (int e = 0; e < 2; ++e) dma16(asrc[e] + kc * a_step, ldsA(stage) + (wave + 4 * e) * 64);
#pragma unroll
        for (int e = 0; e < 2; ++e) dma16(bsrc[e] + kc * b_step, ldsB(stage) + (wave + 4 * e) * 64);
    };
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int kg = lane >> 5, li = lane & 31;
    stage_in(0, 0);
    __syncthreads();
    for (int kc = 0; kc < g.KC; ++kc) {
        const int cur = kc & 1;
        if (kc + 1 < g.KC) stage_in(kc + 1, cur ^ 1);
        const u32x4* As = ldsA(cur);
        const u32x4* Bs = ldsB(cur);
        u32x4 af[2], bf[4][2];
#pragma unroll
        for (int term = 0; term < 2; ++term) af[term] = As[slot<sTM>(term, 32 * wave + li, kg)];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int term = 0; term < 2; ++term) bf[t][term] = Bs[slot<kTN>(term, 32 * t + li, kg)];
#define STEP(AT, BT)                                                                                                             \
        _Pragma("unroll") for (int t = 0; t < 4; ++t)                                                                            \
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(af[AT]), as_f16x8(bf[t][BT]), acc[t], 0, 0, 0);
        STEP(1, 0) STEP(0, 1) STEP(0, 0)
#undef STEP
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    const long rowbase = (long)b * g.HW;
    const float up1 = 1.0f, up2 = g.scale;
    const int rows_left = g.HW - i0;
    const long n1 = (long)g.h1 * g.w1, n2 = (long)g.h2 * g.w2;
    const unsigned kDrop = 0x80000000u;
    const __amdgpu_buffer_rsrc_t r0 = make_rsrc(g.out0 + (rowbase + i0) * g.n, (unsigned)min((long)rows_left, (long)sTM) * g.n * 4u);
    const __amdgpu_buffer_rsrc_t r1 = make_rsrc(g.out1 + (rowbase + i0) * n1, (unsigned)(min((long)rows_left, (long)sTM) * n1 * 4));
    const __amdgpu_buffer_rsrc_t r2 = make_rsrc(g.out2 + (rowbase + i0) * n2, (unsigned)(min((long)rows_left, (long)sTM) * n2 * 4));
    const int x = 32 * tc + li, yb = 4 * tr;
    const unsigned pitch0 = (unsigned)g.n * 4u, pitch1 = (unsigned)n1 * 4u, pitch2 = (unsigned)n2 * 4u;
    const int x1 = x >> 1, x2 = x >> 2, y2 = yb >> 2;
    unsigned l0[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
        l0[t] = (x < g.W && yb + t < g.H) ? ((unsigned)(yb + t) * g.W + x) * 4u + (unsigned)(4 * kg) * pitch0 : kDrop;
    unsigned l1[2];
#pragma unroll
    for (int p = 0; p < 2; ++p)
        l1[p] = (!(lane & 1) && x1 < g.w1 && (yb >> 1) + p < g.h1) ? ((unsigned)((yb >> 1) + p) * g.w1 + x1) * 4u + (unsigned)(4 * kg) * pitch1 : kDrop;
    const unsigned l2 = (!(lane & 3) && x2 < g.w2 && y2 < g.h2) ? ((unsigned)y2 * g.w2 + x2) * 4u + (unsigned)(4 * kg) * pitch2 : kDrop;
    const float c0 = up1 * up2, c1 = up1 * (0.25f * up2), c2 = up1 * (0.0625f * up2);
    const unsigned wrow = (unsigned)(32 * wave);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const unsigned row = wrow + (unsigned)((r & 3) + 8 * (r >> 2));
        float s4 = 0.f;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float v0 = acc[2 * p][r], v1 = acc[2 * p + 1][r];
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v0 * c0), r0, l0[2 * p] + row * pitch0, 0, 2);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v1 * c0), r0, l0[2 * p + 1] + row * pitch0, 0, 2);
            float s2 = v0 + v1;
            s2 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s2), 0xB1, 0xf, 0xf, false));
            s4 += s2;
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s2 * c1), r1, l1[p] + row * pitch1, 0, 2);
        }
        s4 += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s4), 0x4E, 0xf, 0xf, false));
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s4 * c2), r2, l2 + row * pitch2, 0, 2);
    }
}

template <int WGS>
float run_small(const Args& g0, int reps, hipStream_t st) {
    Args g = g0;
    const int tiles_m = (g.HW + sTM - 1) / sTM;
    g.nblocks = (unsigned)((long)((tiles_m + kRowGroup - 1) / kRowGroup) * kRowGroup * g.tiles_r * g.tiles_c * g.B);
    const int lds = 2 * sStage * 16;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm3_small<WGS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((gemm3_small<WGS>), dim3(g.nblocks), dim3(kThreads), lds, st, g);
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm3_small<WGS>), dim3(g.nblocks), dim3(kThreads), lds, st, g);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

__global__ void fill_kernel(uint16_t* p, long n, unsigned seed, float mag) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        const float u = ((h & 0xffffff) / 16777216.0f - 0.5f) * 2.0f * mag;
        const _Float16 v = (_Float16)u;
        p[i] = __builtin_bit_cast(uint16_t, v);
    }
}

template <int LOOP, int EPI, int PRIO = 0, int AUX = 2>
float run(const Args& g, int lds_stages, int reps, hipStream_t st) {
    const int lds = lds_stages * (LOOP == 3 ? kGranB : kStageGran) * 16;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm3<LOOP, EPI, PRIO, AUX>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((gemm3<LOOP, EPI, PRIO, AUX>), dim3(g.nblocks), dim3(kThreads), lds, st, g);
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm3<LOOP, EPI, PRIO, AUX>), dim3(g.nblocks), dim3(kThreads), lds, st, g);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

static double max_diff(const float* a, const float* b, long off, long n) {
    std::vector<float> ha(n), hb(n);
    CK(hipMemcpy(ha.data(), a + off, n * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hb.data(), b + off, n * 4, hipMemcpyDeviceToHost));
    double m = 0;
    for (long i = 0; i < n; ++i) {
        const double d = std::fabs((double)ha[i] - (double)hb[i]);
        if (!(d <= m)) m = d;   // NaN propagates
    }
    return m;
}

int main(int argc, char** argv) {
    const int B = 4, C = 256, H = 90, W = 160, KC = C / 16, HW = H * W;
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    const long nsplit = (long)B * KC * 2 * HW * 16;
    uint16_t *a, *b;
    CK(hipMalloc(&a, nsplit * 2));
    CK(hipMalloc(&b, nsplit * 2));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, a, nsplit, 1u, 0.5f);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, st, b, nsplit, 2u, 0.5f);
    const int h1 = H / 2, w1 = W / 2, h2 = h1 / 2, w2 = w1 / 2;
    const long n0 = (long)B * HW * HW, n1 = (long)B * HW * h1 * w1, n2 = (long)B * HW * h2 * w2;
    float *ref0, *ref1, *ref2, *o0, *o1, *o2;
    CK(hipMalloc(&ref0, n0 * 4)); CK(hipMalloc(&ref1, n1 * 4)); CK(hipMalloc(&ref2, n2 * 4));
    CK(hipMalloc(&o0, n0 * 4)); CK(hipMalloc(&o1, n1 * 4)); CK(hipMalloc(&o2, n2 * 4));
    Args g;
    g.a = a; g.b = b;
    g.B = B; g.KC = KC; g.HW = HW; g.H = H; g.W = W; g.n = HW;
    g.h1 = h1; g.w1 = w1; g.h2 = h2; g.w2 = w2;
    g.tiles_m = (HW + kTM - 1) / kTM; g.tiles_r = (H + 3) / 4; g.tiles_c = (W + 31) / 32;
    g.scale = 1.0f / 16.0f;
    g.stagger = 0;
    g.nblocks = (unsigned)((long)((g.tiles_m + kRowGroup - 1) / kRowGroup) * kRowGroup * g.tiles_r * g.tiles_c * B);

    auto report = [&](const char* name, float ms, bool check) {
        double d0 = -1, d1 = -1, d2 = -1;
        if (check) {
            // the first 1500 and the last 700 rows of the volume (the partial row tile of the last batch item is among them)
            const long rows_a = 1500, rows_b = 700, rows = (long)B * HW;
            const long p0 = HW, p1 = (long)h1 * w1, p2 = (long)h2 * w2;
            d0 = std::fmax(max_diff(ref0, o0, 0, rows_a * p0), max_diff(ref0, o0, (rows - rows_b) * p0, rows_b * p0));
            d1 = std::fmax(max_diff(ref1, o1, 0, rows_a * p1), max_diff(ref1, o1, (rows - rows_b) * p1, rows_b * p1));
            d2 = std::fmax(max_diff(ref2, o2, 0, rows_a * p2), max_diff(ref2, o2, (rows - rows_b) * p2, rows_b * p2));
        }
        printf("{\"experiment\": \"corr_gemm_exp\", \"variant\": \"%s\", \"ms\": %.4f, \"max_diff_l0\": %g, \"max_diff_l1\": %g, \"max_diff_l2\": %g}\n",
               name, ms, d0, d1, d2);
        fflush(stdout);
    };
    g.out0 = ref0; g.out1 = ref1; g.out2 = ref2;
    float ms = run<0, 0>(g, 2, reps, st);
    report("LOOP0 EPI0 (library)", ms, false);
    g.out0 = o0; g.out1 = o1; g.out2 = o2;
    auto clear = [&]() { CK(hipMemsetAsync(o0, 0xff, n0 * 4, st)); CK(hipMemsetAsync(o1, 0xff, n1 * 4, st)); CK(hipMemsetAsync(o2, 0xff, n2 * 4, st)); };
    ms = run<0, 1>(g, 2, reps, st); report("LOOP0 EPI1 (loop alone)", ms, false);
    ms = run<9, 0>(g, 2, reps, st); report("LOOP9 EPI0 (epilogue alone)", ms, false);
    ms = run<9, 2>(g, 2, reps, st); report("LOOP9 EPI2 (buffer-store epilogue alone)", ms, false);
    ms = run<1, 1>(g, 2, reps, st); report("LOOP1 EPI1 (register double buffer, loop alone)", ms, false);
    ms = run<2, 1>(g, 3, reps, st); report("LOOP2 EPI1 (register double buffer, 3 stages, loop alone)", ms, false);
    clear(); ms = run<0, 2>(g, 2, reps, st); report("LOOP0 EPI2", ms, true);
    clear(); ms = run<1, 0>(g, 2, reps, st); report("LOOP1 EPI0", ms, true);
    clear(); ms = run<1, 2>(g, 2, reps, st); report("LOOP1 EPI2", ms, true);
    clear(); ms = run<2, 0>(g, 3, reps, st); report("LOOP2 EPI0", ms, true);
    clear(); ms = run<2, 2>(g, 3, reps, st); report("LOOP2 EPI2", ms, true);
    ms = run<3, 1>(g, 2, reps, st); report("LOOP3 EPI1 (A straight from memory, loop alone)", ms, false);
    clear(); ms = run<3, 0>(g, 2, reps, st); report("LOOP3 EPI0", ms, true);
    clear(); ms = run<3, 2>(g, 2, reps, st); report("LOOP3 EPI2", ms, true);
    for (int rep = 0; rep < 2; ++rep) {
        clear(); ms = run_small<4>(g, reps, st); report("128-row tiles, 4 workgroups per CU, EPI2", ms, true);
        clear(); ms = run_small<3>(g, reps, st); report("128-row tiles, 3 workgroups per CU, EPI2", ms, true);
        clear(); ms = run<0, 2>(g, 2, reps, st); report("LOOP0 EPI2 (again)", ms, true);
    }
    clear(); ms = run<0, 2, 3>(g, 2, reps, st); report("LOOP0 EPI2 setprio 3 in the loop", ms, true);
    clear(); ms = run<0, 2, 1>(g, 2, reps, st); report("LOOP0 EPI2 setprio 1 in the loop", ms, true);
    clear(); ms = run<0, 2, 0, 0>(g, 2, reps, st); report("LOOP0 EPI2 plain stores (no nt)", ms, true);
    clear(); ms = run<0, 2, 0, 3>(g, 2, reps, st); report("LOOP0 EPI2 stores sc0 nt", ms, true);
    clear(); ms = run<0, 2, 0, 18>(g, 2, reps, st); report("LOOP0 EPI2 stores nt sc1", ms, true);
    for (int sg : {4}) {
        g.stagger = sg;
        char name[96];
        clear(); ms = run<0, 2>(g, 2, reps, st); snprintf(name, sizeof name, "LOOP0 EPI2 stagger %d", sg); report(name, ms, true);
        clear(); ms = run<3, 2>(g, 2, reps, st); snprintf(name, sizeof name, "LOOP3 EPI2 stagger %d", sg); report(name, ms, true);
    }
    g.stagger = 0;
    // ---- trace: who is in which phase when (CSV to stderr-named file) -------------------------------------------------------------
    unsigned long long* tr;
    unsigned* tk;
    CK(hipMalloc(&tr, (size_t)g.nblocks * 7 * 8));
    CK(hipMalloc(&tk, 2048 * 4));
    for (int pass = 0; pass < 3; ++pass) {
        g.trace = tr;
        g.tickets = pass == 2 ? tk : nullptr;
        g.stagger = pass == 0 ? 0 : 4;
        CK(hipMemsetAsync(tr, 0, (size_t)g.nblocks * 7 * 8, st));
        CK(hipMemsetAsync(tk, 0, 2048 * 4, st));
        const int lds = 2 * kStageGran * 16;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL((gemm3<0, 2>), dim3(g.nblocks), dim3(kThreads), lds, st, g);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms1; CK(hipEventElapsedTime(&ms1, e0, e1));
        std::vector<unsigned long long> h((size_t)g.nblocks * 7);
        CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
        char fn[64];
        snprintf(fn, sizeof fn, "corr_trace_%d.csv", pass);
        FILE* f = fopen(fn, "w");
        fprintf(f, "# LOOP0 EPI2, %s, single launch %.4f ms\nblock,hwid,start,loop_end,issue_end,done,rt_start,rt_end\n", pass == 0 ? "no stagger" : pass == 1 ? "stagger 4 by block index" : "stagger 4 by CU ticket", ms1);
        for (unsigned i = 0; i < g.nblocks; ++i)
            if (h[i * 7 + 1]) fprintf(f, "%u,%llu,%llu,%llu,%llu,%llu,%llu,%llu\n", i, h[i * 7], h[i * 7 + 1], h[i * 7 + 2], h[i * 7 + 3], h[i * 7 + 4], h[i * 7 + 5], h[i * 7 + 6]);
        fclose(f);
        printf("{\"experiment\": \"corr_gemm_exp\", \"trace\": \"%s\", \"ms_single_launch\": %.4f}\n", fn, ms1);
    }
    // ticket stagger timed
    g.trace = nullptr; g.tickets = tk;
    for (int sg : {3, 5}) {
        g.stagger = sg;
        float tot = 0;
        for (int i = 0; i < 10; ++i) {
            CK(hipMemsetAsync(tk, 0, 2048 * 4, st));
            tot += run<0, 2>(g, 2, 1, st);   // NOTE: run() warms up 5 times with stale tickets; the timed launch follows a fresh memset only if reps == 1 and warm-ups are harmless
        }
        printf("{\"experiment\": \"corr_gemm_exp\", \"variant\": \"LOOP0 EPI2 ticket stagger %d (tickets not re-zeroed between warm-ups)\", \"ms\": %.4f}\n", sg, tot / 10);
    }
    return 0;
}

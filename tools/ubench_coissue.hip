// ubench_coissue.hip - how much other work fits beside the matrix pipe on one gfx950 SIMD?
//
// A machine model for DESIGN.md 4.1b.  Every wave runs ITER x 16 MFMAs on independent accumulators; behind each MFMA it issues NV
// "filler" instructions of one KIND, each independent of everything near it (16 rotating registers), pinned in place by being in
// the same asm volatile block order.  Waves per SIMD = 1 or 2 (block of 256 / 512 threads, one block per CU).
// Output: nanoseconds and core cycles (at the clock measured by an MFMA-only run of known cost) per MFMA per SIMD.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/ubench_coissue tools/ubench_coissue.hip && tools/_build/ubench_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Kind { K_NONE = 0, K_FMA, K_PKFMA, K_EXP, K_CVT, K_DSREAD, K_ACCREAD, K_MIX, K_GLOAD, K_LDSDMA, K_DOT2 };
static const char* kind_name[] = {"none", "v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_cvt_pk_bf16_f32", "ds_read_b128",
                                  "v_mov from acc", "softplus+split mix", "global_load_dwordx4", "global_load_lds_dwordx4", "v_dot2c_f32_bf16"};

struct Fill {
    float f[16];
    f32x2 p[8];
    u32x4 d[4];
    unsigned lds_addr;
    u32x4 g[4];            // staging registers of the "global load + ds_write" form of the weight stream
    const char* gsrc;      // this iteration's window of the 2 MiB weight-like buffer (wave-uniform)
    unsigned voff;         // lane * 16
};

template <int KIND, int I>
__device__ __forceinline__ void filler(Fill& s, f32x4* acc) {
    constexpr int r = I & 15;
    if constexpr (KIND == K_FMA) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s.f[r]) : "v"(s.f[(r + 8) & 15]), "v"(s.f[(r + 4) & 15]));
    } else if constexpr (KIND == K_PKFMA) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(s.p[r & 7]) : "v"(s.p[(r + 4) & 7]));
    } else if constexpr (KIND == K_EXP) {
        asm volatile("v_exp_f32 %0, %0" : "+v"(s.f[r]));
    } else if constexpr (KIND == K_CVT) {
        unsigned o;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o) : "v"(s.f[r]), "v"(s.f[(r + 1) & 15]));
        asm volatile("" ::"v"(o));
    } else if constexpr (KIND == K_DSREAD) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(s.d[r & 3]) : "v"(s.lds_addr));
    } else if constexpr (KIND == K_ACCREAD) {
        // stands for the epilogue fetching pre-activations out of the accumulator half
        asm volatile("v_mov_b32 %0, %1" : "=v"(s.f[r]) : "v"(acc[r & 7][I & 3]));
    } else if constexpr (KIND == K_GLOAD) {
        // 1 KiB per instruction straight into registers, streaming through an L2-resident buffer every workgroup reads
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(s.d[r & 3]) : "v"(s.voff), "s"(s.gsrc + (I & ~3) * 1024), "i"((I & 3) * 1024) : "memory");
    } else if constexpr (KIND == K_LDSDMA) {
        // the same 1 KiB into LDS (M0 = LDS base, set once before the loop)
        asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(s.voff), "s"(s.gsrc + (I & ~3) * 1024), "i"((I & 3) * 1024) : "memory");
    } else if constexpr (KIND == K_DOT2) {
        asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s.f[r]) : "v"(s.lds_addr), "v"(s.voff));
    } else if constexpr (KIND == K_MIX) {
        // the K2 epilogue's instruction mix per pair of values, roughly: 2 exp, 2 log, 6 plain, 1 cvt  (11 instructions)
        constexpr int m = I % 11;
        if constexpr (m == 0 || m == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(s.f[r]));
        else if constexpr (m == 2 || m == 3) asm volatile("v_log_f32 %0, %0" : "+v"(s.f[r]));
        else if constexpr (m == 4) {
            unsigned o;
            asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o) : "v"(s.f[r]), "v"(s.f[(r + 1) & 15]));
            asm volatile("" ::"v"(o));
        } else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s.f[r]) : "v"(s.f[(r + 8) & 15]), "v"(s.f[(r + 4) & 15]));
    }
}

template <int KIND, int NV, int I0>
__device__ __forceinline__ void fillers(Fill& s, f32x4* acc) {
    if constexpr (NV > 0) {
        filler<KIND, I0>(s, acc);
        fillers<KIND, NV - 1, I0 + 1>(s, acc);
    }
}

template <int KIND, int NV, int M>
struct Body0 {
    static __device__ __forceinline__ void run(f32x4 (&acc)[16], const u32x4& a, const u32x4& b, Fill& s) {
        if constexpr (M < 16) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[M]) : "v"(a), "v"(b));
            fillers<KIND, NV, M * NV>(s, acc);
            Body0<KIND, NV, M + 1>::run(acc, a, b, s);
        }
    }
};

template <int KIND, int NV, int M>
struct Body1 {
    static __device__ __forceinline__ void run(f32x16 (&acc)[8], f32x4 (&dummy)[8], const u32x4& a, const u32x4& b, Fill& s) {
        if constexpr (M < 8) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[M]) : "v"(a), "v"(b));
            fillers<KIND, NV, M * NV>(s, dummy);
            Body1<KIND, NV, M + 1>::run(acc, dummy, a, b, s);
        }
    }
};

// SHAPE 0: v_mfma_f32_16x16x32_bf16, 16 accumulators of 4 registers.  SHAPE 1: v_mfma_f32_32x32x16_bf16, 8 accumulators of 16 (AGPR).
template <int SHAPE, int KIND, int NV, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 1) k_bench(float* out, int iters, const char* gbuf) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 1.0f;
    __syncthreads();
    Fill s;
#pragma unroll
    for (int i = 0; i < 16; ++i) s.f[i] = 0.001f * (float)(threadIdx.x + i);
#pragma unroll
    for (int i = 0; i < 8; ++i) s.p[i] = f32x2{0.5f, 0.25f};
#pragma unroll
    for (int i = 0; i < 4; ++i) s.d[i] = u32x4{0, 0, 0, 0};
    s.lds_addr = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;
    s.voff = (threadIdx.x & 63) * 16;
    s.gsrc = gbuf;
    constexpr unsigned WINDOW = 16 * NV * 1024;          // bytes one wave touches per iteration
    const unsigned wave_skew = (threadIdx.x >> 6) * 4096;
    if constexpr (KIND == K_LDSDMA) asm volatile("s_mov_b32 m0, %0" ::"s"((unsigned)(size_t)lds) : "memory");
    u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
    if constexpr (SHAPE == 0) {
        f32x4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
            if constexpr (SHAPE == 0) {}
            Body0<KIND, NV, 0>::run(acc, a, b, s);
            if constexpr (KIND == K_DSREAD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (KIND == K_GLOAD || KIND == K_LDSDMA) {
                // one iteration's loads stay in flight: the loop measures issue cost and throughput, not the L2 round trip
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SHAPE == 0 ? 16 : 8) * NV) : "memory");
                s.gsrc = gbuf + __builtin_amdgcn_readfirstlane(((unsigned)(it + 1) * WINDOW + wave_skew) & (2u * 1024 * 1024 - 1));
            }
        }
        float r = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][3];
#pragma unroll
        for (int i = 0; i < 16; ++i) r += s.f[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) r += s.p[i][0];
#pragma unroll
        for (int i = 0; i < 4; ++i) r += __uint_as_float(s.d[i][0]);
        if (r == 123.456f) out[0] = r;
    } else {
        f32x16 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        f32x4 dummy[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) dummy[i] = f32x4{1, 2, 3, 4};
        for (int it = 0; it < iters; ++it) {
            Body1<KIND, NV, 0>::run(acc, dummy, a, b, s);
            if constexpr (KIND == K_DSREAD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (KIND == K_GLOAD || KIND == K_LDSDMA) {
                // one iteration's loads stay in flight: the loop measures issue cost and throughput, not the L2 round trip
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SHAPE == 0 ? 16 : 8) * NV) : "memory");
                s.gsrc = gbuf + __builtin_amdgcn_readfirstlane(((unsigned)(it + 1) * WINDOW + wave_skew) & (2u * 1024 * 1024 - 1));
            }
        }
        float r = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][15];
#pragma unroll
        for (int i = 0; i < 16; ++i) r += s.f[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) r += s.p[i][0];
#pragma unroll
        for (int i = 0; i < 4; ++i) r += __uint_as_float(s.d[i][0]);
        if (r == 123.456f) out[0] = r;
    }
}

struct Row { int shape, kind, nv, waves; double ns_per_mfma_simd; };
static std::vector<Row> rows;
static float* d_out;
static char* d_buf;

template <int SHAPE, int KIND, int NV, int WAVES>
static void run() {
    const int iters = 2000, per_iter = SHAPE == 0 ? 16 : 8, blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_bench<SHAPE, KIND, NV, WAVES><<<blocks, 64 * WAVES>>>(d_out, 50, d_buf);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        k_bench<SHAPE, KIND, NV, WAVES><<<blocks, 64 * WAVES>>>(d_out, iters, d_buf);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // a SIMD hosts WAVES / 4 of the block's waves, each doing iters * per_iter MFMAs
    const double mfma_per_simd = (double)iters * per_iter * (WAVES / 4);
    rows.push_back({SHAPE, KIND, NV, WAVES / 4, best * 1e6 / mfma_per_simd});
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

template <int SHAPE, int KIND, int WAVES>
static void sweep() {
    run<SHAPE, KIND, 1, WAVES>();
    run<SHAPE, KIND, 2, WAVES>();
    run<SHAPE, KIND, 3, WAVES>();
    run<SHAPE, KIND, 4, WAVES>();
    run<SHAPE, KIND, 6, WAVES>();
    run<SHAPE, KIND, 8, WAVES>();
    if constexpr (SHAPE == 1) { run<SHAPE, KIND, 12, WAVES>(); run<SHAPE, KIND, 16, WAVES>(); }
}

template <int SHAPE, int WAVES>
static void all_kinds() {
    run<SHAPE, K_NONE, 0, WAVES>();
    sweep<SHAPE, K_FMA, WAVES>();
    sweep<SHAPE, K_PKFMA, WAVES>();
    sweep<SHAPE, K_EXP, WAVES>();
    sweep<SHAPE, K_CVT, WAVES>();
    sweep<SHAPE, K_DSREAD, WAVES>();
    if constexpr (SHAPE == 0) sweep<SHAPE, K_ACCREAD, WAVES>();
    sweep<SHAPE, K_MIX, WAVES>();
    sweep<SHAPE, K_DOT2, WAVES>();
    run<SHAPE, K_GLOAD, 1, WAVES>();
    run<SHAPE, K_GLOAD, 2, WAVES>();
    run<SHAPE, K_LDSDMA, 1, WAVES>();
    run<SHAPE, K_LDSDMA, 2, WAVES>();
}

// ---------------------------------------------------------------------------------------------------------------------
// Combination runs: per 12 MFMAs, V12 VALU instructions in the proportions of the K2 epilogue, D12 fragment reads
// (ds_read_b128) and G12 LDS-DMA pieces, all spread evenly behind the MFMAs and all independent of each other - the issue
// capacity of a SIMD for K2's actual mix with every latency hidden (an upper bound for any schedule of that mix).
//   PAT 0: today's epilogue per pair of values - 2 exp, 2 log, 2 cvt_pk, 16 plain (22);  PAT 1: a leaner one - 2 exp, 2 log,
//   2 cvt_pk, 6 plain, 2 v_dot2c_f32_bf16 (14).
template <int PAT, int I>
__device__ __forceinline__ void mix_op(Fill& s) {
    constexpr int L = PAT == 0 ? 22 : 14;
    constexpr int m = I % L, r = I & 15;
    constexpr bool is_exp = (m == 0 || m == L / 2), is_log = (m == 3 || m == L / 2 + 3), is_cvt = (m == 6 || m == L / 2 + 6),
                   is_dot = PAT == 1 && (m == 5 || m == L / 2 + 5);
    if constexpr (is_exp) asm volatile("v_exp_f32 %0, %0" : "+v"(s.f[r]));
    else if constexpr (is_log) asm volatile("v_log_f32 %0, %0" : "+v"(s.f[r]));
    else if constexpr (is_cvt) {
        unsigned o;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o) : "v"(s.f[r]), "v"(s.f[(r + 1) & 15]));
        asm volatile("" ::"v"(o));
    } else if constexpr (is_dot) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(s.f[r]) : "v"(s.lds_addr), "v"(s.voff));
    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s.f[r]) : "v"(s.f[(r + 8) & 15]), "v"(s.f[(r + 4) & 15]));
}
template <int PAT, int I0, int N>
__device__ __forceinline__ void mix_ops(Fill& s) {
    if constexpr (N > 0) { mix_op<PAT, I0>(s); mix_ops<PAT, I0 + 1, N - 1>(s); }
}
template <int I0, int N>
__device__ __forceinline__ void ds_ops(Fill& s) {
    if constexpr (N > 0) { asm volatile("ds_read_b128 %0, %1" : "=v"(s.d[I0 & 3]) : "v"(s.lds_addr)); ds_ops<I0 + 1, N - 1>(s); }
}
// GM 0: LDS-DMA.  GM 1: the same KiB through registers (ds_write_b128 of the piece loaded a round earlier, then the load; the
// write does not wait for the load here - timing only).  GM 2: LDS-DMA behind an s_waitcnt lgkmcnt(0) (no LDS read in flight).
template <int I0, int N, int GM = 0>
__device__ __forceinline__ void dma_ops(Fill& s) {
    if constexpr (N > 0) {
        if constexpr (GM == 1) {
            asm volatile("ds_write_b128 %0, %1 offset:8192" ::"v"(s.lds_addr), "v"(s.g[I0 & 3]) : "memory");
            asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(s.g[I0 & 3]) : "v"(s.voff), "s"(s.gsrc + (I0 & ~3) * 1024), "i"((I0 & 3) * 1024) : "memory");
        } else {
            if constexpr (GM == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(s.voff), "s"(s.gsrc + (I0 & ~3) * 1024), "i"((I0 & 3) * 1024) : "memory");
        }
        dma_ops<I0 + 1, N - 1, GM>(s);
    }
}
constexpr int upto(int m, int total) { return (m * total) / 12; }     // how many of `total` have been issued before MFMA m of 12

template <int SHAPE, int PAT, int V12, int D12, int G12, int M>
struct ComboBody {
    template <class ACC>
    static __device__ __forceinline__ void run(ACC& acc, const u32x4& a, const u32x4& b, Fill& s) {
        if constexpr (M < 12) {
            if constexpr (SHAPE == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[M]) : "v"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[M & 7]) : "v"(a), "v"(b));
            constexpr int NG = upto(M + 1, G12) - upto(M, G12), NVV = upto(M + 1, V12) - upto(M, V12);
            if constexpr ((PAT & 32) != 0 && NG > 0) asm volatile("s_nop 7");                                 // pat & 32: 8 idle cycles, then the piece
            if constexpr ((PAT & (8 | 32)) != 0) dma_ops<upto(M, G12), NG, 0>(s);                             // pat & 8: the piece first
            ds_ops<upto(M, D12), upto(M + 1, D12) - upto(M, D12)>(s);
            if constexpr ((PAT & 16) != 0) dma_ops<upto(M, G12), NG, 0>(s);                                   // pat & 16: between reads and VALU
            if constexpr ((PAT & 64) != 0) {                                                                 // pat & 64: one VALU, the piece, the rest
                mix_ops<(PAT & 1), upto(M, V12), (NVV > 0 ? 1 : 0)>(s);
                dma_ops<upto(M, G12), NG, 0>(s);
                mix_ops<(PAT & 1), upto(M, V12) + (NVV > 0 ? 1 : 0), NVV - (NVV > 0 ? 1 : 0)>(s);
            } else mix_ops<(PAT & 1), upto(M, V12), NVV>(s);
            if constexpr ((PAT & (8 | 16 | 32 | 64)) == 0) dma_ops<upto(M, G12), NG, ((PAT >> 1) & 3)>(s);
            ComboBody<SHAPE, PAT, V12, D12, G12, M + 1>::run(acc, a, b, s);
        }
    }
};

template <int SHAPE, int PAT, int V12, int D12, int G12, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 1) k_combo(float* out, int iters, const char* gbuf) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 1.0f;
    __syncthreads();
    Fill s;
#pragma unroll
    for (int i = 0; i < 16; ++i) s.f[i] = 0.001f * (float)(threadIdx.x + i);
#pragma unroll
    for (int i = 0; i < 4; ++i) s.d[i] = u32x4{0, 0, 0, 0};
    s.lds_addr = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;
    s.voff = (threadIdx.x & 63) * 16;
    s.gsrc = gbuf;
#pragma unroll
    for (int i = 0; i < 4; ++i) s.g[i] = u32x4{0, 0, 0, 0};
    const unsigned wave_skew = (threadIdx.x >> 6) * 4096;
    asm volatile("s_mov_b32 m0, %0" ::"s"((unsigned)(size_t)lds + 8192) : "memory");
    u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
    float r = 0.f;
    if constexpr (SHAPE == 0) {
        f32x4 acc[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
            ComboBody<SHAPE, PAT, V12, D12, G12, 0>::run(acc, a, b, s);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (G12 > 0) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G12) : "memory");
                s.gsrc = gbuf + __builtin_amdgcn_readfirstlane(((unsigned)(it + 1) * (G12 * 1024u) * 8u + wave_skew) & (2u * 1024 * 1024 - 1));
            }
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) r += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
            ComboBody<SHAPE, PAT, V12, D12, G12, 0>::run(acc, a, b, s);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (G12 > 0) {
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G12) : "memory");
                s.gsrc = gbuf + __builtin_amdgcn_readfirstlane(((unsigned)(it + 1) * (G12 * 1024u) * 4u + wave_skew) & (2u * 1024 * 1024 - 1));
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][15];
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) r += s.f[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) r += __uint_as_float(s.d[i][0]) + __uint_as_float(s.g[i][1]);
    if (r == 123.456f) out[0] = r;
}

struct CRow { int shape, pat, v12, d12, g12, waves; double ns; };
static std::vector<CRow> crows;

template <int SHAPE, int PAT, int V12, int D12, int G12, int WAVES>
static void crun() {
    const int iters = 3000, blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_combo<SHAPE, PAT, V12, D12, G12, WAVES><<<blocks, 64 * WAVES>>>(d_out, 50, d_buf);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        k_combo<SHAPE, PAT, V12, D12, G12, WAVES><<<blocks, 64 * WAVES>>>(d_out, iters, d_buf);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    crows.push_back({SHAPE, PAT, V12, D12, G12, WAVES / 4, best * 1e6 / ((double)iters * 12 * (WAVES / 4))});
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

// exactness probe for the leaner split: y - float(hi) by v_dot2c_f32_bf16 against shift + subtract
__global__ void k_dot2_probe(const float* y, unsigned* bad, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float y0 = y[2 * i], y1 = y[2 * i + 1];
    unsigned hi;
    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(y0), "v"(y1));
    const float ref0 = y0 - __uint_as_float(hi << 16), ref1 = y1 - __uint_as_float(hi & 0xffff0000u);
    float d0 = y0, d1 = y1;
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(d0) : "v"(hi), "v"(0x0000bf80u));
    asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(d1) : "v"(hi), "v"(0xbf800000u));
    if (__float_as_uint(d0) != __float_as_uint(ref0) || __float_as_uint(d1) != __float_as_uint(ref1)) atomicAdd(bad, 1u);
}

// ---------------------------------------------------------------------------------------------------------------------
// The 8-wave kernel's item loop, feature by feature: an item = 3 MFMAs (hi.hi, hi.lo, lo.hi) of one output tile whose A fragments
// come from a 3-slot ring read two items ahead (2 x ds_read_b128 per item), V12 epilogue-mix VALU and G12 LDS-DMA pieces per 4
// items.  DEP: the three MFMAs of an item share one accumulator (as mfma3 does) instead of three.  REAL: the MFMAs really
// consume the ring (s_waitcnt lgkmcnt(4) in front of every item).  NOPS: the "s_nop 0; s_nop 1" hipcc puts in front of an item.
struct RingU { u32x4 h[3], l[3]; };

template <int V12, int G12, bool DEP, bool REAL, bool NOPS, int IT>
struct V1Body {
    static __device__ __forceinline__ void run(f32x4 (&acc)[12], const u32x4& a, const u32x4& b, const u32x4& b2, Fill& s, RingU& r) {
        if constexpr (IT < 4) {
            constexpr int S = IT % 3, S2 = (IT + 2) % 3;
            asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                         : "=&v"(r.h[S2]), "=&v"(r.l[S2]) : "v"(s.lds_addr), "i"(IT * 2048), "i"(IT * 2048 + 1024));
            if constexpr (REAL) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(r.h[S]), "+v"(r.l[S]));
            if constexpr (NOPS) asm volatile("s_nop 0\n\ts_nop 1");
            constexpr int A0 = DEP ? IT * 3 : IT * 3, A1 = DEP ? IT * 3 : IT * 3 + 1, A2 = DEP ? IT * 3 : IT * 3 + 2;
            if constexpr (REAL) {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[A0]) : "v"(r.h[S]), "v"(b));
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[A1]) : "v"(r.h[S]), "v"(b2));
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[A2]) : "v"(r.l[S]), "v"(b));
            } else {
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[A0]) : "v"(a), "v"(b));
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[A1]) : "v"(a), "v"(b2));
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[A2]) : "v"(a), "v"(b));
            }
            mix_ops<0, upto(3 * IT, V12), upto(3 * IT + 3, V12) - upto(3 * IT, V12)>(s);
            dma_ops<upto(3 * IT, G12), upto(3 * IT + 3, G12) - upto(3 * IT, G12)>(s);
            V1Body<V12, G12, DEP, REAL, NOPS, IT + 1>::run(acc, a, b, b2, s, r);
        }
    }
};

// the same item with its fillers INSIDE the triple: MFMA, reads + a third of the VALU, MFMA, a third, MFMA, a third + DMA
template <int V12, int G12, int IT>
struct V1Split {
    static __device__ __forceinline__ void run(f32x4 (&acc)[12], const u32x4& b, const u32x4& b2, Fill& s, RingU& r) {
        if constexpr (IT < 4) {
            constexpr int S = IT % 3, S2 = (IT + 2) % 3, A = IT * 3;
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(r.h[S]), "+v"(r.l[S]));
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[A]) : "v"(r.h[S]), "v"(b));
            asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                         : "=&v"(r.h[S2]), "=&v"(r.l[S2]) : "v"(s.lds_addr), "i"(IT * 2048), "i"(IT * 2048 + 1024));
            mix_ops<0, upto(3 * IT, V12), upto(3 * IT + 1, V12) - upto(3 * IT, V12)>(s);
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[A]) : "v"(r.h[S]), "v"(b2));
            mix_ops<0, upto(3 * IT + 1, V12), upto(3 * IT + 2, V12) - upto(3 * IT + 1, V12)>(s);
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[A]) : "v"(r.l[S]), "v"(b));
            mix_ops<0, upto(3 * IT + 2, V12), upto(3 * IT + 3, V12) - upto(3 * IT + 2, V12)>(s);
            dma_ops<upto(3 * IT, G12), upto(3 * IT + 3, G12) - upto(3 * IT, G12)>(s);
            V1Split<V12, G12, IT + 1>::run(acc, b, b2, s, r);
        }
    }
};

template <int V12, int G12, bool DEP, bool REAL, bool NOPS, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 1) k_v1like(float* out, int iters, const char* gbuf) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 1.0f;
    __syncthreads();
    Fill s;
#pragma unroll
    for (int i = 0; i < 16; ++i) s.f[i] = 0.001f * (float)(threadIdx.x + i);
    s.lds_addr = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;
    s.voff = (threadIdx.x & 63) * 16;
    s.gsrc = gbuf;
    const unsigned wave_skew = (threadIdx.x >> 6) * 4096;
    asm volatile("s_mov_b32 m0, %0" ::"s"((unsigned)(size_t)lds + 16384) : "memory");
    u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a, b2 = a;
    RingU r;
#pragma unroll
    for (int i = 0; i < 3; ++i) { r.h[i] = a; r.l[i] = a; }
    f32x4 acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if constexpr (!DEP && REAL && NOPS) V1Split<V12, G12, 0>::run(acc, b, b2, s, r);      // (this flag combination selects the split form)
        else V1Body<V12, G12, DEP, REAL, NOPS, 0>::run(acc, a, b, b2, s, r);
        if constexpr (G12 > 0) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G12) : "memory");
            s.gsrc = gbuf + __builtin_amdgcn_readfirstlane(((unsigned)(it + 1) * (G12 * 1024u) * 8u + wave_skew) & (2u * 1024 * 1024 - 1));
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 12; ++i) q += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 16; ++i) q += s.f[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) q += __uint_as_float(r.h[i][0] ^ r.l[i][1]);
    if (q == 123.456f) out[0] = q;
}

struct VRow { int v12, g12, dep, real, nops, waves; double ns; };
static std::vector<VRow> vrows;
template <int V12, int G12, bool DEP, bool REAL, bool NOPS, int WAVES>
static void vrun() {
    const int iters = 3000, blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_v1like<V12, G12, DEP, REAL, NOPS, WAVES><<<blocks, 64 * WAVES>>>(d_out, 50, d_buf);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        k_v1like<V12, G12, DEP, REAL, NOPS, WAVES><<<blocks, 64 * WAVES>>>(d_out, iters, d_buf);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    vrows.push_back({V12, G12, DEP, REAL, NOPS, WAVES / 4, best * 1e6 / ((double)iters * 12 * (WAVES / 4))});
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
}

int main() {
    (void)hipMalloc(&d_out, 64);
    (void)hipMalloc(&d_buf, 4u * 1024 * 1024);
    (void)hipMemset(d_buf, 0, 4u * 1024 * 1024);
    all_kinds<0, 4>();
    all_kinds<0, 8>();
    all_kinds<1, 4>();
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate * 1e-6;
    printf("# device %s, %d CUs, clockRate %.3f GHz (nominal: cycles below = ns * this)\n", p.name, p.multiProcessorCount, ghz);
    printf("# shape 0 = v_mfma_f32_16x16x32_bf16 (acc in VGPR), shape 1 = v_mfma_f32_32x32x16_bf16 (acc in AGPR)\n");
    printf("%-6s %-6s %-22s %3s %12s %10s %14s\n", "shape", "w/simd", "filler", "nv", "ns/mfma", "cyc/mfma", "cyc/filler(+)");
    double base[2][3] = {{0, 0, 0}, {0, 0, 0}};
    for (const Row& r : rows)
        if (r.kind == K_NONE) base[r.shape][r.waves] = r.ns_per_mfma_simd;
    for (const Row& r : rows) {
        const double cyc = r.ns_per_mfma_simd * ghz, b = base[r.shape][r.waves] * ghz;
        printf("%-6d %-6d %-22s %3d %12.3f %10.2f %14.2f\n", r.shape, r.waves, kind_name[r.kind], r.nv, r.ns_per_mfma_simd, cyc,
               r.nv ? (cyc - b) / r.nv : 0.0);
    }
    // ---- combinations (per 12 MFMAs): the v1 kernel's mix (16x16x32, 2 waves / SIMD) and the one-wave 32x32x16 kernel's
    crun<0, 0, 0, 0, 0, 8>();
    crun<0, 0, 22, 0, 0, 8>();  crun<0, 0, 37, 0, 0, 8>();  crun<0, 0, 0, 8, 0, 8>();  crun<0, 0, 0, 0, 1, 8>();
    crun<0, 0, 22, 8, 0, 8>();  crun<0, 0, 22, 8, 1, 8>();  crun<0, 0, 37, 8, 1, 8>();
    crun<0, 1, 14, 8, 1, 8>();  crun<0, 1, 29, 8, 1, 8>();
    crun<0, 0, 22, 8, 1, 4>();  crun<0, 0, 37, 8, 1, 4>();  crun<0, 1, 14, 8, 1, 4>();
    crun<1, 0, 0, 0, 0, 4>();
    crun<1, 0, 44, 0, 0, 4>();  crun<1, 0, 0, 8, 0, 4>();   crun<1, 0, 0, 0, 2, 4>();
    crun<1, 0, 44, 8, 0, 4>();  crun<1, 0, 44, 8, 2, 4>();  crun<1, 0, 60, 8, 2, 4>();
    crun<1, 1, 28, 8, 2, 4>();  crun<1, 1, 44, 8, 2, 4>();
    // the weight stream in other forms (pat 2: through registers + ds_write_b128; pat 4: LDS-DMA with no LDS read in flight)
    crun<1, 2, 44, 8, 2, 4>();  crun<1, 2, 0, 8, 2, 4>();   crun<1, 2, 0, 0, 2, 4>();
    crun<1, 4, 44, 8, 2, 4>();  crun<1, 4, 0, 8, 2, 4>();
    crun<1, 0, 0, 8, 2, 4>();   crun<1, 0, 44, 0, 2, 4>();
    crun<0, 2, 22, 8, 1, 8>();  crun<0, 4, 22, 8, 1, 8>();
    // placement of the LDS-DMA piece inside its gap (pat 8: right behind the MFMA; pat 16: between the reads and the VALU)
    crun<1, 8, 44, 8, 2, 4>();  crun<1, 16, 44, 8, 2, 4>(); crun<1, 8, 44, 0, 2, 4>();
    crun<0, 8, 22, 8, 1, 8>();
    crun<1, 32, 44, 8, 2, 4>(); crun<1, 64, 44, 8, 2, 4>(); crun<1, 16, 44, 12, 2, 4>(); crun<1, 16, 60, 8, 2, 4>(); crun<1, 16, 36, 8, 2, 4>();
    printf("\n# combinations, per 12 MFMAs: V12 epilogue-mix VALU (pat & 1: 0 = today's 22 per value pair, 1 = leaner 14), D12 ds_read_b128, G12 weight pieces (pat >> 1: 0 LDS-DMA, 1 global load + ds_write_b128, 2 LDS-DMA behind lgkmcnt(0))\n");
    printf("%-6s %-6s %-4s %4s %4s %4s %12s %10s\n", "shape", "w/simd", "pat", "V12", "D12", "G12", "ns/mfma", "cyc/mfma");
    for (const CRow& r : crows)
        printf("%-6d %-6d %-4d %4d %4d %4d %12.3f %10.2f\n", r.shape, r.waves, r.pat, r.v12, r.d12, r.g12, r.ns, r.ns * ghz);
    vrun<0, 0, false, false, false, 8>();
    vrun<23, 1, false, false, false, 8>();
    vrun<23, 1, true, false, false, 8>();
    vrun<23, 1, false, true, false, 8>();
    vrun<23, 1, true, true, false, 8>();
    vrun<23, 1, true, true, true, 8>();
    vrun<23, 0, true, true, true, 8>();
    vrun<0, 0, true, true, true, 8>();
    vrun<0, 0, true, true, false, 8>();
    vrun<0, 0, true, false, false, 8>();
    vrun<23, 1, false, true, true, 8>();      // split form (fillers inside the triple)
    vrun<23, 0, false, true, true, 8>();
    vrun<0, 0, false, true, true, 8>();
    vrun<23, 1, true, true, true, 4>();
    vrun<0, 0, true, false, false, 4>();
    printf("\n# the 8-wave kernel's item loop (16x16x32; item = 3 MFMAs + 2 ds_read_b128; V12 / G12 per 4 items)\n");
    printf("# (dep 0, real 1, nops 1) = the split form: reads and VALU between the MFMAs of an item\n");
    printf("%-6s %4s %4s %4s %5s %5s %12s %10s\n", "w/simd", "V12", "G12", "dep", "real", "nops", "ns/mfma", "cyc/mfma");
    for (const VRow& r : vrows)
        printf("%-6d %4d %4d %4d %5d %5d %12.3f %10.2f\n", r.waves, r.v12, r.g12, r.dep, r.real, r.nops, r.ns, r.ns * ghz);
    {
        const int n = 1 << 20;
        std::vector<float> h(n);
        unsigned seed = 12345u;
        for (int i = 0; i < n; ++i) {
            seed = seed * 1664525u + 1013904223u;
            const float u = (float)(seed >> 8) * (1.0f / 16777216.0f);
            h[i] = (i & 1 ? -1.f : 1.f) * u * ((i % 7) == 0 ? 1e-6f : ((i % 5) == 0 ? 300.f : 1.f));
        }
        float* dy; unsigned* dbad; unsigned bad = 0;
        (void)hipMalloc(&dy, n * 4); (void)hipMalloc(&dbad, 4);
        (void)hipMemcpy(dy, h.data(), n * 4, hipMemcpyHostToDevice);
        (void)hipMemset(dbad, 0, 4);
        k_dot2_probe<<<n / 2 / 256, 256>>>(dy, dbad, n);
        (void)hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost);
        printf("\n# v_dot2c_f32_bf16 as (y - float(hi)): %u of %d pairs differ from shift + subtract\n", bad, n / 2);
    }
    return 0;
}

// mlp_chain_bf16.hip - the chained-MLP kernels on the bf16 matrix cores with SPLIT operands
// ("bf16x3"): every fp32 operand is split x = hi + lo (hi = bf16_rne(x), lo = bf16_rne(x - hi)) and
// a.b is evaluated as a_hi.b_hi + a_hi.b_lo + a_lo.b_hi with fp32 accumulation - three
// v_mfma_f32_16x16x32_bf16 per 32-deep k-step.  The dropped a_lo.b_lo term and the residual of the
// two-term split are each ~2^-16 relative per product (measured on the SDF net: 2e-5 max / 3e-6
// mean absolute sdf error vs fp64, against 2.5e-6 for plain fp32), i.e. ~17 significand bits where
// TF32 (what the reference's published RTX 3090 number used) has 11.  The bf16 MFMA rate is 16x the
// fp32 MFMA rate, so the split form is 16/3 = 5.3x the fp32-exact kernels of mlp_chain.hip.
//
// Structure = mlp_chain.hip's (8 waves x 16 columns per workgroup, two waves per SIMD so that one wave's
// epilogue / barrier / LDS-DMA wait hides under the other's MFMAs; k-outer; activations never leave
// registers), with the 16x16x32 layouts:
//   * lane (g = lane>>4, j = lane&15).  A: lane (g,i) holds 8 k-slots of row i; B: lane (g,j) holds 8
//     k-slots of column j; C: reg r of tile T <-> row 16T + 4g + r.  Which feature a k-slot means is our
//     choice (the MFMA only pairs slot (g,e) of A with slot (g,e) of B): the 4+4 registers of output tiles
//     2u and 2u+1 become, after activation + split + v_cvt_pk_bf16_f32, "unit" u (32 slots) of the next
//     layer's B operand directly in registers; packing.py (bf16 plans) applies the matching permutation.
//   * the 16 accumulator tiles of a layer (64 VGPRs) stay live; a weight chunk = 2 k-steps x 16 tiles x
//     (hi, lo) fragments = 64 KiB, LDS-DMA double buffered, one barrier per chunk (96 MFMAs per wave).
//   * A fragments are read from LDS two items ahead with inline-asm ds_read_b128 and counted lgkmcnt waits
//     (hipcc sinks plain reads back to their use and waits lgkmcnt(0) every 3 MFMAs).
// An earlier variant (32x32x16, 4 waves x 32 columns, one wave per SIMD; git history) ran at 42 % MFMA
// utilisation: an ablation attributed 24 % of its time to exposed LDS-DMA waits and 24 % to the epilogue,
// neither of which a single wave per SIMD can hide.
#include "mlp_common.h"

namespace nerfart {
namespace b16 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int WG_THREADS = 512;
constexpr int WAVES = 8;
constexpr int XU_MAX = 10;                          // input units (32 feature slots each)
constexpr int TS_FLOATS = 512;                      // one item = (k-step, output tile): (hi, lo) x 64 lanes x 16 B = 2 KiB
constexpr int KS_FLOATS = 16 * TS_FLOATS;           // one k-step of a chunk: 16 output tiles = 32 KiB
constexpr int CHUNK_KS = 2;                         // k-steps per chunk
constexpr int CHUNK_FLOATS_MAX = CHUNK_KS * KS_FLOATS;   // 64 KiB
constexpr int AUX_FLOATS_MAX = 2560;
constexpr int LDS_FLOATS = 2 * CHUNK_FLOATS_MAX + AUX_FLOATS_MAX + TAB_INTS;   // 141,824 B
using Pipe = PipeT<WAVES, CHUNK_FLOATS_MAX>;

struct Act { u32x4 h[XU_MAX]; u32x4 l[XU_MAX]; };   // packed bf16 pairs: hi and lo terms of 8 slots per unit
struct Acc { f32x4 t[16]; };

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    f32x2_ v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));       // v_cvt_pk_bf16_f32 (RNE)
}
__device__ __forceinline__ void split2(float y0, float y1, unsigned& hi, unsigned& lo) {
    hi = pack_bf16(y0, y1);
    const float h0 = __uint_as_float(hi << 16), h1 = __uint_as_float(hi & 0xffff0000u);
    lo = pack_bf16(y0 - h0, y1 - h1);
}
__device__ __forceinline__ f32x4 mfma3(const u32x4 ah, const u32x4 al, const u32x4 bh, const u32x4 bl, f32x4 acc) {
#ifdef NERFART_ABLATE_MFMA      // timing experiments only (tools/ablate_bf16.py): keep the operands live, skip the matrix work
    asm volatile("" :: "v"(ah), "v"(al), "v"(bh), "v"(bl));
    return acc;
#endif
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
    return acc;
}

// ---------------------------------------------------------------------------------------
// The MFMAs of one weight chunk: NKS k-steps x 16 output tiles, 3 MFMAs each.  LDS returns in order: with
// items it, it+1, it+2 outstanding (2 reads each) item it has landed at lgkmcnt(4)
// (cdna_hip_programming.md 5.7, form ii: the wait statement names the destinations "+v").
// ---------------------------------------------------------------------------------------
struct Ring { u32x4 h0, l0, h1, l1, h2, l2; };

template <int OFF>
__device__ __forceinline__ void lds_read_pair(u32x4& fh, u32x4& fl, unsigned addr) {
#ifdef NERFART_ABLATE_LDSREAD    // timing experiments only: fragments are whatever the registers held
    asm volatile("; no read %0 %1 %2" : "=&v"(fh), "=&v"(fl) : "v"(addr));
    return;
#endif
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                 : "=&v"(fh), "=&v"(fl) : "v"(addr), "i"(OFF), "i"(OFF + 1024));
}
template <int CNT>
__device__ __forceinline__ void lds_wait_pair(u32x4& fh, u32x4& fl) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fh), "+v"(fl) : "i"(CNT));
}

template <int IT, int N>
struct ChunkSteps {
    static __device__ __forceinline__ void run(Acc& A, const Act& X, int ks0, unsigned addr, Ring& r) {
        if constexpr (IT < N) {
            constexpr int S = IT % 3, S2 = (IT + 2) % 3;
            if constexpr (IT + 2 < N) {
                if constexpr (S2 == 0) lds_read_pair<(IT + 2) * 2048>(r.h0, r.l0, addr);
                else if constexpr (S2 == 1) lds_read_pair<(IT + 2) * 2048>(r.h1, r.l1, addr);
                else lds_read_pair<(IT + 2) * 2048>(r.h2, r.l2, addr);
            }
            constexpr int PENDING = (IT + 2 < N) ? 4 : ((IT + 1 < N) ? 2 : 0);
            constexpr int kk = IT >> 4, T = IT & 15;
            if constexpr (S == 0) { lds_wait_pair<PENDING>(r.h0, r.l0); A.t[T] = mfma3(r.h0, r.l0, X.h[ks0 + kk], X.l[ks0 + kk], A.t[T]); }
            else if constexpr (S == 1) { lds_wait_pair<PENDING>(r.h1, r.l1); A.t[T] = mfma3(r.h1, r.l1, X.h[ks0 + kk], X.l[ks0 + kk], A.t[T]); }
            else { lds_wait_pair<PENDING>(r.h2, r.l2); A.t[T] = mfma3(r.h2, r.l2, X.h[ks0 + kk], X.l[ks0 + kk], A.t[T]); }
            ChunkSteps<IT + 1, N>::run(A, X, ks0, addr, r);
        }
    }
};

template <int NKS>
__device__ __forceinline__ void chunk_mma(Acc& A, const Act& X, int ks0, const float* w) {
    constexpr int N = NKS * 16;
    const unsigned addr = (unsigned)(size_t)w;          // LDS byte address of this lane's 16 bytes of item 0
    // everything the compiler itself has in flight on the LDS queue must be drained first: the counted waits
    // below assume only these reads are outstanding
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    Ring r;
    lds_read_pair<0>(r.h0, r.l0, addr);
    lds_read_pair<2048>(r.h1, r.l1, addr);
    ChunkSteps<0, N>::run(A, X, ks0, addr, r);
}

// Run-time (wave-uniform) description of what a layer's epilogue does (branch free: run-time branches in the
// epilogue split it into small basic blocks and stop hipcc from overlapping anything).
struct Epi {
    const float* bias;     // LDS, natural feature order
    float floor;           // ReLU family: 0 (ReLU) or -inf (no activation)
    const float* rows;     // LDS: NROWS x 256 weights of the final linear layer (LAST bodies)
    float* h7;             // per-lane destination of the fp32 activations (tangent kernel, LAST body) or null
};

// activation of one accumulator tile (features 16T + 4g + r)
template <bool SOFTPLUS, bool TANGENT>
__device__ __forceinline__ f32x4 activate(const f32x4 a, const Epi& e, bool is_val) {
    f32x4 y;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#ifdef NERFART_ABLATE_EPI       // timing experiments only: no activation arithmetic
        y[r] = a[r];
        continue;
#endif
        if (SOFTPLUS) {
            if (TANGENT) {
                float v, d;
                softplus100_vd(a[r], v, d);       // value lanes carry the bias from the accumulator init
                d = quad_bcast0(d);
                y[r] = is_val ? v : d * a[r];
            } else {
                y[r] = softplus100(a[r]);
            }
        } else {
            y[r] = fmaxf(a[r], e.floor);
        }
    }
    return y;
}

// Epilogue of unit U = output tiles 2U and 2U+1: either the next layer's input unit (slot e < 4: tile 2U reg e,
// e >= 4: tile 2U+1 reg e-4) or, in LAST bodies, the final rows' dot products (+ the fp32 activations h7).
template <int U, bool SOFTPLUS, bool TANGENT, bool LAST, int NROWS>
__device__ __forceinline__ void epi_unit(const Acc& A, const Epi& e, Act& X, float (&dot)[NROWS], int g, bool is_val) {
    const f32x4 y0 = activate<SOFTPLUS, TANGENT>(A.t[2 * U], e, is_val);
    const f32x4 y1 = activate<SOFTPLUS, TANGENT>(A.t[2 * U + 1], e, is_val);
    if (LAST) {
#pragma unroll
        for (int n = 0; n < NROWS; ++n) {
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(e.rows + n * 256 + 32 * U + 4 * g);
            const f32x4 w1 = *reinterpret_cast<const f32x4*>(e.rows + n * 256 + 32 * U + 16 + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) { dot[n] = fmaf(y0[r], w0[r], dot[n]); dot[n] = fmaf(y1[r], w1[r], dot[n]); }
        }
        if (TANGENT) {
            if (e.h7 != nullptr && is_val) {
                *reinterpret_cast<f32x4*>(e.h7 + 32 * U + 4 * g) = y0;
                *reinterpret_cast<f32x4*>(e.h7 + 32 * U + 16 + 4 * g) = y1;
            }
        }
    } else {
#ifdef NERFART_ABLATE_EPI
        X.h[U] = u32x4{__float_as_uint(y0[0]), __float_as_uint(y0[1]), __float_as_uint(y0[2]), __float_as_uint(y0[3])};
        X.l[U] = u32x4{__float_as_uint(y1[0]), __float_as_uint(y1[1]), __float_as_uint(y1[2]), __float_as_uint(y1[3])};
        return;
#endif
        u32x4 hi, lo;
        unsigned a, b;
        split2(y0[0], y0[1], a, b); hi[0] = a; lo[0] = b;
        split2(y0[2], y0[3], a, b); hi[1] = a; lo[1] = b;
        split2(y1[0], y1[1], a, b); hi[2] = a; lo[2] = b;
        split2(y1[2], y1[3], a, b); hi[3] = a; lo[3] = b;
        X.h[U] = hi;
        X.l[U] = lo;
    }
}

// A whole dense layer, in place on X.  Chunk sequence (must match packing.py, bf16 plans): ceil(NU_BASE/2)
// chunks of the base units, then (if nextra) one chunk with the nextra (<= 2) extra units.
template <int NU_BASE, int NU_EXTRA_MAX, bool SOFTPLUS, bool TANGENT, bool LAST, int NROWS>
__device__ __forceinline__ void run_layer(Act& X, Pipe& p, const Epi& e, float (&dot)[NROWS], int nextra) {
    const int lane = lane_id();
    const int g = lane >> 4;
    const bool is_val = !TANGENT || ((lane & 3) == 0);
    Acc A;
    // start at the bias (value columns; derivative columns start at 0): reg r of tile T is feature 16T + 4g + r
#pragma unroll
    for (int T = 0; T < 16; ++T) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(e.bias + 16 * T + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) A.t[T][r] = is_val ? b[r] : 0.f;
    }
#pragma unroll
    for (int c = 0; c < (NU_BASE + CHUNK_KS - 1) / CHUNK_KS; ++c) {
        const float* w = pipe_acquire(p) + lane * 4;
        constexpr int REM = NU_BASE % CHUNK_KS;
        if (REM != 0 && c == NU_BASE / CHUNK_KS) chunk_mma<1>(A, X, c * CHUNK_KS, w);
        else chunk_mma<CHUNK_KS>(A, X, c * CHUNK_KS, w);
    }
    if (NU_EXTRA_MAX > 0) {
        if (nextra > 0) {
            const float* w = pipe_acquire(p) + lane * 4;
            if (NU_EXTRA_MAX == 1 || nextra == 1) chunk_mma<1>(A, X, NU_BASE, w);
            else chunk_mma<2>(A, X, NU_BASE, w);
        }
    }
    epi_unit<0, SOFTPLUS, TANGENT, LAST, NROWS>(A, e, X, dot, g, is_val);
    epi_unit<1, SOFTPLUS, TANGENT, LAST, NROWS>(A, e, X, dot, g, is_val);
    epi_unit<2, SOFTPLUS, TANGENT, LAST, NROWS>(A, e, X, dot, g, is_val);
    epi_unit<3, SOFTPLUS, TANGENT, LAST, NROWS>(A, e, X, dot, g, is_val);
    epi_unit<4, SOFTPLUS, TANGENT, LAST, NROWS>(A, e, X, dot, g, is_val);
    epi_unit<5, SOFTPLUS, TANGENT, LAST, NROWS>(A, e, X, dot, g, is_val);
    epi_unit<6, SOFTPLUS, TANGENT, LAST, NROWS>(A, e, X, dot, g, is_val);
    epi_unit<7, SOFTPLUS, TANGENT, LAST, NROWS>(A, e, X, dot, g, is_val);
}

// ---------------------------------------------------------------------------------------
// Positional encoding of the SDF net in unit order: 2 units x (4 groups x 8 slots) = 64 slots.  Lane group
// g < 3 owns coordinate g: local index m = 8q + e: m = 0 raw, m = 1 + 2k sin(2^k x_g), m = 2 + 2k
// cos(2^k x_g) for k < 6, m = 13..15 pad; lane group 3 is padding (reference Embedder, models/base.py:38-64;
// slot map packing.unit_feature_enc).  dq < 0: values; dq = 0..2: derivative w.r.t. coordinate dq.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void encode_units(float x, float y, float z, int g, int dq, Act& X, int u0) {
    const float cg = (g == 0) ? x : ((g == 1) ? y : z);
    const bool live = g < 3;
    const bool own = (dq == g);
    float m[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) m[k] = 0.f;
    m[0] = (dq < 0) ? cg : (own ? 1.f : 0.f);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float f = (float)(1 << k);
        float s, c;
        sincosf(cg * f, &s, &c);
        m[1 + 2 * k] = (dq < 0) ? s : (own ? c * f : 0.f);
        m[2 + 2 * k] = (dq < 0) ? c : (own ? -(s * f) : 0.f);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        u32x4 hi, lo;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const float a = live ? m[8 * q + 2 * pr] : 0.f, b = live ? m[8 * q + 2 * pr + 1] : 0.f;
            unsigned sh, sl;
            split2(a, b, sh, sl);
            hi[pr] = sh; lo[pr] = sl;
        }
        X.h[u0 + q] = hi;
        X.l[u0 + q] = lo;
    }
}

constexpr int SURF_AUX_ROW = 2048;
constexpr int SURF_AUX_B8 = 2304;
constexpr int SURF_AUX_FLOATS = 2308;
constexpr int RAD_AUX_ROWS = 1280;
constexpr int RAD_AUX_BF = 2048;
constexpr int RAD_AUX_FLOATS = 2052;

__device__ __forceinline__ void load_aux(float* aux_lds, const float* blob, const int* hdr, int nfloats) {
    const float* src = blob + hdr[4];
    for (int i = threadIdx.x; i < nfloats; i += WG_THREADS) aux_lds[i] = src[i];
    int* tab = reinterpret_cast<int*>(aux_lds + AUX_FLOATS_MAX);
    if (threadIdx.x < TAB_INTS) tab[threadIdx.x] = hdr[NERFART_HDR_OFFS + threadIdx.x];
    __syncthreads();
}

// The 8 hidden layers of the SDF net, in place on X; the last one accumulates dot[0] = row0 . h7 (and stores
// h7 if asked).  Bodies: layer 0 (2 input units), layers 1..6 (one body in a run-time loop), layer 7 (LAST).
template <bool TANGENT>
__device__ __forceinline__ float surface_chain(Act& X, float px, float py, float pz, int g, int dq, Pipe& p,
                                               const float* aux, float* h7_lane) {
    float dot[1] = {0.f};
    Epi e{aux, 0.f, aux + SURF_AUX_ROW, h7_lane};
    encode_units(px, py, pz, g, dq, X, 0);
    run_layer<2, 0, true, TANGENT, false, 1>(X, p, e, dot, 0);
#pragma nounroll
    for (int L = 1; L < 7; ++L) {
        // skip: cat[h(217 -> 7 units), enc(2 units)] / sqrt(2) - the 1/sqrt(2) is folded into layer 4's weights
        if (L == 4) encode_units(px, py, pz, g, dq, X, 7);
        e.bias = aux + L * 256;
        run_layer<8, 1, true, TANGENT, false, 1>(X, p, e, dot, (L == 4) ? 1 : 0);
    }
    e.bias = aux + 7 * 256;
    run_layer<8, 0, true, TANGENT, true, 1>(X, p, e, dot, 0);
    return sum_over_groups(dot[0]);        // the 4 lane groups of a column hold complementary feature sets
}

// =======================================================================================
// K2 (split bf16): sdf only, 128 points per workgroup tile, 16 per wave.
// =======================================================================================
__global__ void __launch_bounds__(WG_THREADS, 2)
k_sdf_only_bf16(const float* __restrict__ blob, PointSrc src, float R_bg, float* __restrict__ sdf_out, int out_stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int* hdr = reinterpret_cast<const int*>(blob);
    float* aux = smem + 2 * CHUNK_FLOATS_MAX;
    const int lane = lane_id(), g = lane >> 4, j = lane & 15, wv = wave_id();
    load_aux(aux, blob, hdr, SURF_AUX_FLOATS);
    const unsigned ntiles = (src.M + 127u) / 128u;
    if (blockIdx.x >= ntiles) return;
    Pipe p{blob, reinterpret_cast<const int*>(aux + AUX_FLOATS_MAX), smem, hdr[2], 0, 0, 0, 0, false};
    p.wrap = (blockIdx.x + gridDim.x) < ntiles;
    pipe_start(p);
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        p.wrap = (tile + gridDim.x) < ntiles;
        const unsigned m = tile * 128u + wv * 16 + j;
        const Pt pt = fetch_point(src, m, false);
        Act X;
        float sdf = surface_chain<false>(X, pt.x, pt.y, pt.z, g, -1, p, aux, nullptr) + aux[SURF_AUX_B8];
        if (R_bg > 0.f) sdf = fminf(sdf, R_bg - sqrtf(pt.x * pt.x + pt.y * pt.y + pt.z * pt.z));
        if (g == 0 && m < src.M) {
            if (src.pts) sdf_out[m] = sdf;
            else {
                const unsigned slot = m / (unsigned)src.n_per_ray;
                sdf_out[(size_t)slot * out_stride + (m - slot * (unsigned)src.n_per_ray)] = sdf;
            }
        }
    }
}

// =======================================================================================
// K3a (split bf16): sdf + nabla + h7, forward mode; 32 points per workgroup tile (4 per wave, quads).
// =======================================================================================
__global__ void __launch_bounds__(WG_THREADS, 2)
k_sdf_nabla_bf16(const float* __restrict__ blob, PointSrc src, float R_bg, float* __restrict__ sdf_out,
                 float* __restrict__ nabla_out, float* __restrict__ h7_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int* hdr = reinterpret_cast<const int*>(blob);
    float* aux = smem + 2 * CHUNK_FLOATS_MAX;
    const int lane = lane_id(), g = lane >> 4, j = lane & 15, wv = wave_id();
    const int cq = j & 3;
    load_aux(aux, blob, hdr, SURF_AUX_FLOATS);
    const unsigned ntiles = (src.M + 31u) / 32u;
    if (blockIdx.x >= ntiles) return;
    Pipe p{blob, reinterpret_cast<const int*>(aux + AUX_FLOATS_MAX), smem, hdr[2], 0, 0, 0, 0, false};
    p.wrap = (blockIdx.x + gridDim.x) < ntiles;
    pipe_start(p);
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        p.wrap = (tile + gridDim.x) < ntiles;
        const unsigned m = tile * 32u + wv * 4 + (j >> 2);
        const Pt pt = fetch_point(src, m, false);
        Act X;
        float* h7_lane = (h7_out != nullptr && m < src.M) ? h7_out + (size_t)m * 256 : nullptr;
        const float v = surface_chain<true>(X, pt.x, pt.y, pt.z, g, cq - 1, p, aux, h7_lane);
        if (m < src.M && g == 0) {
            if (cq == 0) {
                float sdf = v + aux[SURF_AUX_B8];
                if (R_bg > 0.f) {
                    const float d_bg = R_bg - sqrtf(pt.x * pt.x + pt.y * pt.y + pt.z * pt.z);
                    sdf = (d_bg < sdf) ? d_bg : sdf;
                }
                sdf_out[m] = sdf;
            } else {
                nabla_out[(size_t)m * 3 + (cq - 1)] = v;
            }
        }
    }
}

// =======================================================================================
// K3b (split bf16): radiance net.  VE extra units: 1 (VolSDF, 9 extras) or 2 (NeuS, 33 extras).
// =======================================================================================
template <int VE>
__device__ __forceinline__ void radiance_extras(const Pt& pt, float nx, float ny, float nz, int g, Act& X) {
    constexpr int NE = (VE == 1) ? 9 : 33;
    float ex[VE * 32];
#pragma unroll
    for (int k = 0; k < VE * 32; ++k) ex[k] = 0.f;
    ex[0] = pt.x; ex[1] = pt.y; ex[2] = pt.z;
    const float v[3] = {pt.vx, pt.vy, pt.vz};
#pragma unroll
    for (int c = 0; c < 3; ++c) ex[3 + c] = v[c];
    if (VE == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float s, co;
                sincosf(v[c] * (float)(1 << k), &s, &co);
                ex[6 + 6 * k + c] = s;
                ex[6 + 6 * k + 3 + c] = co;
            }
    }
    ex[NE - 3] = nx; ex[NE - 2] = ny; ex[NE - 1] = nz;
    // slot (q, g, e) <-> extra index 32q + 8g + e
#pragma unroll
    for (int q = 0; q < VE; ++q) {
        u32x4 hi, lo;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const int o = 32 * q + 2 * pr;
            const float a = (g == 0) ? ex[o] : ((g == 1) ? ex[o + 8] : ((g == 2) ? ex[o + 16] : ex[o + 24]));
            const float b = (g == 0) ? ex[o + 1] : ((g == 1) ? ex[o + 9] : ((g == 2) ? ex[o + 17] : ex[o + 25]));
            unsigned sh, sl;
            split2(a, b, sh, sl);
            hi[pr] = sh; lo[pr] = sl;
        }
        X.h[8 + q] = hi;
        X.l[8 + q] = lo;
    }
}

template <int VE>
__global__ void __launch_bounds__(WG_THREADS, 2)
k_radiance_bf16(const float* __restrict__ blob, PointSrc src, const float* __restrict__ nabla_in,
                const float* __restrict__ h7_in, float* __restrict__ rgb_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int* hdr = reinterpret_cast<const int*>(blob);
    float* aux = smem + 2 * CHUNK_FLOATS_MAX;
    const int lane = lane_id(), g = lane >> 4, j = lane & 15, wv = wave_id();
    load_aux(aux, blob, hdr, RAD_AUX_FLOATS);
    const unsigned ntiles = (src.M + 127u) / 128u;
    if (blockIdx.x >= ntiles) return;
    Pipe p{blob, reinterpret_cast<const int*>(aux + AUX_FLOATS_MAX), smem, hdr[2], 0, 0, 0, 0, false};
    p.wrap = (blockIdx.x + gridDim.x) < ntiles;
    pipe_start(p);
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        p.wrap = (tile + gridDim.x) < ntiles;
        const unsigned m = tile * 128u + wv * 16 + j;
        const bool valid = m < src.M;
        const Pt pt = fetch_point(src, m, true);
        Act X;
        float nx = 0.f, ny = 0.f, nz = 0.f;
        if (valid) { nx = nabla_in[(size_t)m * 3 + 0]; ny = nabla_in[(size_t)m * 3 + 1]; nz = nabla_in[(size_t)m * 3 + 2]; }
        // h7 -> units: unit u slot e < 4: feature 32u + 4g + e; e >= 4: 32u + 16 + 4g + (e - 4)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            f32x4 lo4 = {0.f, 0.f, 0.f, 0.f}, hi4 = {0.f, 0.f, 0.f, 0.f};
            if (valid) {
                const float* s0 = h7_in + (size_t)m * 256 + 32 * u + 4 * g;
                lo4 = *reinterpret_cast<const f32x4*>(s0);
                hi4 = *reinterpret_cast<const f32x4*>(s0 + 16);
            }
            unsigned sh[4], sl[4];
            split2(lo4[0], lo4[1], sh[0], sl[0]);
            split2(lo4[2], lo4[3], sh[1], sl[1]);
            split2(hi4[0], hi4[1], sh[2], sl[2]);
            split2(hi4[2], hi4[3], sh[3], sl[3]);
            X.h[u] = u32x4{sh[0], sh[1], sh[2], sh[3]};
            X.l[u] = u32x4{sl[0], sl[1], sl[2], sl[3]};
        }
        radiance_extras<VE>(pt, nx, ny, nz, g, X);
        float dot[3] = {0.f, 0.f, 0.f};
        Epi e{aux, -INFINITY, aux + RAD_AUX_ROWS, nullptr};
        // L = 0: geometry feature (no activation); L = 1: [feat | x, v, n] -> 256 ReLU; L = 2, 3: ReLU; L = 4: LAST
#pragma nounroll
        for (int L = 0; L < 4; ++L) {
            e.bias = aux + L * 256;
            e.floor = (L == 0) ? -INFINITY : 0.f;
            run_layer<8, VE, false, false, false, 3>(X, p, e, dot, (L == 1) ? VE : 0);
        }
        e.bias = aux + 4 * 256;
        e.floor = 0.f;
        run_layer<8, 0, false, false, true, 3>(X, p, e, dot, 0);
        float c[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) c[n] = sigmoidf_(sum_over_groups(dot[n]) + aux[RAD_AUX_BF + n]);
        if (valid && g < 3) rgb_out[(size_t)m * 3 + g] = (g == 0) ? c[0] : ((g == 1) ? c[1] : c[2]);
    }
}

template <typename K, typename... Args>
static int launch_chain(int prof_cls, long long units, K kernel, unsigned ntiles, hipStream_t stream, Args... args) {
    const size_t lds = LDS_FLOATS * sizeof(float);
    NERFART_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = ntiles < (unsigned)num_cus() ? ntiles : (unsigned)num_cus();
    void* ph = nullptr;
    if (profile_enabled()) profile_open(prof_cls, units, stream, &ph);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(WG_THREADS), lds, stream, args...);
    profile_close(ph, stream);
    NERFART_HIP(hipGetLastError());
    return 0;
}

}  // namespace b16
}  // namespace nerfart

using namespace nerfart;

// Entry points used by mlp_chain.hip's dispatchers when precision = 1 (split bf16).
namespace nerfart {
int sdf_bf16(const float* blob, const PointSrc& s, float R_bg, float* out, int out_stride, hipStream_t st) {
    return b16::launch_chain(0, (long long)s.M, b16::k_sdf_only_bf16, (s.M + 127u) / 128u, st, blob, s, R_bg, out, out_stride);
}
int sdf_nabla_bf16(const float* blob, const PointSrc& s, float R_bg, float* sdf, float* nabla, float* h7, hipStream_t st) {
    return b16::launch_chain(1, (long long)s.M, b16::k_sdf_nabla_bf16, (s.M + 31u) / 32u, st, blob, s, R_bg, sdf, nabla, h7);
}
int radiance_bf16(const float* blob, int view_tiles, const PointSrc& s, const float* nabla, const float* h7, float* rgb, hipStream_t st) {
    const unsigned nt = (s.M + 127u) / 128u;
    if (view_tiles == 1) return b16::launch_chain(2, (long long)s.M, b16::k_radiance_bf16<1>, nt, st, blob, s, nabla, h7, rgb);
    if (view_tiles == 3) return b16::launch_chain(2, (long long)s.M, b16::k_radiance_bf16<2>, nt, st, blob, s, nabla, h7, rgb);
    set_last_error("radiance_fwd: view_tiles must be 1 (raw view dirs) or 3 (multires_view = 4)");
    return 2;
}
}  // namespace nerfart

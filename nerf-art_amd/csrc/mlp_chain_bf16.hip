// mlp_chain_bf16.hip - the chained-MLP kernels on the bf16 matrix cores with SPLIT operands
// ("bf16x3"): every fp32 operand is split x = hi + lo (hi = bf16_rne(x), lo = bf16_rne(x - hi)) and
// a.b is evaluated as a_hi.b_hi + a_hi.b_lo + a_lo.b_hi with fp32 accumulation - three
// v_mfma_f32_32x32x16_bf16 per 16-deep k-step.  The dropped a_lo.b_lo term and the residual of the
// two-term split are each ~2^-16 relative per product (measured on the SDF net: 2e-5 max / 3e-6
// mean absolute sdf error vs fp64, against 2.5e-6 for plain fp32), i.e. ~17 significand bits where
// TF32 (what the reference's published RTX 3090 number used) has 11.  The bf16 MFMA rate is 16x the
// fp32 MFMA rate, so the split form is 16/3 = 5.3x the fp32-exact kernels of mlp_chain.hip.
//
// Same structure as mlp_chain.hip with the 32x32x16 layouts:
//   * out^T = W . h^T; a wave owns 32 columns; lane (h = lane>>5, j = lane&31);
//     A: lane (h,i) holds W[row i][8 k-slots of half h]; B: lane (h,j) holds 8 k-slots of column j;
//     C: reg r <-> row (r&3) + 8(r>>2) + 4h.  Which feature a k-slot means is our choice (the MFMA only
//     pairs slot (h,e) of A with slot (h,e) of B), so regs 8u..8u+7 of output tile T become, after the
//     activation + split + v_cvt_pk_bf16_f32, "unit" 2T+u of the next layer's B operand directly in
//     registers; the weight packing (packing.py: bf16 plans) applies the matching permutation.
//   * k-outer order: the 8 accumulator tiles of a layer (128 registers, AGPRs) stay live; a weight chunk is
//     up to 4 k-steps x 8 output tiles x (hi, lo) fragments = 64 KiB, LDS-DMA double buffered; input units
//     die as they are consumed and the epilogue writes the next layer's units in place (X + acc = 264
//     registers; a t-outer variant with a deferred epilogue needs X + Y + 2 acc = 300 and spills because
//     vector results must land in the 256 arch VGPRs).
//   * 4 waves x 32 columns = 128 columns per workgroup (one wave per SIMD), persistent grid.
#include "mlp_common.h"

namespace nerfart {
namespace b16 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int WG_THREADS = 256;
constexpr int WAVES = 4;
constexpr int XU_MAX = 19;                          // input units (16 feature slots each)
constexpr int TS_FLOATS = 512;                      // one (k-step, output tile): (hi, lo) x 64 lanes x 16 B = 2 KiB
constexpr int KS_FLOATS = 8 * TS_FLOATS;            // one k-step of a chunk: 8 output tiles = 16 KiB
constexpr int CHUNK_KS = 4;                         // k-steps per chunk
constexpr int CHUNK_FLOATS_MAX = CHUNK_KS * KS_FLOATS;   // 64 KiB
constexpr int AUX_FLOATS_MAX = 2560;
constexpr int LDS_FLOATS = 2 * CHUNK_FLOATS_MAX + AUX_FLOATS_MAX + TAB_INTS;   // 141,824 B
using Pipe = PipeT<WAVES, CHUNK_FLOATS_MAX>;

struct Act { u32x4 h[XU_MAX]; u32x4 l[XU_MAX]; };   // packed bf16 pairs: hi and lo terms of 8 slots per unit

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
__device__ __forceinline__ f32x16 mfma3(const u32x4 ah, const u32x4 al, const u32x4 bh, const u32x4 bl, f32x16 acc) {
#ifdef NERFART_ABLATE_MFMA      // timing experiments only: keep the operands live, skip the matrix work
    asm volatile("" :: "v"(ah), "v"(al), "v"(bh), "v"(bl));
    return acc;
#endif
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh), acc, 0, 0, 0);
    return acc;
}

// Run-time (wave-uniform) description of what a layer's epilogue does (branch free).
struct Epi {
    const float* bias;     // LDS, natural feature order
    float floor;           // ReLU family: 0 (ReLU) or -inf (no activation)
    const float* rows;     // LDS: NROWS x 256 weights of the final linear layer (LAST bodies)
    float* h7;             // per-lane destination of the fp32 activations (tangent kernel, LAST body) or null
};

struct Acc { f32x16 t[8]; };

// The MFMAs of one weight chunk: NKS k-steps x 8 output tiles, 3 MFMAs each.  The A fragments (two
// ds_read_b128 per item) are fetched TWO ITEMS AHEAD through a ring of three register pairs.  hipcc cannot be
// talked into this (it sinks every read back to its use and waits lgkmcnt(0), exposing the LDS latency every 96
// cycles), so the reads are inline asm and the waits are counted by hand (cdna_hip_programming.md 5.7, form
// ii: the wait statement names the destinations "+v", which is what orders the MFMAs behind it).  LDS returns
// in order: with items it, it+1, it+2 outstanding (2 reads each) item it has landed at lgkmcnt(4).
// (The 217-wide layer simply runs its zero-padded 8th tile: a branch here would break the pipeline.)
struct Ring { u32x4 h0, l0, h1, l1, h2, l2; };

template <int OFF>
__device__ __forceinline__ void lds_read_pair(u32x4& fh, u32x4& fl, unsigned addr) {
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
            constexpr int kk = IT >> 3, T = IT & 7;
            if constexpr (S == 0) { lds_wait_pair<PENDING>(r.h0, r.l0); A.t[T] = mfma3(r.h0, r.l0, X.h[ks0 + kk], X.l[ks0 + kk], A.t[T]); }
            else if constexpr (S == 1) { lds_wait_pair<PENDING>(r.h1, r.l1); A.t[T] = mfma3(r.h1, r.l1, X.h[ks0 + kk], X.l[ks0 + kk], A.t[T]); }
            else { lds_wait_pair<PENDING>(r.h2, r.l2); A.t[T] = mfma3(r.h2, r.l2, X.h[ks0 + kk], X.l[ks0 + kk], A.t[T]); }
            ChunkSteps<IT + 1, N>::run(A, X, ks0, addr, r);
        }
    }
};

template <int NKS>
__device__ __forceinline__ void chunk_mma(Acc& A, const Act& X, int ks0, const float* w) {
    constexpr int N = NKS * 8;
    const unsigned addr = (unsigned)(size_t)w;          // LDS byte address of this lane's 16 bytes of item 0
    // everything the compiler itself has in flight on the LDS queue must be drained first: the counted waits
    // below assume only these reads are outstanding
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    Ring r;
    lds_read_pair<0>(r.h0, r.l0, addr);
    if constexpr (N > 1) lds_read_pair<2048>(r.h1, r.l1, addr);
    ChunkSteps<0, N>::run(A, X, ks0, addr, r);
}

// Epilogue of output tile T, pair P (registers 2P, 2P+1): activation, then either the next layer's
// unit 2T + (P>>2) (hi/lo split, packed) or the final rows' dot products.
template <int T, int P, bool SOFTPLUS, bool TANGENT, bool LAST, int NROWS>
__device__ __forceinline__ void epi_pair(const f32x16& acc, const Epi& e, Act& X, float (&dot)[NROWS], int h, bool is_val) {
    constexpr int g = P >> 1, c = 2 * (P & 1);                 // feature = 32T + 8g + 4h + c + {0,1}
    const int fo = 32 * T + 8 * g + 4 * h;
#ifdef NERFART_ABLATE_EPI       // timing experiments only: no activation / split arithmetic
    if (!LAST) { X.h[2 * T + (P >> 2)][P & 3] = __float_as_uint(acc[2 * P]); X.l[2 * T + (P >> 2)][P & 3] = __float_as_uint(acc[2 * P + 1]); }
    else dot[0] += acc[2 * P] + acc[2 * P + 1];
    return;
#endif
    float y[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float a = acc[2 * P + k];
        if (SOFTPLUS) {
            if (TANGENT) {
                float v, d;
                softplus100_vd(a, v, d);        // value lanes carry the bias from the accumulator init
                d = quad_bcast0(d);
                y[k] = is_val ? v : d * a;
            } else {
                y[k] = softplus100(a);
            }
        } else {
            y[k] = fmaxf(a, e.floor);
        }
    }
    if (LAST) {
#pragma unroll
        for (int n = 0; n < NROWS; ++n) {
            const f32x2 wv = *reinterpret_cast<const f32x2*>(e.rows + n * 256 + fo + c);
            dot[n] = fmaf(y[0], wv[0], dot[n]);
            dot[n] = fmaf(y[1], wv[1], dot[n]);
        }
        if (TANGENT) {
            if (e.h7 != nullptr && is_val) *reinterpret_cast<f32x2*>(e.h7 + fo + c) = f32x2{y[0], y[1]};
        }
    } else {
        unsigned hi, lo;
        split2(y[0], y[1], hi, lo);
        X.h[2 * T + (P >> 2)][P & 3] = hi;
        X.l[2 * T + (P >> 2)][P & 3] = lo;
    }
}

template <int T, bool SOFTPLUS, bool TANGENT, bool LAST, int NROWS>
__device__ __forceinline__ void epi_tile(const f32x16& acc, const Epi& e, Act& X, float (&dot)[NROWS], int h, bool is_val) {
    epi_pair<T, 0, SOFTPLUS, TANGENT, LAST, NROWS>(acc, e, X, dot, h, is_val);
    epi_pair<T, 1, SOFTPLUS, TANGENT, LAST, NROWS>(acc, e, X, dot, h, is_val);
    epi_pair<T, 2, SOFTPLUS, TANGENT, LAST, NROWS>(acc, e, X, dot, h, is_val);
    epi_pair<T, 3, SOFTPLUS, TANGENT, LAST, NROWS>(acc, e, X, dot, h, is_val);
    epi_pair<T, 4, SOFTPLUS, TANGENT, LAST, NROWS>(acc, e, X, dot, h, is_val);
    epi_pair<T, 5, SOFTPLUS, TANGENT, LAST, NROWS>(acc, e, X, dot, h, is_val);
    epi_pair<T, 6, SOFTPLUS, TANGENT, LAST, NROWS>(acc, e, X, dot, h, is_val);
    epi_pair<T, 7, SOFTPLUS, TANGENT, LAST, NROWS>(acc, e, X, dot, h, is_val);
}

// A whole dense layer, in place on X.  Chunk sequence (must match packing.py, bf16 plans): ceil(NU_BASE/4)
// chunks of the base units, then (if nextra) one chunk with the nextra extra units.
template <int NU_BASE, int NU_EXTRA_MAX, bool SOFTPLUS, bool TANGENT, bool LAST, int NROWS>
__device__ __forceinline__ void run_layer(Act& X, Pipe& p, const Epi& e, float (&dot)[NROWS], int ntiles, int nextra) {
    const int lane = lane_id();
    const int h = lane >> 5;
    const bool is_val = !TANGENT || ((lane & 3) == 0);
    (void)ntiles;
    Acc A;
#pragma unroll
    for (int T = 0; T < 8; ++T) {
        // start at the bias (value columns; derivative columns start at 0): register 4g + c of lane (h, j) is
        // feature 32T + 8g + 4h + c
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(e.bias + 32 * T + 8 * g + 4 * h);
#pragma unroll
            for (int c = 0; c < 4; ++c) A.t[T][4 * g + c] = is_val ? b[c] : 0.f;
        }
    }
#pragma unroll
    for (int c = 0; c < (NU_BASE + CHUNK_KS - 1) / CHUNK_KS; ++c) {
        const float* w = pipe_acquire(p) + lane * 4;
        constexpr int REM = NU_BASE % CHUNK_KS;
        if (REM != 0 && c == NU_BASE / CHUNK_KS) chunk_mma<(REM ? REM : 1)>(A, X, c * CHUNK_KS, w);
        else chunk_mma<CHUNK_KS>(A, X, c * CHUNK_KS, w);
    }
    if (NU_EXTRA_MAX > 0) {
        if (nextra > 0) {
            const float* w = pipe_acquire(p) + lane * 4;
            if (NU_EXTRA_MAX == 1 || nextra == 1) chunk_mma<1>(A, X, NU_BASE, w);
            else chunk_mma<(NU_EXTRA_MAX > 1 ? NU_EXTRA_MAX : 1)>(A, X, NU_BASE, w);
        }
    }
    epi_tile<0, SOFTPLUS, TANGENT, LAST, NROWS>(A.t[0], e, X, dot, h, is_val);
    __builtin_amdgcn_sched_barrier(0);
    epi_tile<1, SOFTPLUS, TANGENT, LAST, NROWS>(A.t[1], e, X, dot, h, is_val);
    __builtin_amdgcn_sched_barrier(0);
    epi_tile<2, SOFTPLUS, TANGENT, LAST, NROWS>(A.t[2], e, X, dot, h, is_val);
    __builtin_amdgcn_sched_barrier(0);
    epi_tile<3, SOFTPLUS, TANGENT, LAST, NROWS>(A.t[3], e, X, dot, h, is_val);
    __builtin_amdgcn_sched_barrier(0);
    epi_tile<4, SOFTPLUS, TANGENT, LAST, NROWS>(A.t[4], e, X, dot, h, is_val);
    __builtin_amdgcn_sched_barrier(0);
    epi_tile<5, SOFTPLUS, TANGENT, LAST, NROWS>(A.t[5], e, X, dot, h, is_val);
    __builtin_amdgcn_sched_barrier(0);
    epi_tile<6, SOFTPLUS, TANGENT, LAST, NROWS>(A.t[6], e, X, dot, h, is_val);
    __builtin_amdgcn_sched_barrier(0);
    epi_tile<7, SOFTPLUS, TANGENT, LAST, NROWS>(A.t[7], e, X, dot, h, is_val);
}

// ---------------------------------------------------------------------------------------
// Positional encoding of the SDF net in unit order: 3 units x (2 halves x 8 slots) = 48 slots, slot
// (q, h, e) <-> feature 8q + e + 21h (h = 0: x, y, z and bands 0..2; h = 1: bands 3..5), i.e. the
// reference's own feature order (models/base.py:53-61) split in two halves of 21 and 18.
// dq < 0: values; dq = 0..2: derivative w.r.t. coordinate dq.  Output scaled by 1/div when div != 1.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void encode_units(float x, float y, float z, int h, int dq, bool scale, Act& X, int u0) {
    // m0: slot values of half 0 (raw xyz, bands 0..2), m1: of half 1 (bands 3..5); each lane keeps its half's
    float m0[24], m1[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) { m0[k] = 0.f; m1[k] = 0.f; }
    const float co[3] = {x, y, z};
    const float fb = h ? 8.f : 1.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) m0[c] = (dq < 0) ? co[c] : ((dq == c) ? 1.f : 0.f);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float f = fb * (float)(1 << k);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float s, cs;
            sincosf(co[c] * f, &s, &cs);
            float vs = s, vc = cs;
            if (dq >= 0) { vs = (dq == c) ? cs * f : 0.f; vc = (dq == c) ? -(s * f) : 0.f; }
            m0[3 + 6 * k + c] = vs; m0[6 + 6 * k + c] = vc;
            m1[6 * k + c] = vs;     m1[3 + 6 * k + c] = vc;
        }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            float a = h ? m1[8 * q + 2 * pr] : m0[8 * q + 2 * pr];
            float b = h ? m1[8 * q + 2 * pr + 1] : m0[8 * q + 2 * pr + 1];
            if (scale) { a *= 0.70710678118654752440f; b *= 0.70710678118654752440f; }
            unsigned hi, lo;
            split2(a, b, hi, lo);
            X.h[u0 + q][pr] = hi;
            X.l[u0 + q][pr] = lo;
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
// h7 if asked).  Bodies: layer 0 (3 input units), layers 1..6 (one body in a run-time loop), layer 7 (LAST).
template <bool TANGENT>
__device__ __forceinline__ float surface_chain(Act& X, float px, float py, float pz, int h, int dq, Pipe& p,
                                               const float* aux, float* h7_lane) {
    float dot[1] = {0.f};
    Epi e{aux, 0.f, aux + SURF_AUX_ROW, h7_lane};
    encode_units(px, py, pz, h, dq, false, X, 0);
    run_layer<3, 0, true, TANGENT, false, 1>(X, p, e, dot, 8, 0);
#pragma nounroll
    for (int L = 1; L < 7; ++L) {
        // skip: cat[h(217), enc(39)] / sqrt(2) - the 1/sqrt(2) is folded into layer 4's packed weights
        if (L == 4) encode_units(px, py, pz, h, dq, false, X, 14);
        e.bias = aux + L * 256;
        run_layer<16, 1, true, TANGENT, false, 1>(X, p, e, dot, (L == 3) ? 7 : 8, (L == 4) ? 1 : 0);
    }
    e.bias = aux + 7 * 256;
    run_layer<16, 0, true, TANGENT, true, 1>(X, p, e, dot, 8, 0);
    // the two halves of a column hold complementary feature sets
    return dot[0] + __shfl_xor(dot[0], 32, 64);
}

// =======================================================================================
// K2 (split bf16): sdf only, 128 points per workgroup tile, 32 per wave.
// =======================================================================================
__global__ void __launch_bounds__(WG_THREADS, 1)
k_sdf_only_bf16(const float* __restrict__ blob, PointSrc src, float R_bg, float* __restrict__ sdf_out, int out_stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int* hdr = reinterpret_cast<const int*>(blob);
    float* aux = smem + 2 * CHUNK_FLOATS_MAX;
    const int lane = lane_id(), h = lane >> 5, j = lane & 31, wv = wave_id();
    load_aux(aux, blob, hdr, SURF_AUX_FLOATS);
    const unsigned ntiles = (src.M + 127u) / 128u;
    if (blockIdx.x >= ntiles) return;
    Pipe p{blob, reinterpret_cast<const int*>(aux + AUX_FLOATS_MAX), smem, hdr[2], 0, 0, 0, 0, false};
    p.wrap = (blockIdx.x + gridDim.x) < ntiles;
    pipe_start(p);
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        p.wrap = (tile + gridDim.x) < ntiles;
        const unsigned m = tile * 128u + wv * 32 + j;
        const Pt pt = fetch_point(src, m, false);
        Act X;
        float sdf = surface_chain<false>(X, pt.x, pt.y, pt.z, h, -1, p, aux, nullptr) + aux[SURF_AUX_B8];
        if (R_bg > 0.f) sdf = fminf(sdf, R_bg - sqrtf(pt.x * pt.x + pt.y * pt.y + pt.z * pt.z));
        if (h == 0 && m < src.M) {
            if (src.pts) sdf_out[m] = sdf;
            else {
                const unsigned slot = m / (unsigned)src.n_per_ray;
                sdf_out[(size_t)slot * out_stride + (m - slot * (unsigned)src.n_per_ray)] = sdf;
            }
        }
    }
}

// =======================================================================================
// K3a (split bf16): sdf + nabla + h7, forward mode; 32 points per workgroup tile (8 per wave, quads).
// =======================================================================================
__global__ void __launch_bounds__(WG_THREADS, 1)
k_sdf_nabla_bf16(const float* __restrict__ blob, PointSrc src, float R_bg, float* __restrict__ sdf_out,
                 float* __restrict__ nabla_out, float* __restrict__ h7_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int* hdr = reinterpret_cast<const int*>(blob);
    float* aux = smem + 2 * CHUNK_FLOATS_MAX;
    const int lane = lane_id(), h = lane >> 5, j = lane & 31, wv = wave_id();
    const int cq = j & 3;
    load_aux(aux, blob, hdr, SURF_AUX_FLOATS);
    const unsigned ntiles = (src.M + 31u) / 32u;
    if (blockIdx.x >= ntiles) return;
    Pipe p{blob, reinterpret_cast<const int*>(aux + AUX_FLOATS_MAX), smem, hdr[2], 0, 0, 0, 0, false};
    p.wrap = (blockIdx.x + gridDim.x) < ntiles;
    pipe_start(p);
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        p.wrap = (tile + gridDim.x) < ntiles;
        const unsigned m = tile * 32u + wv * 8 + (j >> 2);
        const Pt pt = fetch_point(src, m, false);
        Act X;
        float* h7_lane = (h7_out != nullptr && m < src.M) ? h7_out + (size_t)m * 256 : nullptr;
        const float v = surface_chain<true>(X, pt.x, pt.y, pt.z, h, cq - 1, p, aux, h7_lane);
        if (m < src.M && h == 0) {
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
// K3b (split bf16): radiance net.  VE extra units: 1 (VolSDF, 9 extras) or 3 (NeuS, 33 extras).
// =======================================================================================
template <int VE>
__device__ __forceinline__ void radiance_extras(const Pt& pt, float nx, float ny, float nz, int h, Act& X) {
    constexpr int NE = (VE == 1) ? 9 : 33;
    float ex[VE * 16];
#pragma unroll
    for (int k = 0; k < VE * 16; ++k) ex[k] = 0.f;
    ex[0] = pt.x; ex[1] = pt.y; ex[2] = pt.z;
    const float v[3] = {pt.vx, pt.vy, pt.vz};
#pragma unroll
    for (int c = 0; c < 3; ++c) ex[3 + c] = v[c];
    if (VE == 3) {
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
    // slot (q, h, e) <-> extra index 16q + 8h + e
#pragma unroll
    for (int q = 0; q < VE; ++q)
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            const float a = h ? ex[16 * q + 8 + 2 * pr] : ex[16 * q + 2 * pr];
            const float b = h ? ex[16 * q + 8 + 2 * pr + 1] : ex[16 * q + 2 * pr + 1];
            unsigned hi, lo;
            split2(a, b, hi, lo);
            X.h[16 + q][pr] = hi;
            X.l[16 + q][pr] = lo;
        }
}

template <int VE>
__global__ void __launch_bounds__(WG_THREADS, 1)
k_radiance_bf16(const float* __restrict__ blob, PointSrc src, const float* __restrict__ nabla_in,
                const float* __restrict__ h7_in, float* __restrict__ rgb_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int* hdr = reinterpret_cast<const int*>(blob);
    float* aux = smem + 2 * CHUNK_FLOATS_MAX;
    const int lane = lane_id(), h = lane >> 5, j = lane & 31, wv = wave_id();
    load_aux(aux, blob, hdr, RAD_AUX_FLOATS);
    const unsigned ntiles = (src.M + 127u) / 128u;
    if (blockIdx.x >= ntiles) return;
    Pipe p{blob, reinterpret_cast<const int*>(aux + AUX_FLOATS_MAX), smem, hdr[2], 0, 0, 0, 0, false};
    p.wrap = (blockIdx.x + gridDim.x) < ntiles;
    pipe_start(p);
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        p.wrap = (tile + gridDim.x) < ntiles;
        const unsigned m = tile * 128u + wv * 32 + j;
        const bool valid = m < src.M;
        const Pt pt = fetch_point(src, m, true);
        Act X;
        float nx = 0.f, ny = 0.f, nz = 0.f;
        if (valid) { nx = nabla_in[(size_t)m * 3 + 0]; ny = nabla_in[(size_t)m * 3 + 1]; nz = nabla_in[(size_t)m * 3 + 2]; }
        // h7 -> units: unit 2T+u, slot e <-> feature 32T + (r&3) + 8(r>>2) + 4h with r = 8u + e
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            f32x4 lo4 = {0.f, 0.f, 0.f, 0.f}, hi4 = {0.f, 0.f, 0.f, 0.f};
            if (valid) {
                const float* s0 = h7_in + (size_t)m * 256 + 32 * (u >> 1) + 16 * (u & 1) + 4 * h;
                lo4 = *reinterpret_cast<const f32x4*>(s0);          // r = 8u' + 0..3  -> rows 8(2u') + 4h + 0..3
                hi4 = *reinterpret_cast<const f32x4*>(s0 + 8);      // r = 8u' + 4..7  -> rows 8(2u'+1) + 4h + 0..3
            }
            unsigned sh[4], sl[4];
            split2(lo4[0], lo4[1], sh[0], sl[0]);
            split2(lo4[2], lo4[3], sh[1], sl[1]);
            split2(hi4[0], hi4[1], sh[2], sl[2]);
            split2(hi4[2], hi4[3], sh[3], sl[3]);
            X.h[u] = u32x4{sh[0], sh[1], sh[2], sh[3]};
            X.l[u] = u32x4{sl[0], sl[1], sl[2], sl[3]};
        }
        radiance_extras<VE>(pt, nx, ny, nz, h, X);
        float dot[3] = {0.f, 0.f, 0.f};
        Epi e{aux, -INFINITY, aux + RAD_AUX_ROWS, nullptr};
        // L = 0: geometry feature (no activation); L = 1: [feat | x, v, n] -> 256 ReLU; L = 2, 3: ReLU; L = 4: LAST
#pragma nounroll
        for (int L = 0; L < 4; ++L) {
            e.bias = aux + L * 256;
            e.floor = (L == 0) ? -INFINITY : 0.f;
            run_layer<16, VE, false, false, false, 3>(X, p, e, dot, 8, (L == 1) ? VE : 0);
        }
        e.bias = aux + 4 * 256;
        e.floor = 0.f;
        run_layer<16, 0, false, false, true, 3>(X, p, e, dot, 8, 0);
        float c[3];
#pragma unroll
        for (int n = 0; n < 3; ++n) c[n] = sigmoidf_(dot[n] + __shfl_xor(dot[n], 32, 64) + aux[RAD_AUX_BF + n]);
        if (valid && h == 0) { rgb_out[(size_t)m * 3 + 0] = c[0]; rgb_out[(size_t)m * 3 + 1] = c[1]; rgb_out[(size_t)m * 3 + 2] = c[2]; }
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

// Entry points used by mlp_chain.hip's dispatchers when the blob carries a split-bf16 program.
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
    if (view_tiles == 3) return b16::launch_chain(2, (long long)s.M, b16::k_radiance_bf16<3>, nt, st, blob, s, nabla, h7, rgb);
    set_last_error("radiance_fwd: view_tiles must be 1 (raw view dirs) or 3 (multires_view = 4)");
    return 2;
}
}  // namespace nerfart

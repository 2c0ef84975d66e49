// mlp_k2_w32.hip - K2 (SDF only, split-bf16) restructured: ONE wave per SIMD, 32 columns per wave.
//
// Same arithmetic as k_sdf_only_bf16 (mlp_chain_bf16.hip: split-bf16 operands, hi.hi + hi.lo + lo.hi, fp32 accumulate) on
// v_mfma_f32_32x32x16_bf16: a workgroup is 4 waves (one per SIMD, 512 registers each: both accumulator sets in the ACC half), a
// wave carries 32 columns, its own program of the surface blob (packing.py: fragments in the 32x32x16 layout, header words 8 / 9).
//   * half the MFMA instructions and half the LDS fragment bytes per flop of the 16x16x32 form (one 1 KiB fragment feeds a
//     32 x 32 x 16 product), the matrix pipe's better shape (32 cycles per instruction);
//   * no co-resident partner wave on the SIMD: the 8-wave kernel's two waves per SIMD ran the same stream in lockstep and queued
//     for the matrix pipe together.  Here overlap is in-wave: a "double item" = tiles 2 D, 2 D + 1 of one
//     16-deep k-step = 3 MFMA pairs on independent accumulators (no MFMA depends on the one issued before it), each pair
//     followed by its share of the fillers - the 4 fragment reads of the next double item, the LDS-DMA pieces, one epilogue
//     slice (of the unit the NEXT k-step consumes) - pinned with sched_barrier;
//   * the chunk barrier synchronises 4 waves instead of 8.
// Weight stream: 64 KiB chunks (4 k-steps x 8 tiles x (hi, lo) x 1 KiB), double buffered, 16 LDS-DMA pieces per wave per chunk.
#include "mlp_bf16_core.h"
#include <cstdlib>

namespace nerfart {
namespace w32 {

using namespace b16;

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int W_THREADS = 256;
constexpr int W_KS = 4;                   // 16-deep k-steps per 64 KiB chunk (8 output tiles x (hi, lo) x 1 KiB each)

struct Acc32 { f32x16 t[8]; };            // output tile T: reg r <-> feature 32 T + 8 (r >> 2) + 4 h + (r & 3), h = lane >> 5
struct Frag { u32x4 h, l; };              // A fragment of one item (32 rows x 16 k), hi and lo terms

// One MFMA.  Accumulators live in the ACC half of the unified register file ("a"): a one-wave-per-SIMD kernel has 512 registers of
// which VALU instructions address only the 256 architectural ones; the epilogue stages fetch the values they need with
// v_accvgpr_read.  An in-order wave that issues two MFMAs back to back sits at the second one until the matrix pipe frees up
// (32 cycles) with nothing else issued: every MFMA is therefore followed by its own share of fillers (measured: pairs of MFMAs
// with the fillers behind the pair ran 335 cycles per 6 MFMAs, i.e. nothing overlapped).
__device__ __forceinline__ void mfma1(const u32x4 a, const u32x4 b, f32x16& q) {
#ifdef W32_NO_MFMA           // timing experiments only (tools/archive/ablate_w32.py): operands kept live, no matrix work - results are wrong
    asm volatile("" : "+a"(q) : "v"(a), "v"(b));
    return;
#endif
    asm volatile("s_nop 0\n\t"
                 "v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"
                 : "+a"(q) : "v"(a), "v"(b));
}

// ---- weight stream: wave w copies bytes [16 KiB w, 16 KiB (w + 1)) of the next chunk, 16 pieces of 1 KiB -----------------
struct StreamW {
    const float* blob;
    const int* tab;
    float* lds;
    int nc;
    const float* src_lo;       // pieces 0..7
    const float* src_hi;       // pieces 8..15
    unsigned dst_lo, dst_hi;
    unsigned voff_a, voff_b;   // lane * 16 (+ 4096)
    int nxt, nxt_o0;
    int pb;
    bool wrap;
};
// piece J: global = src + voff (lane * 16, + 4096 for J & 4) + (J & 3) KiB; LDS = M0 + (J & 3) KiB + lane * 16 with M0 = dst + (J & 4) KiB
// (the instruction's immediate offset applies to both addresses, the vector offset to the global one only)
template <int J>
__device__ __forceinline__ void w_piece(const StreamW& s) {
#ifdef W32_NO_DMA
    return;
#endif
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_add_u32 m0, %3, %4\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:%5\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"((J & 4) ? s.voff_b : s.voff_a), "s"(J < 8 ? s.src_lo : s.src_hi), "s"(J < 8 ? s.dst_lo : s.dst_hi), "i"((J & 4) * 1024),
                   "i"((J & 3) * 1024)
                 : "memory", "scc");
}
template <int J0, int J1>
__device__ __forceinline__ void w_pieces(const StreamW& s) {
    if constexpr (J0 < J1) { w_piece<J0>(s); w_pieces<J0 + 1, J1>(s); }
}
__device__ __forceinline__ void w_lookup(StreamW& s, int chunk) {
    s.nxt = chunk;
    s.nxt_o0 = __builtin_amdgcn_readfirstlane(s.tab[chunk >= 0 ? chunk : 0]);
}
__device__ __forceinline__ int w_next_of(const StreamW& s, int c) {
    if (c < 0) return -1;
    return (c + 1 == s.nc) ? (s.wrap ? 0 : -1) : c + 1;
}
__device__ __forceinline__ void w_target(StreamW& s, int o0, int buf) {
    const int w = wave_id();
    s.src_lo = s.blob + o0 + w * 4096;
    s.src_hi = s.src_lo + 2048;
    s.dst_lo = lds_addr(s.lds + buf * CHUNK_FLOATS) + w * 16384;
    s.dst_hi = s.dst_lo + 8192;
}
__device__ __forceinline__ void w_start(StreamW& s) {
    s.voff_a = lane_id() * 16;
    s.voff_b = lane_id() * 16 + 4096;
    w_lookup(s, 0);
    w_target(s, s.nxt_o0, 0);
    w_pieces<0, 16>(s);
    s.pb = 0;
    w_lookup(s, w_next_of(s, 0));
}
__device__ __forceinline__ const float* w_acquire(StreamW& s) {
    wait_glds();
    __syncthreads();
    const float* w = s.lds + s.pb * CHUNK_FLOATS;
    w_target(s, s.nxt_o0, s.pb ^ 1);
    w_lookup(s, w_next_of(s, s.nxt));
    s.pb ^= 1;
    return w;
}

// ---- layer shape: NH input units (16-deep k-steps) come from act(P), NX are ready-made units xs[]; NEXT0: the last k-step also
// builds unit 0 of the next layer from this layer's tile 0.
template <int NH_, int NX_, bool NEXT0_>
struct LCfg {
    static constexpr int NH = NH_, NX = NX_, NXA = NX_ > 0 ? NX_ : 1, NKS = NH_ + NX_;
    static constexpr bool NEXT0 = NEXT0_;
    static constexpr int hosted(int ks) { return (ks + 1 < NH_) ? ks + 1 : ((NEXT0_ && ks == NH_ + NX_ - 1) ? 100 : -1); }
};

// ---- epilogue: softplus(beta = 100) + split to bf16 hi / lo of one unit (32 features x 32 columns: 4 pairs of values per lane), cut
// into STAGES of mutually independent instructions.  A wave alone on its SIMD has nobody to cover its own dependent VALU chains
// (exp -> add -> log -> fma -> cvt ...): consecutive stages of a pair therefore run in consecutive GAPS (behind successive MFMA
// pairs, >= 64 cycles apart) and all four pairs of the unit advance in lockstep, so every instruction inside a gap is independent
// of the others.  Same arithmetic as epi_phase<0, .> of mlp_bf16_core.h.
struct PS { float z0, z1, u0, u1, r0, r1, y0, y1; unsigned hi; };

template <int ST>
__device__ __forceinline__ void ps_stage(PS& p, float zin0, float zin1, unsigned& out_hi, unsigned& out_lo) {
    if constexpr (ST == 0) { p.z0 = zin0; p.z1 = zin1; }                                                  // fetch from the ACC file
    else if constexpr (ST == 1) { p.u0 = fabsf(p.z0) * -144.269504088896340736f; p.u1 = fabsf(p.z1) * -144.269504088896340736f; }
    else if constexpr (ST == 2) { p.r0 = relu1(p.z0); p.r1 = relu1(p.z1); }
    else if constexpr (ST == 3) { p.u0 = __builtin_amdgcn_exp2f(p.u0); p.u1 = __builtin_amdgcn_exp2f(p.u1); }   // exp(-|100 z|)
    else if constexpr (ST == 4) { p.u0 = 1.0f + p.u0; p.u1 = 1.0f + p.u1; }
    else if constexpr (ST == 5) { p.u0 = __builtin_amdgcn_logf(p.u0); p.u1 = __builtin_amdgcn_logf(p.u1); }
    else if constexpr (ST == 6) { p.y0 = fmaf(p.u0, 0.69314718055994530942f / 100.0f, p.r0); p.y1 = fmaf(p.u1, 0.69314718055994530942f / 100.0f, p.r1); }
    else if constexpr (ST == 7) { p.hi = pack_bf16(p.y0, p.y1); }
    else if constexpr (ST == 8) { p.u0 = __uint_as_float(p.hi << 16); p.u1 = __uint_as_float(p.hi & 0xffff0000u); }
    else if constexpr (ST == 9) { p.u0 = p.y0 - p.u0; p.u1 = p.y1 - p.u1; }
    else if constexpr (ST == 10) { out_hi = p.hi; out_lo = pack_bf16(p.u0, p.u1); }
}
constexpr int N_STAGES = 11;

// A k-step has 4 double items x 6 MFMAs = 24 gaps (gap gk = 6 D + G sits behind MFMA G of double item D); a gap advances TWO of the
// unit's four pairs by one stage:
//   a unit built from P (the previous layer's accumulators): gap gk -> stage gk / 2 of pairs 2 (gk & 1), 2 (gk & 1) + 1 (gaps 0..21);
//   the NEXT layer's unit 0 (from this layer's tile 0, whose last MFMA is the fifth of double item 0: its result is read no earlier
//   than three MFMAs later): gaps 8..23, eight compressed stages (0+1, 2+4, 9+10 merged; 3, 5..8 alone).
template <int HU, int GK>
__device__ __forceinline__ void gap_stages(const Acc32& P, const Acc32& Q, Unit (&xb)[2], Unit& x0n, PS (&ps)[4]) {
#ifdef W32_NO_EPI
    return;
#endif
    if constexpr (HU == 100) {
        if constexpr (GK >= 8) {
            constexpr int c = (GK - 8) / 2, pg = GK & 1;
#pragma unroll
            for (int p = 2 * pg; p < 2 * pg + 2; ++p) {
                unsigned hi = 0, lo = 0;
                const float z0 = Q.t[0][2 * p], z1 = Q.t[0][2 * p + 1];
                if constexpr (c == 0) { ps_stage<0>(ps[p], z0, z1, hi, lo); ps_stage<1>(ps[p], z0, z1, hi, lo); }
                else if constexpr (c == 1) ps_stage<3>(ps[p], z0, z1, hi, lo);
                else if constexpr (c == 2) { ps_stage<2>(ps[p], z0, z1, hi, lo); ps_stage<4>(ps[p], z0, z1, hi, lo); }
                else if constexpr (c < 7) ps_stage<c + 2>(ps[p], z0, z1, hi, lo);
                else {
                    ps_stage<9>(ps[p], z0, z1, hi, lo);
                    ps_stage<10>(ps[p], z0, z1, hi, lo);
                    x0n.h[p] = hi; x0n.l[p] = lo;
                }
            }
        }
    } else if constexpr (HU >= 0 && GK < 2 * N_STAGES) {
        constexpr int st = GK / 2, pg = GK & 1, r0 = 8 * (HU & 1);
#pragma unroll
        for (int p = 2 * pg; p < 2 * pg + 2; ++p) {
            unsigned hi = 0, lo = 0;
            ps_stage<st>(ps[p], P.t[HU >> 1][r0 + 2 * p], P.t[HU >> 1][r0 + 2 * p + 1], hi, lo);
            if constexpr (st == 10) { xb[HU & 1].h[p] = hi; xb[HU & 1].l[p] = lo; }
        }
    }
}

template <int OFF>
__device__ __forceinline__ void lds_read_one(u32x4& f, unsigned addr) {
#ifdef W32_NO_LDS
    asm volatile("; no read" : "=&v"(f) : "v"(addr));
    return;
#endif
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(f) : "v"(addr), "i"(OFF));
}

// One chunk = NKC k-steps x 4 double items (tiles 2 D, 2 D + 1).  DI = double item index inside the chunk.  MFMA order inside a
// double item: (tile, term) = (T, hh) (T+1, hh) (T, hl) (T+1, hl) (T, lh) (T+1, lh) - no MFMA depends on the one issued before it.
// Gaps 0..3 also carry one of the four fragment reads of the NEXT double item, gap 4 this double item's share of the LDS-DMA pieces.
template <class L, int C, int NKC, int DI>
struct Items32 {
    static __device__ __forceinline__ void run(const Acc32& P, Acc32& Q, Unit (&xb)[2], const Unit (&xs)[L::NXA], Unit& x0n, PS (&ps)[4],
                                               Frag (&r)[2][2], unsigned addr, const StreamW& s) {
        constexpr int ND = NKC * 4;
        if constexpr (DI < ND) {
            constexpr int kk = DI >> 2, D = DI & 3, ks = W_KS * C + kk, T = 2 * D;
            constexpr int S = DI & 1;
            constexpr bool NEXT = DI + 1 < ND;
            // this double item's four fragments were requested during the previous one (or by run_chunk_w)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[S][0].h), "+v"(r[S][0].l), "+v"(r[S][1].h), "+v"(r[S][1].l));
            u32x4 bh, bl;
            if constexpr (ks < L::NH) { bh = xb[ks & 1].h; bl = xb[ks & 1].l; }
            else { bh = xs[ks - L::NH].h; bl = xs[ks - L::NH].l; }
            constexpr int HU = L::hosted(ks);
#ifndef W32_DMA_AT          // where the double item's LDS-DMA pieces go: gap * 2 + (0: right behind the gap's MFMA / read, 1: behind its stages).
#define W32_DMA_AT 0        // Swept on MI355X (tools/archive/sweep_w32.py, profiles/r02s_sweep_w32.log): 9.47 ms (gap 0) .. 9.57 ms (gap 5), same bits.
#endif
#define W32_PIECES(G, POS) if constexpr (W32_DMA_AT == 2 * (G) + (POS)) w_pieces<(DI * 16) / ND, ((DI + 1) * 16) / ND>(s)
            mfma1(r[S][0].h, bh, Q.t[T]);
            if constexpr (NEXT) lds_read_one<(2 * DI + 2) * 2048>(r[S ^ 1][0].h, addr);
            W32_PIECES(0, 0);
            gap_stages<HU, 6 * D + 0>(P, Q, xb, x0n, ps);
            W32_PIECES(0, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma1(r[S][1].h, bh, Q.t[T + 1]);
            if constexpr (NEXT) lds_read_one<(2 * DI + 3) * 2048>(r[S ^ 1][1].h, addr);
            W32_PIECES(1, 0);
            gap_stages<HU, 6 * D + 1>(P, Q, xb, x0n, ps);
            W32_PIECES(1, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma1(r[S][0].h, bl, Q.t[T]);
            if constexpr (NEXT) lds_read_one<(2 * DI + 2) * 2048 + 1024>(r[S ^ 1][0].l, addr);
            W32_PIECES(2, 0);
            gap_stages<HU, 6 * D + 2>(P, Q, xb, x0n, ps);
            W32_PIECES(2, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma1(r[S][1].h, bl, Q.t[T + 1]);
            if constexpr (NEXT) lds_read_one<(2 * DI + 3) * 2048 + 1024>(r[S ^ 1][1].l, addr);
            W32_PIECES(3, 0);
            gap_stages<HU, 6 * D + 3>(P, Q, xb, x0n, ps);
            W32_PIECES(3, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma1(r[S][0].l, bh, Q.t[T]);
            W32_PIECES(4, 0);
            gap_stages<HU, 6 * D + 4>(P, Q, xb, x0n, ps);
            W32_PIECES(4, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma1(r[S][1].l, bh, Q.t[T + 1]);
            W32_PIECES(5, 0);
            gap_stages<HU, 6 * D + 5>(P, Q, xb, x0n, ps);
            W32_PIECES(5, 1);
            __builtin_amdgcn_sched_barrier(0);
#undef W32_PIECES
            Items32<L, C, NKC, DI + 1>::run(P, Q, xb, xs, x0n, ps, r, addr, s);
        }
    }
};

template <class L, int C>
__device__ __forceinline__ void run_chunk_w(const Acc32& P, Acc32& Q, Unit (&xb)[2], const Unit (&xs)[L::NXA], Unit& x0n, PS (&ps)[4], StreamW& s) {
    constexpr int NKC = (L::NKS - W_KS * C) >= W_KS ? W_KS : (L::NKS - W_KS * C);
    if constexpr (NKC > 0) {
        const float* wp = w_acquire(s) + lane_id() * 4;
        const unsigned addr = (unsigned)(size_t)wp;        // LDS byte address of this lane's 16 bytes of item 0
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        Frag r[2][2];
        lds_read_one<0>(r[0][0].h, addr);
        lds_read_one<2048>(r[0][1].h, addr);
        lds_read_one<1024>(r[0][0].l, addr);
        lds_read_one<3072>(r[0][1].l, addr);
        Items32<L, C, NKC, 0>::run(P, Q, xb, xs, x0n, ps, r, addr, s);
        run_chunk_w<L, C + 1>(P, Q, xb, xs, x0n, ps, s);
    }
}

// One dense layer: Q = bias + W . [act(P) | xs].  x0 = unit 0 of act(P) (built by the previous layer).
template <class L>
__device__ __forceinline__ void layer_w(const Acc32& P, Acc32& Q, const Unit& x0, const Unit (&xs)[L::NXA], Unit& x0n, StreamW& s, const float* bias) {
    const int h = lane_id() >> 5;
    Unit xb[2];
    xb[0] = x0;
    xb[1] = x0;
#pragma unroll
    for (int T = 0; T < 8; ++T)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 32 * T + 8 * j + 4 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) Q.t[T][4 * j + i] = b[i];
        }
    PS ps[4] = {};
    run_chunk_w<L, 0>(P, Q, xb, xs, x0n, ps, s);
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");     // last MFMA result -> first reader of the accumulators
}

// Positional encoding in the unit order of packing.w32_feature_enc: half 0 holds x, y, z and the (sin, cos) pairs of bands 0..2,
// half 1 the pairs of bands 3..5 (reference Embedder, models/base.py:38-64): 9 sincos per lane.
__device__ __forceinline__ void encode_w32(float x, float y, float z, int h, Unit (&X)[3]) {
    float sc[18];
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int c = j % 3;
        const float cg = (c == 0) ? x : ((c == 1) ? y : z);
        const float f = (h == 0) ? (float)(1 << (j / 3)) : (float)(8 << (j / 3));
        sincosf(cg * f, &sc[2 * j], &sc[2 * j + 1]);
    }
    float v[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        const float a = (i < 3) ? ((i == 0) ? x : ((i == 1) ? y : z)) : sc[i < 3 ? 0 : (i - 3 < 18 ? i - 3 : 17)];
        const float b = (i < 18) ? sc[i] : 0.f;
        v[i] = (h == 0) ? ((i < 21) ? a : 0.f) : b;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        u32x4 hi, lo;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
            unsigned sh, sl;
            split2(v[8 * q + 2 * pr], v[8 * q + 2 * pr + 1], sh, sl);
            hi[pr] = sh; lo[pr] = sl;
        }
        X[q].h = hi;
        X[q].l = lo;
    }
}

// The 8 hidden layers for the wave's 32 columns; returns row0 . h7 (summed over the two lane halves).
__device__ __forceinline__ float surface_chain_w(float px, float py, float pz, int h, StreamW& s, const float* aux) {
    Acc32 A, B;
    Unit x0, x0n, enc[3], none[1];
    encode_w32(px, py, pz, h, enc);
    none[0] = enc[0];
    x0 = enc[0];
    // A = the previous layer's pre-activations, B = the layer being accumulated (both in the ACC file).  Three layer bodies only
    // (first / generic / skip; the last layer runs the generic body and its unused next-unit epilogue): 80 KB of code instead of the
    // 250 KB of eight unrolled bodies, which streamed through the instruction cache on every tile.
    layer_w<LCfg<0, 3, true>>(A, B, x0, enc, x0n, s, aux);
    A = B;
    x0 = x0n;
#pragma nounroll
    for (int L = 1; L < 8; ++L) {
        if (L == 4) layer_w<LCfg<14, 3, true>>(A, B, x0, enc, x0n, s, aux + L * 256);      // skip: 14 hidden units + 3 encoding units
        else layer_w<LCfg<16, 0, true>>(A, B, x0, none, x0n, s, aux + L * 256);
        if (L < 7) A = B;
        x0 = x0n;
    }
    const float* rows = aux + SURF_AUX_ROW;
    float d = 0.f;
#pragma unroll
    for (int T = 0; T < 8; ++T)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(rows + 32 * T + 8 * j + 4 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) d = fmaf(softplus100(B.t[T][4 * j + i]), wv[i], d);
        }
    return d + __shfl_xor(d, 32, 64);
}

__global__ void __launch_bounds__(W_THREADS, 1)
k_sdf_only_w32(const float* __restrict__ blob, PointSrc src, float R_bg, float* __restrict__ sdf_out, int out_stride) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int* hdr = reinterpret_cast<const int*>(blob);
    float* aux = smem + 2 * CHUNK_FLOATS;
    const int lane = lane_id(), h = lane >> 5, j = lane & 31, wv = wave_id();
    {
        const float* asrc = blob + hdr[4];
        for (int i = threadIdx.x; i < SURF_AUX_FLOATS; i += W_THREADS) aux[i] = asrc[i];
        int* tab = reinterpret_cast<int*>(aux + AUX_FLOATS_MAX);
        if (threadIdx.x < TAB_INTS) tab[threadIdx.x] = hdr[NERFART_HDR_OFFS + threadIdx.x];
        __syncthreads();
    }
    const unsigned ntiles = (src.M + 127u) / 128u;
    if (blockIdx.x >= ntiles) return;
    StreamW s;
    s.blob = blob; s.tab = reinterpret_cast<const int*>(aux + AUX_FLOATS_MAX) + hdr[9]; s.lds = smem; s.nc = hdr[8];     // the w32 program
    s.src_lo = s.src_hi = blob; s.dst_lo = s.dst_hi = 0; s.voff_a = s.voff_b = 0; s.nxt = -1; s.nxt_o0 = 0; s.pb = 0;
    s.wrap = (blockIdx.x + gridDim.x) < ntiles;
    w_start(s);
    for (unsigned tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        s.wrap = (tile + gridDim.x) < ntiles;
        const unsigned m = tile * 128u + wv * 32 + j;
        const Pt pt = fetch_point(src, m, false);
        float sdf = surface_chain_w(pt.x, pt.y, pt.z, h, s, aux) + aux[SURF_AUX_B8];
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

}  // namespace w32

// precision 1 entry used by the dispatchers of mlp_chain.hip.  The 8-wave kernel of mlp_chain_bf16.hip (k_sdf_only_bf16) stays the
// default: measured on MI355X (round 2, tools/archive/k2_ab.py, 4 M points) 9.45 ms against 9.56 .. 9.95 ms for every scheduling variant of
// this kernel - the ablation (tools/archive/ablate_w32.py, profiles/r02k_ablate_w32.log) puts the matrix work alone at 6.1 ms and prices
// the three filler classes at +2.3 ms (LDS-DMA issue, ~78 cycles per 1 KiB piece), +1.8 ms (fragment reads, ~16 cycles per
// ds_read_b128) and +1.7 ms (epilogue), of which a lone in-order wave hides only 2.3 ms: both designs are bound by weight bytes moved
// per column (L2 -> LDS -> registers), which only more columns per fragment would lower and the register file does not allow.
// NERFART_K2=w32 selects this kernel (same results to ~1e-5: a different summation order).
int sdf_bf16_v1(const float* blob, const PointSrc& s, float R_bg, float* out, int out_stride, hipStream_t st);
int sdf_bf16(const float* blob, const PointSrc& s, float R_bg, float* out, int out_stride, hipStream_t st) {
    static const bool use_w32 = [] { const char* e = std::getenv("NERFART_K2"); return e != nullptr && e[0] == 'w'; }();
    if (!use_w32) return sdf_bf16_v1(blob, s, R_bg, out, out_stride, st);
    const size_t lds = b16::LDS_FLOATS * sizeof(float);
    NERFART_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(w32::k_sdf_only_w32), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned ntiles = (s.M + 127u) / 128u;
    const unsigned grid = ntiles < (unsigned)num_cus() ? ntiles : (unsigned)num_cus();
    void* ph = nullptr;
    if (profile_enabled()) profile_open(0, (long long)s.M, st, &ph);
    hipLaunchKernelGGL(w32::k_sdf_only_w32, dim3(grid), dim3(w32::W_THREADS), lds, st, blob, s, R_bg, out, out_stride);
    profile_close(ph, st);
    NERFART_HIP(hipGetLastError());
    return 0;
}

}  // namespace nerfart

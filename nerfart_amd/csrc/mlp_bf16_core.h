// mlp_bf16_core.h - shared device code of the split-bf16 ("bf16x3") chained-MLP kernels: the weight stream, the epilogue
// slices, the item pipeline (Cfg / Items / layer) and the launcher.  See mlp_chain_bf16.hip for the design notes.
#pragma once
#include "mlp_common.h"

namespace nerfart {
namespace b16 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int WG_THREADS = 512;
constexpr int WAVES = 8;
constexpr int TS_FLOATS = 512;                      // one item = (k-step, output tile): (hi, lo) x 64 lanes x 16 B = 2 KiB
constexpr int KS_FLOATS = 16 * TS_FLOATS;           // one k-step of a chunk: 16 output tiles = 32 KiB
constexpr int CHUNK_KS = 2;                         // k-steps per chunk
constexpr int CHUNK_FLOATS = CHUNK_KS * KS_FLOATS;  // 64 KiB
constexpr int AUX_FLOATS_MAX = 2560;
constexpr int LDS_FLOATS = 2 * CHUNK_FLOATS + AUX_FLOATS_MAX + TAB_INTS;   // 141,824 B
// dump sizes shared by producers and consumers (bf16 or unorm16 units of 16 B per lane)
constexpr int RAD_DUMP_PER_TILE = 5 * 8 * 8 * 1024;      // radiance: 5 activations x 8 units x 8 waves, per 128 points

struct Unit { u32x4 h, l; };                        // 32 feature slots: packed bf16 pairs, hi and lo terms
struct Acc { f32x4 t[16]; };
struct Work { float e0, e1, r0, r1, y0, y1; };      // values carried between the slices of one epilogue pair
struct Work2 : Work { Work b; };                    // ... of two pairs in flight (plain softplus layers, see Items)
struct EpiCtx {
    float floor_p, floor_q;                         // ReLU family: activation floor of the previous / this layer's output
    bool is_val;                                    // tangent kernels: this lane is a value column
};

// NERFART_F16X2 (csrc/mlp_chain_f16x2.hip compiles this whole core a second time, namespace f16x2, C-ABI precision 4): the
// 2-MFMA split - ONE fp16 activation term (11 significant bits, TF32 class) against fp16 hi + lo weight terms:
//     a . w  ~  a16 . w_hi + a16 . w_lo            (v_mfma_f32_16x16x32_f16 x 2, fp32 accumulate)
// on the k-steps whose input unit is built from the previous layer's accumulators; the READY-MADE input units (positional encodings of
// layers 0 and 4, the radiance net's [x | v | n] and the h7 rows it reads from memory) keep a hi + lo pair and the three-term form - raw
// coordinates and their sines are where 11 bits hurt most (2e-3 absolute at |x| = 8), and they are 4 of K2's 59 k-steps.
// Same data flow, same blob geometry (the packer writes fp16 fragments), 2/3 of the matrix work.  A measurement variant: it
// exists to put the "fewer MFMAs per product" question to the statistics the shipped bf16x3 mode is held to.
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    f32x2_ v = {a, b};
#ifdef NERFART_F16X2
    typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_));      // v_cvt_pk_f16_f32 (RNE)
#else
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));       // v_cvt_pk_bf16_f32 (RNE)
#endif
}
__device__ __forceinline__ void split2(float y0, float y1, unsigned& hi, unsigned& lo) {
    hi = pack_bf16(y0, y1);
#ifdef NERFART_F16X2
    // the lo term is consumed only on the k-steps of READY-MADE input units (encodings, extras, h7 from memory: wait_mfma3<FULL>);
    // for the units built from accumulators nothing reads it and the three instructions below are dead code
    typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ hf = __builtin_convertvector(__builtin_bit_cast(f16x2_, hi), f32x2_);
    lo = pack_bf16(y0 - hf[0], y1 - hf[1]);
#else
    const float h0 = __uint_as_float(hi << 16), h1 = __uint_as_float(hi & 0xffff0000u);
    lo = pack_bf16(y0 - h0, y1 - h1);
#endif
}
__device__ __forceinline__ f32x4 mfma3(const u32x4 ah, const u32x4 al, const u32x4 bh, const u32x4 bl, f32x4 acc) {
#ifdef NERFART_ABLATE_MFMA      // timing experiments only (tools/ablate_bf16.py): keep the operands live, skip the matrix work
    asm volatile("" :: "v"(ah), "v"(al), "v"(bh), "v"(bl));
    return acc;
#endif
    // One asm statement so that nothing is scheduled BETWEEN the three MFMAs of an accumulator chain (an extra
    // issue slot there costs ~43 cycles, MI355X_MICROARCH.md "per-instruction cycle constants"); fillers go
    // between triples.  Hazards the compiler no longer pads: a VALU write of an A/B operand just before the
    // statement (s_nop 1 inside); the result read by a VALU after it (readers are >= 2 triples away, except at
    // the end of a layer: layer() ends with explicit nops).
#ifdef NERFART_F16X2
    asm volatile("s_nop 1\n\t"
                 "v_mfma_f32_16x16x32_f16 %0, %1, %3, %0\n\t"
                 "v_mfma_f32_16x16x32_f16 %0, %2, %3, %0"
                 : "+v"(acc) : "v"(ah), "v"(al), "v"(bh));
    return acc;
#endif
    asm volatile("s_nop 1\n\t"
                 "v_mfma_f32_16x16x32_bf16 %0, %1, %3, %0\n\t"
                 "v_mfma_f32_16x16x32_bf16 %0, %1, %4, %0\n\t"
                 "v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0"
#ifdef NERFART_EXP_AGPR      // experiment: accumulators in the ACC register file
                 : "+a"(acc) : "v"(ah), "v"(al), "v"(bh), "v"(bl));
#else
                 : "+v"(acc) : "v"(ah), "v"(al), "v"(bh), "v"(bl));
#endif
    return acc;
}

// The item's wait for its A fragments and its three MFMAs as ONE statement (round 3): hipcc pads one wait state after every asm
// statement whose VGPR outputs the next instruction touches, so `s_waitcnt` (outputs = the fragments) + `mfma3` cost an s_nop 0
// between them plus the s_nop 1 inside mfma3 - 3 issue cycles per item that nothing needs: the A operands come from LDS (the wait
// itself), and the B unit was written by VALU instructions at least two LDS reads + the wait earlier (its last `v_cvt_pk` sits in
// item 15 of the previous k-step; PAD adds one state in front of item 0 anyway).  The fragments are plain inputs: named as
// outputs, hipcc pads the next VALU that recycles their registers; that no compiler copy of them sits between the ds_read
// statement and this one is what tools/audit_asm_loads.py (tests/test_asm_audit.py) checks in the ISA.
// FULL (NERFART_F16X2 only; ignored by the split-bf16 build, which always runs three terms): this k-step's B unit is a ready-made
// input unit with a real lo term - a_hi b_hi + a_hi b_lo + a_lo b_hi in fp16; otherwise the 2-MFMA form a_hi b_hi + a_lo b_hi.
template <int CNT, bool PAD, bool FULL = false>
__device__ __forceinline__ f32x4 wait_mfma3(u32x4& ah, u32x4& al, const u32x4 bh, const u32x4 bl, f32x4 acc) {
#if defined(NERFART_ABLATE_MFMA) || defined(NERFART_OLD_ITEM)
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(ah), "+v"(al) : "i"(CNT));
    return mfma3(ah, al, bh, bl, acc);
#elif defined(NERFART_F16X2)
    if constexpr (FULL) {
        if constexpr (PAD) {
            asm volatile("s_waitcnt lgkmcnt(%5)\n\t"
                         "s_nop 0\n\t"
                         "v_mfma_f32_16x16x32_f16 %0, %1, %3, %0\n\t"
                         "v_mfma_f32_16x16x32_f16 %0, %1, %4, %0\n\t"
                         "v_mfma_f32_16x16x32_f16 %0, %2, %3, %0"
                         : "+v"(acc) : "v"(ah), "v"(al), "v"(bh), "v"(bl), "i"(CNT));
        } else {
            asm volatile("s_waitcnt lgkmcnt(%5)\n\t"
                         "v_mfma_f32_16x16x32_f16 %0, %1, %3, %0\n\t"
                         "v_mfma_f32_16x16x32_f16 %0, %1, %4, %0\n\t"
                         "v_mfma_f32_16x16x32_f16 %0, %2, %3, %0"
                         : "+v"(acc) : "v"(ah), "v"(al), "v"(bh), "v"(bl), "i"(CNT));
        }
    }
#ifdef NERFART_F16X1          // ONE term each: a_hi b_hi (the lo fragment of this item is never read from LDS: Items)
    else if constexpr (PAD) {
        asm volatile("s_waitcnt lgkmcnt(%3)\n\t"
                     "s_nop 0\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0"
                     : "+v"(acc) : "v"(ah), "v"(bh), "i"(CNT));
    } else {
        asm volatile("s_waitcnt lgkmcnt(%3)\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0"
                     : "+v"(acc) : "v"(ah), "v"(bh), "i"(CNT));
    }
#else
    else if constexpr (PAD) {
        asm volatile("s_waitcnt lgkmcnt(%4)\n\t"
                     "s_nop 0\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %1, %3, %0\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %2, %3, %0"
                     : "+v"(acc) : "v"(ah), "v"(al), "v"(bh), "i"(CNT));
    } else {
        asm volatile("s_waitcnt lgkmcnt(%4)\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %1, %3, %0\n\t"
                     "v_mfma_f32_16x16x32_f16 %0, %2, %3, %0"
                     : "+v"(acc) : "v"(ah), "v"(al), "v"(bh), "i"(CNT));
    }
#endif
    return acc;
#else
    if constexpr (PAD) {
        asm volatile("s_waitcnt lgkmcnt(%5)\n\t"
                     "s_nop 0\n\t"
                     "v_mfma_f32_16x16x32_bf16 %0, %1, %3, %0\n\t"
                     "v_mfma_f32_16x16x32_bf16 %0, %1, %4, %0\n\t"
                     "v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0"
                     : "+v"(acc) : "v"(ah), "v"(al), "v"(bh), "v"(bl), "i"(CNT));
    } else {
        asm volatile("s_waitcnt lgkmcnt(%5)\n\t"
                     "v_mfma_f32_16x16x32_bf16 %0, %1, %3, %0\n\t"
                     "v_mfma_f32_16x16x32_bf16 %0, %1, %4, %0\n\t"
                     "v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0"
                     : "+v"(acc) : "v"(ah), "v"(al), "v"(bh), "v"(bl), "i"(CNT));
    }
    return acc;
#endif
}

// Two accumulator chains interleaved: the triples of tiles T-1 and T (same B unit) issued alternately, so that no MFMA
// depends on the one just before it.
__device__ __forceinline__ void mfma6(const u32x4 ah0, const u32x4 al0, const u32x4 ah1, const u32x4 al1, const u32x4 bh,
                                      const u32x4 bl, f32x4& a0, f32x4& a1) {
#ifdef NERFART_ABLATE_MFMA
    asm volatile("" :: "v"(ah0), "v"(al0), "v"(ah1), "v"(al1), "v"(bh), "v"(bl));
    return;
#endif
    asm volatile("s_nop 1\n\t"
                 "v_mfma_f32_16x16x32_bf16 %0, %2, %6, %0\n\t"
                 "v_mfma_f32_16x16x32_bf16 %1, %4, %6, %1\n\t"
                 "v_mfma_f32_16x16x32_bf16 %0, %2, %7, %0\n\t"
                 "v_mfma_f32_16x16x32_bf16 %1, %4, %7, %1\n\t"
                 "v_mfma_f32_16x16x32_bf16 %0, %3, %6, %0\n\t"
                 "v_mfma_f32_16x16x32_bf16 %1, %5, %6, %1"
                 : "+v"(a0), "+v"(a1) : "v"(ah0), "v"(al0), "v"(ah1), "v"(al1), "v"(bh), "v"(bl));
}

// Four accumulator chains interleaved (tiles T..T+3 against the same B unit): 12 MFMAs none of which depends on its three
// predecessors, each chain still hh -> hl -> lh.  One wave can keep the matrix pipe busy alone with this block.
__device__ __forceinline__ void mfma12(const u32x4 (&ah)[4], const u32x4 (&al)[4], const u32x4 bh, const u32x4 bl,
                                       f32x4& a0, f32x4& a1, f32x4& a2, f32x4& a3) {
#ifdef NERFART_ABLATE_MFMA
    asm volatile("" :: "v"(ah[0]), "v"(al[0]), "v"(ah[1]), "v"(al[1]), "v"(ah[2]), "v"(al[2]), "v"(ah[3]), "v"(al[3]), "v"(bh), "v"(bl));
    return;
#endif
    asm volatile("s_nop 1\n\t"
                 "v_mfma_f32_16x16x32_bf16 %0, %4, %12, %0\n\t"
                 "v_mfma_f32_16x16x32_bf16 %1, %5, %12, %1\n\t"
                 "v_mfma_f32_16x16x32_bf16 %2, %6, %12, %2\n\t"
                 "v_mfma_f32_16x16x32_bf16 %3, %7, %12, %3\n\t"
                 "v_mfma_f32_16x16x32_bf16 %0, %4, %13, %0\n\t"
                 "v_mfma_f32_16x16x32_bf16 %1, %5, %13, %1\n\t"
                 "v_mfma_f32_16x16x32_bf16 %2, %6, %13, %2\n\t"
                 "v_mfma_f32_16x16x32_bf16 %3, %7, %13, %3\n\t"
                 "v_mfma_f32_16x16x32_bf16 %0, %8, %12, %0\n\t"
                 "v_mfma_f32_16x16x32_bf16 %1, %9, %12, %1\n\t"
                 "v_mfma_f32_16x16x32_bf16 %2, %10, %12, %2\n\t"
                 "v_mfma_f32_16x16x32_bf16 %3, %11, %12, %3"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                 : "v"(ah[0]), "v"(ah[1]), "v"(ah[2]), "v"(ah[3]), "v"(al[0]), "v"(al[1]), "v"(al[2]), "v"(al[3]), "v"(bh), "v"(bl));
}

// ---------------------------------------------------------------------------------------
// Weight stream: chunk c is consumed from LDS buffer pb while chunk c+1 is streamed into buffer pb^1 in
// 1 KiB pieces issued BETWEEN the MFMAs of chunk c (stream_piece), not in a burst after the barrier.
// ---------------------------------------------------------------------------------------
struct Stream {
    const float* blob;
    const int* tab;            // LDS copy of the chunk offset table
    float* lds;
    int nc;
    const float* iss_src;      // this wave's 8 KiB share of the chunk being streamed during the current chunk
    unsigned iss_dst;          // LDS byte address of that share
    unsigned voff_a, voff_b;   // per-lane byte offsets of pieces 0..3 / 4..7 (lane * 16, + 4096)
    int nxt, nxt_o0;           // the chunk to stream during the NEXT chunk (offset looked up one acquire ahead)
    int pb;
    bool wrap;                 // another tile follows: chunk 0 comes after chunk nc-1
    int late_st;               // VMEM stores issued after the chunk's last LDS-DMA piece: they may stay in flight across the acquire
    unsigned long long lo_exec;   // NERFART_F16X1: EXEC mask of the odd (lo fragment) pieces of the chunk being streamed: all lanes or none - see lo_needed
};

// NERFART_F16X1 (K2 on surface program 3, chunks 0..29 of pack_blob.hip::chunk_desc): wave w streams items 4w..4w+3 of a chunk, i.e. four (hi, lo)
// fragment pairs of k-step w >> 2 of the chunk - even pieces hi, odd pieces lo.  The lo fragments are read only on the k-steps of the ready-made
// input units (three-term form): both k-steps of chunk 0 (layer 0), k-step 1 of chunk 16 and k-step 0 of chunk 17 (the skip layer's two encoding
// units).  Everywhere else the odd pieces are not issued: half the LDS-DMA instructions and half the L2 -> LDS bytes of the weight stream.
__device__ __forceinline__ bool lo_needed(int chunk) {
    const unsigned mask = (wave_id() >> 2) ? ((1u << 0) | (1u << 16)) : ((1u << 0) | (1u << 17));
    return chunk >= 0 && chunk < 32 && ((mask >> chunk) & 1u);
}

__device__ __forceinline__ void stream_lookup(Stream& s, int chunk) {
    s.nxt = chunk;
    // no next chunk: stream chunk 0 again (never read) - keeps stream_piece branch free
    s.nxt_o0 = __builtin_amdgcn_readfirstlane(s.tab[chunk >= 0 ? chunk : 0]);
}
__device__ __forceinline__ int stream_next_of(const Stream& s, int c) {
    if (c < 0) return -1;
    return (c + 1 == s.nc) ? (s.wrap ? 0 : -1) : c + 1;
}
// Piece J (1 KiB) of this wave's share.  Wave w streams bytes [8192 w, 8192 w + 8192) of the chunk - for a
// one-k-step chunk (32 KiB) waves 4..7 copy the bytes that follow it in the blob into the unused half of the
// buffer (the last chunk of a program is always a full one, packing.py).  M0 = LDS destination base, written in
// the same statement that uses it (cdna_hip_programming.md 5.7); the instruction's immediate offset applies to
// BOTH the global address and the LDS address.
template <int J>
__device__ __forceinline__ void stream_piece(const Stream& s) {
#ifdef NERFART_F16X1
    if constexpr (J & 1) {
#ifdef NERFART_ABLATE_DMA        // timing experiments only
        return;
#endif
        // a lo piece: issued under EXEC = lo_exec (all lanes, or none where the chunk's k-step never reads its lo fragments) - no branch inside the
        // item stream (the counted lgkmcnt windows are straight-line code, tools/audit_asm_loads.py); an instruction with EXEC = 0 moves nothing
        unsigned keep_m0;
        unsigned long long keep_exec;
        asm volatile("s_mov_b32 %0, m0\n\t"
                     "s_add_u32 m0, %4, %5\n\t"
                     "s_mov_b64 %1, exec\n\t"
                     "s_mov_b64 exec, %7\n\t"
                     "global_load_lds_dwordx4 %2, %3 offset:%6\n\t"
                     "s_mov_b64 exec, %1\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep_m0), "=&s"(keep_exec)
                     : "v"(J < 4 ? s.voff_a : s.voff_b), "s"(s.iss_src), "s"(s.iss_dst), "i"((J & 4) * 1024), "i"((J & 3) * 1024), "s"(s.lo_exec)
                     : "memory", "scc");
        return;
    }
#endif
#ifndef NERFART_ABLATE_DMA
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_add_u32 m0, %3, %4\n\t"
                 "s_nop 0\n\t"
#ifdef NERFART_EXP_DMA_NT     // experiment: non-temporal hint on the weight stream
                 "global_load_lds_dwordx4 %1, %2 offset:%5 nt\n\t"
#else
                 "global_load_lds_dwordx4 %1, %2 offset:%5\n\t"
#endif
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(J < 4 ? s.voff_a : s.voff_b), "s"(s.iss_src), "s"(s.iss_dst), "i"((J & 4) * 1024), "i"((J & 3) * 1024)
                 : "memory", "scc");
#endif
}
__device__ __forceinline__ void stream_target(Stream& s, int o0, int buf) {
    const int w = wave_id();
    s.iss_src = s.blob + o0 + w * 2048;
    s.iss_dst = lds_addr(s.lds + buf * CHUNK_FLOATS) + w * 8192;
}
// once per workgroup, after the table is in LDS: chunk 0 in a burst into buffer 0
__device__ __forceinline__ void stream_start(Stream& s) {
    s.voff_a = lane_id() * 16;
    s.voff_b = lane_id() * 16 + 4096;
    stream_lookup(s, 0);
    stream_target(s, s.nxt_o0, 0);
    s.lo_exec = ~0ull;
    stream_piece<0>(s); stream_piece<1>(s); stream_piece<2>(s); stream_piece<3>(s);
    stream_piece<4>(s); stream_piece<5>(s); stream_piece<6>(s); stream_piece<7>(s);
    s.pb = 0;
    stream_lookup(s, stream_next_of(s, 0));
}
__device__ __forceinline__ const float* stream_acquire(Stream& s) {
#ifndef NERFART_ABLATE_VMWAIT   // timing experiments only (tools/ablate_bf16.py): results are wrong without these
    // my pieces of the current chunk have landed.  vmcnt counts loads and stores alike and retires in issue order: a unit store
    // issued AFTER the chunk's last piece (Items, LATE_ST) may stay in flight - its acknowledgement from a missing L2 line
    // takes longer than the rest of the chunk
    if (s.late_st) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else wait_glds();
    s.late_st = 0;
#endif
#ifndef NERFART_ABLATE_BARRIER
    __syncthreads();      // everyone's pieces landed; everyone is done reading buffer pb^1
#endif
    const float* w = s.lds + s.pb * CHUNK_FLOATS;
    stream_target(s, s.nxt_o0, s.pb ^ 1);
#ifdef NERFART_F16X1
    s.lo_exec = __builtin_amdgcn_readfirstlane((int)lo_needed(s.nxt)) != 0 ? ~0ull : 0ull;
#endif
    stream_lookup(s, stream_next_of(s, s.nxt));
    s.pb ^= 1;
    return w;
}

// ---------------------------------------------------------------------------------------
// Epilogue slices.  One pair of values goes through three slices: PH 0 (exp, rcp / decode) -> PH 1 (log, select /
// multiply) -> PH 2 (split to bf16 hi/lo).  MODE:
//   0 softplus(beta = 100)                         3 reverse sweep: z * softplus'(z_l), softplus' from the unorm16 pair `din`
//   1 softplus value / tangent columns (quads)     4 reverse sweep, first step: softplus'(z) itself
//   2 max(z, floor)                                5 forward sweep of the reverse-mode kernel: softplus + `dout` =
//   8 = 2, the unit's bf16 hi part is dumped         softplus'(z) of the pair as packed unorm16
//   7 radiance backward: z * [r > 0], r = the pair of bf16 activations `din` dumped by mode 8; 9: z itself
//     (7 and 9 dump the unit's hi part: the deltas of the weight-gradient GEMMs)
//  10 second-order SDF backward, forward sweep: column PAIRS (value, tangent along a given direction): like 1 with
//     the softplus' taken from the pair's even lane; dumps the unit's hi part and softplus' (unorm16)
//  11 second-order SDF backward, reverse sweep: column pairs (t = d sdf/d a, abar = d loss/d a):
//     even lanes z D, odd lanes z D + 100 t tangent (65535 - D), D = 65535 softplus' (`din`), tangent = `din2` (bf16)
// ---------------------------------------------------------------------------------------
// max(z, 0) in one instruction (fmaxf() first canonicalises z with a v_max_f32 z, z)
__device__ __forceinline__ float relu1(float z) {
#ifdef NERFART_OLD_ITEM
    float y;
    asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(z));
    return y;
#else
    float y;
    asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(z));
    return y;
#endif
}
// max(z, 0) + log2(1 + e) * ln 2 / 100 from the v_log_f32 result L: the v_max and the v_fmac as one statement - as two, hipcc
// pads an s_nop between the asm v_max and the v_fmac that consumes it (round 3).  Same two instructions, same bits.
__device__ __forceinline__ float softplus_tail(float z, float L) {
#ifdef NERFART_OLD_ITEM
    return relu1(z) + L * (0.69314718055994530942f / 100.0f);
#elif defined(NERFART_F16X1)     // scaled recursion: max(z', 0) + log2(1 + e)
    float y;
    asm("v_max_f32 %0, 0, %1\n\tv_add_f32 %0, %0, %2" : "=&v"(y) : "v"(z), "v"(L));
    return y;
#else
    float y;
    asm("v_max_f32 %0, 0, %1\n\tv_fmac_f32 %0, 0x3be32166, %2" : "=&v"(y) : "v"(z), "v"(L));
    return y;
#endif
}

// lane 2i of every pair (2i, 2i+1) to both: one VALU op with DPP quad_perm [0,0,2,2]
__device__ __forceinline__ float pair_bcast0(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xA0, 0xF, 0xF, false));
}

template <int MODE, int PH>
__device__ __forceinline__ void epi_phase(float z0, float z1, Work& w, unsigned& hi, unsigned& lo, float floor, bool is_val,
                                          unsigned din, unsigned& dout, unsigned din2 = 0u) {
#ifdef NERFART_ABLATE_EPI       // timing experiments only: no activation arithmetic
    if (PH == 2) { hi = __float_as_uint(z0); lo = __float_as_uint(z1); }
    return;
#endif
    if constexpr (PH == 0) {
        if constexpr (MODE == 4) {
            // first backward step of the reverse-mode kernel: softplus'(z_7) from the same z_7 the last forward epilogue just took
            // exp2(-|100 z|) of - hipcc merged the two and carried the 64 exponentials per lane across (spill store in the forward
            // epilogue, scratch reload + vmcnt(0) inside the backward item stream).  Opaque copies: recompute, two VALU per pair.
            asm volatile("" : "+v"(z0), "+v"(z1));
        }
        if constexpr (MODE == 0 || MODE == 1 || MODE == 4 || MODE == 5 || MODE == 10) {
#ifdef NERFART_F16X1
            // SCALED recursion (precision 5; the blob comes from nerfart_amd/calibrate.py): the accumulators hold z' = c z, c = 100 log2 e - layer 0's weights,
            // the skip layer's encoding columns and every bias carry c - and the activation is a' = c softplus(z) = max(z', 0) + log2(1 + 2^-|z'|): the
            // hidden layers' weights are the network's own (c a is what they multiply), the last row carries 1 / c.  One multiply per value less in an
            // epilogue that costs the 1-MFMA kernel more than its matrix work.
            w.e0 = __builtin_amdgcn_exp2f(-fabsf(z0));
            w.e1 = __builtin_amdgcn_exp2f(-fabsf(z1));
#else
            w.e0 = __builtin_amdgcn_exp2f(fabsf(z0) * -144.269504088896340736f);     // exp(-|100 z|)
            w.e1 = __builtin_amdgcn_exp2f(fabsf(z1) * -144.269504088896340736f);
#endif
        }
        if constexpr (MODE == 11) {
            w.r0 = (float)(din & 0xffffu);
            w.r1 = (float)(din >> 16);
            w.e0 = __uint_as_float(din2 << 16);     // tangent of the activation (bf16 hi part from the forward sweep)
            w.e1 = __uint_as_float(din2 & 0xffff0000u);
        }
        if constexpr (MODE == 1 || MODE == 4 || MODE == 5 || MODE == 10) {
            w.r0 = __builtin_amdgcn_rcpf(1.0f + w.e0);
            w.r1 = __builtin_amdgcn_rcpf(1.0f + w.e1);
        }
        if constexpr (MODE == 3) {
            w.r0 = (float)(din & 0xffffu);          // 65535 * softplus'(z_l): the 1/65535 lives in the packed weights
            w.r1 = (float)(din >> 16);
#ifdef NERFART_F16X2                                   // ... except in fp16: z * 65535 overflows it, w / 65535 underflows it
            w.r0 *= (1.0f / 65535.0f);
            w.r1 *= (1.0f / 65535.0f);
#endif
        }
        if constexpr (MODE == 7) {
            w.r0 = (din & 0xffffu) ? 1.f : 0.f;     // relu mask from the dumped activation (bf16 bits)
            w.r1 = (din >> 16) ? 1.f : 0.f;
        }
    } else if constexpr (PH == 1) {
        if constexpr (MODE == 2 || MODE == 8) {
            w.y0 = fmaxf(z0, floor);
            w.y1 = fmaxf(z1, floor);
        } else if constexpr (MODE == 9) {
            w.y0 = z0;
            w.y1 = z1;
        } else if constexpr (MODE == 3 || MODE == 7) {
            w.y0 = z0 * w.r0;
            w.y1 = z1 * w.r1;
        } else if constexpr (MODE == 4) {
            w.y0 = (z0 >= 0.f) ? w.r0 : w.e0 * w.r0;
            w.y1 = (z1 >= 0.f) ? w.r1 : w.e1 * w.r1;
        } else if constexpr (MODE == 11) {
            const float t0 = pair_bcast0(z0), t1 = pair_bcast0(z1);                  // the pair's even lane: d sdf / d a
            // arithmetic mask instead of a select: hipcc turns the select into a divergent branch around the extra term
            const float odd = is_val ? 0.f : 100.0f;
            w.y0 = fmaf(odd * t0 * w.e0, 65535.0f - w.r0, z0 * w.r0);
            w.y1 = fmaf(odd * t1 * w.e1, 65535.0f - w.r1, z1 * w.r1);
        } else if constexpr (MODE == 10) {
            const float v0 = softplus_tail(z0, __builtin_amdgcn_logf(1.0f + w.e0));
            const float v1 = softplus_tail(z1, __builtin_amdgcn_logf(1.0f + w.e1));
            const float d0 = pair_bcast0((z0 >= 0.f) ? w.r0 : w.e0 * w.r0);
            const float d1 = pair_bcast0((z1 >= 0.f) ? w.r1 : w.e1 * w.r1);
            w.y0 = is_val ? v0 : d0 * z0;
            w.y1 = is_val ? v1 : d1 * z1;
            typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
            dout = __builtin_bit_cast(unsigned, (u16x2)__builtin_amdgcn_cvt_pknorm_u16(d0, d1));
        } else {
            const float v0 = softplus_tail(z0, __builtin_amdgcn_logf(1.0f + w.e0));
            const float v1 = softplus_tail(z1, __builtin_amdgcn_logf(1.0f + w.e1));
            if constexpr (MODE == 1) {
                // value lanes carry z (bias included); tangent lanes carry dz and take softplus'(z) from their quad's lane 0
                const float d0 = quad_bcast0((z0 >= 0.f) ? w.r0 : w.e0 * w.r0);
                const float d1 = quad_bcast0((z1 >= 0.f) ? w.r1 : w.e1 * w.r1);
                w.y0 = is_val ? v0 : d0 * z0;
                w.y1 = is_val ? v1 : d1 * z1;
            } else {
                w.y0 = v0;
                w.y1 = v1;
            }
            if constexpr (MODE == 5) {
                const float d0 = (z0 >= 0.f) ? w.r0 : w.e0 * w.r0;
                const float d1 = (z1 >= 0.f) ? w.r1 : w.e1 * w.r1;
                typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
                dout = __builtin_bit_cast(unsigned, (u16x2)__builtin_amdgcn_cvt_pknorm_u16(d0, d1));
            }
        }
    } else {
        split2(w.y0, w.y1, hi, lo);
    }
}

// ---------------------------------------------------------------------------------------
// Reverse-mode kernel only: softplus'(z_l) travels from the forward to the backward sweep through a per-workgroup
// scratch in global memory (stays in L2 / Infinity Cache): unit u of layer l = 16 B per lane (4 pairs of unorm16) at
// ws + ((8 l + u) * 8 + wave) * 1024 + lane * 16.
// ---------------------------------------------------------------------------------------
// Unit `idx` = 8 * slot + unit of a dump lives at base + slot * slot_stride + unit * unit_stride + (per-lane) voff:
//   scratch of k_sdf_grad_bf16 (never leaves the kernel): wave-private 1 KiB pieces - unit_stride 8192, slot_stride 65536,
//     voff = lane * 16, base includes wave * 1024;
//   dumps that feed the weight-gradient GEMMs: POINT-MAJOR matrices [slot][row][256 features in unit order] (bf16) so that the
//     GEMMs read them in place - unit_stride 64, slot_stride = rows * 512, voff = row * 512 + lane_group * 16.
struct GradCtx {
    char* ws;                 // base the units are loaded from (wave uniform)
    char* ws_out;             // base finished units are stored to
    size_t slot_stride;
    unsigned unit_stride;
    unsigned voff;            // per-lane byte offset of loads
    unsigned voff2 = 0;       // second-order reverse sweep: ... of the tangent loads (the softplus' loads use voff)
    unsigned voff_out;        // ... of stores
    unsigned voff_st2 = 0;    // second-order forward sweep: ... of the softplus' stores: voff_out on value lanes, out of range on
    unsigned st2_bytes = 0;   //     tangent lanes (a buffer store drops them: no branch inside the item stream); bytes of a slot's value rows
    int layer;                // layer whose weights are being applied (wave uniform)
    u32x4 dbuf[2];            // backward sweep: softplus' units, k-step parity double buffer
    u32x4 abuf[2];            // second-order reverse sweep: tangent units (bf16 hi parts)
    u32x4 dacc2, dpend2;      // second-order forward sweep: the softplus' unit stored next to the hi unit (slot + 8)
    u32x4 dacc;               // forward sweep: unit being packed
    u32x4 dpend;              // forward sweep: finished unit, stored at the first triple of the next k-step
    char* pend_ptr;
    // k_sdf_grad_bf16 (round 3): its scratch is addressed as a buffer - descriptor (uniform) + ONE 32-bit voffset register (wave *
    // 1024 + lane * 16 + the unit's byte offset) - instead of a 64-bit per-lane pointer per unit: hipcc hoisted the 56 unit
    // addresses out of the tile loop (112 VGPRs) and spilled them, ~220 scratch reloads per lane per tile.  The
    // unit offset goes into the vector register, not into the scalar soffset: see the note at mlp_chain.hip's Scratch.
    bool buf = false;
    __amdgpu_buffer_rsrc_t rsrc;
    unsigned pend_soff;
};
__device__ __forceinline__ u32x4 unit_load(const GradCtx& gc, int idx) {
    if (gc.buf) return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(gc.rsrc, (int)(gc.voff + (unsigned)((idx >> 3) * gc.slot_stride + (idx & 7) * gc.unit_stride)), 0, 0));
    return *reinterpret_cast<const u32x4*>(gc.ws + (size_t)(idx >> 3) * gc.slot_stride + (size_t)(idx & 7) * gc.unit_stride + gc.voff);
}
// k_radiance_bwd_bf16's sweep position `layer` (4: R3^T .. 1: R0^T, 0: W8^T) -> slot of the delta that sweep step produces
__device__ __forceinline__ int rad_delta_slot(int layer) { return layer > 0 ? layer - 1 : 4; }
__device__ __forceinline__ size_t uoff(const GradCtx& gc, int idx) {
    return (size_t)(idx >> 3) * gc.slot_stride + (size_t)(idx & 7) * gc.unit_stride;
}
// second-order forward sweep: softplus'(z) is the same number on both lanes of a (value, tangent) pair - only the value rows of
// slots 8..15 hold it.  `base` = the unit's address in row 0 (wave uniform); tangent lanes carry an out-of-range offset
__device__ __forceinline__ void st2_store(const GradCtx& gc, char* base, const u32x4 v) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, gc.st2_bytes, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)gc.voff_st2, 0, 0);
}
// (l, unit) of the second-order kernels: softplus' at slot 8 + l, tangent / activation hi parts at slot l
__device__ __forceinline__ void d_load2(GradCtx& gc, int idx, int buf) {
#ifdef NERFART_ABLATE_SCRATCH      // timing experiments only (tools/ablate_grad.py): no scratch / dump traffic, results wrong
    return;
#endif
    gc.dbuf[buf] = *reinterpret_cast<const u32x4*>(gc.ws + uoff(gc, 64 + idx) + gc.voff);
    gc.abuf[buf] = *reinterpret_cast<const u32x4*>(gc.ws + uoff(gc, idx) + gc.voff2);
}
// Plain (compiler-visible) loads: hipcc then keeps its own vmcnt bookkeeping for them - with the LDS-DMA pieces it
// cannot see this can only make its wait stricter.  (Hand-counted asm loads gave wrong gradients on random waves.)
__device__ __forceinline__ void d_load(GradCtx& gc, int idx, int buf) {
#if defined(NERFART_ABLATE_SCRATCH) || defined(NERFART_ABLATE_SCRATCH_LD)
    return;
#endif
#if defined(NERFART_EXP_SCRATCH_NT) && (NERFART_EXP_SCRATCH_NT & 2)
    gc.dbuf[buf] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(gc.ws + uoff(gc, idx) + gc.voff));
#else
    gc.dbuf[buf] = unit_load(gc, idx);
#endif
}

// ---------------------------------------------------------------------------------------
// The work of one weight chunk, one "item" = (k-step, output tile) at a time: wait for the item's A fragments,
// 3 MFMAs, then the fillers: the fragment reads two items ahead, an LDS-DMA piece (items 5..8 of a k-step),
// one epilogue slice (last 12 items of a hosting k-step).  LDS returns in order: with items it, it+1, it+2
// outstanding (2 reads each) item it has landed at lgkmcnt(4) (cdna_hip_programming.md 5.7, form ii).
// ---------------------------------------------------------------------------------------
#ifndef NERFART_AHEAD        // A-fragment prefetch distance in items (2; 3 measured slower in round 1, re-measured by tools/ablate_bf16.py)
#define NERFART_AHEAD 2
#endif
template <int NS> struct RingT { u32x4 h[NS], l[NS]; };   // AHEAD + 1 slots (+ RING_EXTRA)
#if defined(NERFART_EXP_PAIR) || defined(NERFART_EXP_SEG4)   // experiments: tiles multiplied in pairs (mfma6) / fours (mfma12)
constexpr int RING_EXTRA = 1;
#else
constexpr int RING_EXTRA = 0;
#endif
struct Ring3 { u32x4 h0, l0, h1, l1, h2, l2; };     // fixed 3-slot ring of the reverse-mode tail

template <int OFF>
__device__ __forceinline__ void lds_read_pair(u32x4& fh, u32x4& fl, unsigned addr) {
#ifdef NERFART_ABLATE_LDSREAD    // timing experiments only: fragments are whatever the registers held
    asm volatile("; no read %0 %1 %2" : "=&v"(fh), "=&v"(fl) : "v"(addr));
    return;
#endif
    asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"
                 : "=&v"(fh), "=&v"(fl) : "v"(addr), "i"(OFF), "i"(OFF + 1024));
}
template <int OFF>
__device__ __forceinline__ void lds_read_hi(u32x4& fh, unsigned addr) {      // NERFART_F16X1: the item's hi fragment alone
#ifdef NERFART_ABLATE_LDSREAD    // timing experiments only
    asm volatile("; no read %0 %1" : "=&v"(fh) : "v"(addr));
    return;
#endif
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(fh) : "v"(addr), "i"(OFF));
}
template <int CNT>
__device__ __forceinline__ void lds_wait_pair(u32x4& fh, u32x4& fl) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fh), "+v"(fl) : "i"(CNT));
}
template <int NS>
__device__ __forceinline__ void lds_wait_ring4(RingT<NS>& r) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.h[0]), "+v"(r.l[0]), "+v"(r.h[1]), "+v"(r.l[1]), "+v"(r.h[2]), "+v"(r.l[2]), "+v"(r.h[3]), "+v"(r.l[3]));
}
template <int CNT>
__device__ __forceinline__ void lds_wait_pair2(u32x4& fh0, u32x4& fl0, u32x4& fh1, u32x4& fl1) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fh0), "+v"(fl0), "+v"(fh1), "+v"(fl1) : "i"(CNT));
}

// Layer shape: NH input units come from the previous layer's accumulators P (k-steps 0..NH-1, built just in
// time with epilogue MODE), NX are ready-made units xs[] (k-steps NH..NH+NX-1: encodings, extras, activations
// from memory).  NEXT0: the last k-step also builds unit 0 of the NEXT layer from this layer's tiles 0 and 1, with
// epilogue MQ.  ZINIT: accumulators start at 0 instead of the bias.  PEND_IN / LOADNEXT (reverse-mode kernel): a
// softplus' unit of the previous layer is waiting to be stored / the following layer's first unit needs its load.
template <int MODE_, int MQ_, int NH_, int NX_, bool NEXT0_, bool ZINIT_ = false, bool PEND_IN_ = false, bool LOADNEXT_ = false,
          int AHEAD_ = NERFART_AHEAD>
struct Cfg {
    static constexpr int MODE = MODE_, MQ = MQ_, NH = NH_, NX = NX_, NXA = NX_ > 0 ? NX_ : 1, NKS = NH_ + NX_;
    static constexpr int AHEAD = AHEAD_;      // A fragments are read this many items ahead of their MFMAs (2 or 3)
    static constexpr bool NEXT0 = NEXT0_, ZINIT = ZINIT_, PEND_IN = PEND_IN_, LOADNEXT = LOADNEXT_;
    // unit whose epilogue is hosted by k-step ks: a unit of act(P) (0..7), 100 = unit 0 of act(Q), -1 = none
    static constexpr int hosted(int ks) { return (ks + 1 < NH_) ? ks + 1 : ((NEXT0_ && ks == NH_ + NX_ - 1) ? 100 : -1); }
    static constexpr int mode_of(int hu) { return hu == 100 ? MQ_ : MODE_; }
    static constexpr bool stores(int m) { return m == 5 || m == 7 || m == 8 || m == 9 || m == 10 || m == 11; }   // unit dumped when finished
    static constexpr bool loads(int m) { return m == 3 || m == 7 || m == 11; }               // unit needs a `din` unit
};

template <class L, int C, int NKC, int IT>
struct Items {
    // NERFART_F16X1: does item `it` of this chunk belong to a k-step of ready-made input units (hi + lo fragments, three terms)?
    static constexpr bool item_full(int it) { return (CHUNK_KS * C + (it >> 4)) >= L::NH; }
    static constexpr int pending_after(int it, int n) {
        int p = 0;
        for (int i = 1; i <= L::AHEAD && it + i < n; ++i) p += item_full(it + i) ? 2 : 1;
        return p;
    }
    static __device__ __forceinline__ void run(const Acc& P, Acc& Q, Unit (&xb)[2], const Unit (&xs)[L::NXA], Unit& x0n, Work2& w,
                                               RingT<L::AHEAD + 1 + RING_EXTRA>& r, unsigned addr, const Stream& s, const EpiCtx& ec, GradCtx& gc) {
        constexpr int N = NKC * 16;
        if constexpr (IT < N) {
            constexpr int kk = IT >> 4, T = IT & 15, ks = CHUNK_KS * C + kk;
            constexpr int AH = L::AHEAD, NS = AH + 1 + RING_EXTRA;
            constexpr int S = IT % NS, S2 = (IT + AH) % NS;
            constexpr int LEFT = N - 1 - IT;
#ifdef NERFART_F16X1
            // one fragment read per item on the k-steps whose input unit is built from accumulators (1 MFMA: a_hi b_hi), two on the ready-made
            // units' (three-term form): the counted wait of item IT leaves the reads of the items behind it in flight
            constexpr int PENDING = pending_after(IT, N);
            if constexpr (IT + AH < N) {
                if constexpr (item_full(IT + AH)) lds_read_pair<(IT + AH) * 2048>(r.h[S2], r.l[S2], addr);
                else lds_read_hi<(IT + AH) * 2048>(r.h[S2], addr);
            }
#else
            constexpr int PENDING = 2 * (LEFT < AH ? LEFT : AH);
#ifndef NERFART_EXP_SEG4
            if constexpr (IT + AH < N) lds_read_pair<(IT + AH) * 2048>(r.h[S2], r.l[S2], addr);
#endif
#endif
            u32x4 bh, bl;
            if constexpr (ks < L::NH) { bh = xb[ks & 1].h; bl = xb[ks & 1].l; }
            else { bh = xs[ks - L::NH].h; bl = xs[ks - L::NH].l; }
#if defined(NERFART_EXP_SEG4)
            // segmented stream: items 4s..4s+3 are multiplied as ONE block of 12 independent MFMAs at item 4s, the fragments
            // of the next four items are requested right behind it, and the fillers of the four items follow
            static_assert(NS == 4, "segments of four items need a four-slot ring");
            if constexpr ((T & 3) == 0) {
                lds_wait_ring4(r);
                mfma12(r.h, r.l, bh, bl, Q.t[T], Q.t[T + 1], Q.t[T + 2], Q.t[T + 3]);
                if constexpr (IT + 4 < N) {
                    lds_read_pair<(IT + 4) * 2048>(r.h[0], r.l[0], addr);
                    lds_read_pair<(IT + 5) * 2048>(r.h[1], r.l[1], addr);
                    lds_read_pair<(IT + 6) * 2048>(r.h[2], r.l[2], addr);
                    lds_read_pair<(IT + 7) * 2048>(r.h[3], r.l[3], addr);
                }
            }
#elif defined(NERFART_EXP_PAIR)
            if constexpr ((T & 1) != 0) {
                constexpr int SP = (IT - 1) % NS;
                lds_wait_pair2<PENDING>(r.h[SP], r.l[SP], r.h[S], r.l[S]);
                mfma6(r.h[SP], r.l[SP], r.h[S], r.l[S], bh, bl, Q.t[T - 1], Q.t[T]);
            }
#else
            Q.t[T] = wait_mfma3<PENDING, T == 0, (ks >= L::NH)>(r.h[S], r.l[S], bh, bl, Q.t[T]);
#endif
            constexpr int HU = L::hosted(ks);
            constexpr int HM = L::mode_of(HU);
#ifdef NERFART_EXP_ST_LATE        // experiment: the unit store behind the k-step's LDS-DMA pieces (item 9), left in flight across the acquire
            constexpr int ST_ITEM = 9;
#else
            constexpr int ST_ITEM = 0;
#endif
            if constexpr (T == ST_ITEM) {
                // reverse-mode kernel, forward sweep: store the softplus' unit finished in the previous k-step
                constexpr int HUP = (ks == 0) ? (L::PEND_IN ? 100 : -1) : L::hosted(ks - 1);
                constexpr bool STORE = (ks == 0) ? L::PEND_IN : (HUP >= 0 && L::stores(L::mode_of(HUP)));
#if defined(NERFART_ABLATE_SCRATCH) || defined(NERFART_ABLATE_SCRATCH_ST)
                if constexpr (false) {
#else
                if constexpr (STORE) {
#endif
#ifdef NERFART_EXP_ST_LATE
                    constexpr int PM0 = (ks == 0) ? L::MODE : L::mode_of(HUP);
                    if constexpr (PM0 != 10 && kk == NKC - 1) const_cast<Stream&>(s).late_st = 1;
#endif
#if defined(NERFART_EXP_SCRATCH_NT) && (NERFART_EXP_SCRATCH_NT & 1)
                    __builtin_nontemporal_store(gc.dpend, reinterpret_cast<u32x4*>(gc.pend_ptr + gc.voff_out));
#else
                    if (gc.buf) __builtin_amdgcn_raw_buffer_store_b128(gc.dpend, gc.rsrc, (int)(gc.voff_out + gc.pend_soff), 0, 0);
                    else *reinterpret_cast<u32x4*>(gc.pend_ptr + gc.voff_out) = gc.dpend;
#endif
                    constexpr int PM = (ks == 0) ? L::MODE : L::mode_of(HUP);        // (mode 10 layers use 10 for both kinds)
                    if constexpr (PM == 10) st2_store(gc, gc.pend_ptr + 8 * gc.slot_stride, gc.dpend2);
                }
            }
            if constexpr (T == 0) {
                // backward sweep: load the softplus' unit needed by the slices of the NEXT k-step
                if constexpr (ks + 1 < L::NKS) {
                    constexpr int HN = L::hosted(ks + 1);
                    if constexpr (HN >= 0 && L::loads(L::mode_of(HN))) {
                        if constexpr (L::mode_of(HN) == 11) {
                            if constexpr (HN == 100) d_load2(gc, (gc.layer - 1) * 8, (ks + 1) & 1);
                            else d_load2(gc, gc.layer * 8 + HN, (ks + 1) & 1);
                        } else {
                            if constexpr (HN == 100) d_load(gc, (gc.layer - 1) * 8, (ks + 1) & 1);
                            else d_load(gc, gc.layer * 8 + HN, (ks + 1) & 1);
                        }
                    }
                } else if constexpr (L::LOADNEXT) {
                    if constexpr (L::MODE == 11) d_load2(gc, (gc.layer - 1) * 8 + 1, (ks + 1) & 1);
                    else d_load(gc, (gc.layer - 1) * 8 + 1, (ks + 1) & 1);
                }
            }
            // LDS-DMA: the 8 pieces per wave of the next chunk go out during items 5..8 of each k-step (after the
            // first use of a softplus' unit in item 4: the compiler's wait for that load drains the VMEM queue)
#if defined(NERFART_EXP_DMA_SPREAD)          // experiment: one piece every 4th item instead of 4 in a row
            constexpr bool DMA_HERE = (T % 4) == NERFART_EXP_DMA_SPREAD;
            constexpr int DMA_J = T / 4;
#elif defined(NERFART_EXP_DMA_T0)            // experiment: the 4 pieces of a k-step start at another item
            constexpr bool DMA_HERE = T >= NERFART_EXP_DMA_T0 && T < NERFART_EXP_DMA_T0 + 4;
            constexpr int DMA_J = T - NERFART_EXP_DMA_T0;
#else
            // plain softplus layers (no unit loads / stores on the VMEM queue): items 0..3, which host no epilogue slice
            // (-1.5 % on k_sdf_only_bf16); elsewhere after item 4, see above
            constexpr int DMA_T0 = (L::MODE == 0 && L::MQ == 0) ? 0 : 5;
            constexpr bool DMA_HERE = T >= DMA_T0 && T < DMA_T0 + 4;
            constexpr int DMA_J = T - DMA_T0;
#endif
#ifdef NERFART_EXP_DMA_EARLY      // experiment: all 8 pieces in the first k-step of a two-k-step chunk (items 0..7)
            if constexpr (NKC == 2) {
                if constexpr (kk == 0 && T < 8) stream_piece<T>(s);
            } else
#endif
#ifdef NERFART_F16X1
            // the odd pieces are lo fragments (stream_piece): where the chunk being streamed - chunk C + 1 of this layer, or the next layer's first chunk,
            // which in K2 holds ready-made input units only after the LAST layer (NEXT0 false: layer 0 of the next tile) - has no k-step that reads them, they
            // are not even issued under EXEC = 0: 27 of K2's 30 chunks
            constexpr int NA = CHUNK_KS * (C + 1), NB = NA + 1;
            constexpr bool LO_NEXT = (NA < L::NKS) ? (NA >= L::NH || (NB < L::NKS && NB >= L::NH)) : !L::NEXT0;
#else
            constexpr bool LO_NEXT = true;
#endif
            if constexpr (DMA_HERE) {
                if constexpr (NKC == 2) { if constexpr (LO_NEXT || ((kk * 4 + DMA_J) & 1) == 0) stream_piece<kk * 4 + DMA_J>(s); }
                else { stream_piece<2 * DMA_J>(s); if constexpr (LO_NEXT) stream_piece<2 * DMA_J + 1>(s); }
            }
            // epilogue slice hosted by this item
#ifdef NERFART_EXP_EPI2      // experiment: plain softplus layers run TWO independent pairs per slice (items 4..9), ILP 2
            constexpr bool EPI2 = (HU >= 0) && (HM == 0);
#else
            constexpr bool EPI2 = false;
#endif
            if constexpr (EPI2) {
                if constexpr (T >= 4 && T < 10) {
                    constexpr int ph = (T - 4) % 3, pq = (T - 4) / 3;
                    constexpr int tile = (HU == 100 ? 0 : 2 * HU) + pq;
                    unsigned hiA = 0, loA = 0, hiB = 0, loB = 0, dout = 0;
                    if constexpr (HU == 100) {
                        epi_phase<0, ph>(Q.t[tile][0], Q.t[tile][1], w, hiA, loA, ec.floor_q, ec.is_val, 0u, dout);
                        epi_phase<0, ph>(Q.t[tile][2], Q.t[tile][3], w.b, hiB, loB, ec.floor_q, ec.is_val, 0u, dout);
                        if constexpr (ph == 2) { x0n.h[2 * pq] = hiA; x0n.l[2 * pq] = loA; x0n.h[2 * pq + 1] = hiB; x0n.l[2 * pq + 1] = loB; }
                    } else {
                        epi_phase<0, ph>(P.t[tile][0], P.t[tile][1], w, hiA, loA, ec.floor_p, ec.is_val, 0u, dout);
                        epi_phase<0, ph>(P.t[tile][2], P.t[tile][3], w.b, hiB, loB, ec.floor_p, ec.is_val, 0u, dout);
                        if constexpr (ph == 2) {
                            xb[HU & 1].h[2 * pq] = hiA; xb[HU & 1].l[2 * pq] = loA;
                            xb[HU & 1].h[2 * pq + 1] = hiB; xb[HU & 1].l[2 * pq + 1] = loB;
                        }
                    }
                }
            } else
            if constexpr (HU >= 0 && T >= 4) {
                constexpr int pr = (T - 4) / 3, ph = (T - 4) % 3;
                constexpr int tile = (HU == 100 ? 0 : 2 * HU) + (pr >> 1), r0 = 2 * (pr & 1);
                unsigned hi = 0, lo = 0, dout = 0;
                const unsigned din = L::loads(HM) ? gc.dbuf[ks & 1][pr] : 0u;
                const unsigned din2 = (HM == 11) ? gc.abuf[ks & 1][pr] : 0u;
                if constexpr (HU == 100) {
                    epi_phase<HM, ph>(Q.t[tile][r0], Q.t[tile][r0 + 1], w, hi, lo, ec.floor_q, ec.is_val, din, dout, din2);
                    if constexpr (ph == 2) { x0n.h[pr] = hi; x0n.l[pr] = lo; }
                } else {
                    epi_phase<HM, ph>(P.t[tile][r0], P.t[tile][r0 + 1], w, hi, lo, ec.floor_p, ec.is_val, din, dout, din2);
                    if constexpr (ph == 2) { xb[HU & 1].h[pr] = hi; xb[HU & 1].l[pr] = lo; }
                }
                if constexpr (L::stores(HM)) {
                    if constexpr (HM == 5) { if constexpr (ph == 1) gc.dacc[pr] = dout; }
                    else { if constexpr (ph == 2) gc.dacc[pr] = hi; }
                    if constexpr (HM == 10) { if constexpr (ph == 1) gc.dacc2[pr] = dout; }
                    if constexpr (T == 15) {
                        gc.dpend = gc.dacc;
                        if constexpr (HM == 10) gc.dpend2 = gc.dacc2;
                        // forward sweeps (5, 8): slot = the layer that PRODUCED the activation; radiance backward (7, 9):
                        // rad_delta_slot(layer the delta is the cotangent of): the deltas of R0 .. R3 in the slots of THEIR
                        // layers' input activations (f, r0, r1, r2), then the geometry-feature cotangent
                        const int idx = (HM == 5 || HM == 8 || HM == 10) ? (HU == 100 ? gc.layer * 8 : (gc.layer - 1) * 8 + HU)
                                        : (HM == 11)                     ? (HU == 100 ? (gc.layer - 1) * 8 : gc.layer * 8 + HU)
                                                                         : (HU == 100 ? rad_delta_slot(gc.layer - 1) * 8
                                                                                      : rad_delta_slot(gc.layer) * 8 + HU);
                        if (gc.buf) gc.pend_soff = (unsigned)uoff(gc, idx);
                        else gc.pend_ptr = gc.ws_out + uoff(gc, idx);
                    }
                }
            }
#ifndef NERFART_EXP_NOSCHED
            __builtin_amdgcn_sched_barrier(0);
#endif
            Items<L, C, NKC, IT + 1>::run(P, Q, xb, xs, x0n, w, r, addr, s, ec, gc);
        }
    }
};

template <class L, int C>
__device__ __forceinline__ void run_chunk(const Acc& P, Acc& Q, Unit (&xb)[2], const Unit (&xs)[L::NXA], Unit& x0n,
                                          Work2& w, Stream& s, const EpiCtx& ec, GradCtx& gc) {
    constexpr int NKC = (L::NKS - CHUNK_KS * C) >= CHUNK_KS ? CHUNK_KS : (L::NKS - CHUNK_KS * C);
    if constexpr (NKC > 0) {
        const float* wp = stream_acquire(s) + lane_id() * 4;
        const unsigned addr = (unsigned)(size_t)wp;        // LDS byte address of this lane's 16 bytes of item 0
        // everything the compiler itself has in flight on the LDS queue must be drained first: the counted
        // waits below assume only the ring's reads are outstanding
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        RingT<L::AHEAD + 1 + RING_EXTRA> r;
#ifdef NERFART_F16X1
        static_assert(L::AHEAD == 2, "the 1-MFMA form is written for a two-item read-ahead");
        if constexpr (CHUNK_KS * C >= L::NH) {
            lds_read_pair<0>(r.h[0], r.l[0], addr);
            lds_read_pair<2048>(r.h[1], r.l[1], addr);
        } else {
            lds_read_hi<0>(r.h[0], addr);
            lds_read_hi<2048>(r.h[1], addr);
        }
#else
        lds_read_pair<0>(r.h[0], r.l[0], addr);
        lds_read_pair<2048>(r.h[1], r.l[1], addr);
#endif
#ifdef NERFART_EXP_SEG4
        lds_read_pair<4096>(r.h[2], r.l[2], addr);
        lds_read_pair<6144>(r.h[3], r.l[3], addr);
#else
        if constexpr (L::AHEAD >= 3) lds_read_pair<4096>(r.h[2], r.l[2], addr);
#endif
        Items<L, C, NKC, 0>::run(P, Q, xb, xs, x0n, w, r, addr, s, ec, gc);
        run_chunk<L, C + 1>(P, Q, xb, xs, x0n, w, s, ec, gc);
    }
}

// One dense layer: Q = bias + W . [act(P) | xs].  x0 = unit 0 of act(P) (built by the previous layer).
template <class L>
__device__ __forceinline__ void layer(const Acc& P, Acc& Q, const Unit& x0, const Unit (&xs)[L::NXA], Unit& x0n,
                                      Stream& s, const float* bias, const EpiCtx& ec, GradCtx& gc) {
    const int g = lane_id() >> 4;
    Unit xb[2];
    xb[0] = x0;
    xb[1] = x0;
    // start at the bias (value columns; tangent columns start at 0): reg r of tile T is feature 16T + 4g + r
#pragma unroll
    for (int T = 0; T < 16; ++T) {
        if constexpr (L::ZINIT) {
            Q.t[T] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            const f32x4 b = *reinterpret_cast<const f32x4*>(bias + 16 * T + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) Q.t[T][r] = ec.is_val ? b[r] : 0.f;
        }
    }
    Work2 w = {};
    run_chunk<L, 0>(P, Q, xb, xs, x0n, w, s, ec, gc);
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");     // last MFMA result -> first VALU reader (see mfma3)
}

// Epilogue of the last hidden layer: dot products of act(Q) with NROWS rows (+ the fp32 activations h7).
template <int MODE, int NROWS>
__device__ __forceinline__ void last_epilogue(const Acc& Q, const float* rows, float (&dot)[NROWS], float* h7, const EpiCtx& ec,
                                              char* dump = nullptr, unsigned dump_unit_stride = 0) {     // dump: this lane's unit 0
    const int g = lane_id() >> 4;
    u32x4 du;
#pragma unroll
    for (int T = 0; T < 16; ++T) {
        f32x4 y;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = Q.t[T][r];
            if constexpr (MODE == 2) {
                y[r] = fmaxf(a, ec.floor_q);
            } else if constexpr (MODE == 1) {
                float v, d;
                softplus100_vd(a, v, d);
                d = quad_bcast0(d);
                y[r] = ec.is_val ? v : d * a;
            } else {
#ifdef NERFART_F16X1
                y[r] = fmaxf(a, 0.f) + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(a)));     // a' = c softplus(z), the sdf row carries 1 / c
#else
                y[r] = softplus100(a);
#endif
            }
        }
#pragma unroll
        for (int n = 0; n < NROWS; ++n) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(rows + n * 256 + 16 * T + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) dot[n] = fmaf(y[r], w[r], dot[n]);
        }
        if constexpr (MODE != 2) {
#ifdef NERFART_EXP_H7_NT
            if (h7 != nullptr && ec.is_val) __builtin_nontemporal_store(y, reinterpret_cast<f32x4*>(h7 + 16 * T + 4 * g));
#else
            if (h7 != nullptr && ec.is_val) *reinterpret_cast<f32x4*>(h7 + 16 * T + 4 * g) = y;
#endif
        } else {
            if (dump != nullptr) {          // unit T/2 = (tile T regs 0..3 | tile T+1 regs 0..3), bf16 hi parts
                du[2 * (T & 1)] = pack_bf16(y[0], y[1]);
                du[2 * (T & 1) + 1] = pack_bf16(y[2], y[3]);
                if (T & 1) *reinterpret_cast<u32x4*>(dump + (size_t)(T >> 1) * dump_unit_stride) = du;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// Positional encoding of the SDF net in unit order: 2 units x (4 groups x 8 slots) = 64 slots.  Lane group
// g < 3 owns coordinate g: local index m = 8q + e: m = 0 raw, m = 1 + 2k sin(2^k x_g), m = 2 + 2k
// cos(2^k x_g) for k < 6, m = 13..15 pad; lane group 3 is padding (reference Embedder, models/base.py:38-64;
// slot map packing.unit_feature_enc).  dq < 0: values; dq = 0..2: derivative w.r.t. coordinate dq.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void encode_units(float x, float y, float z, int g, int dq, Unit (&X)[2]) {
    const float cg = (g == 0) ? x : ((g == 1) ? y : z);
    const bool live = g < 3;
    const bool own = (dq == g);
    float m[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) m[k] = 0.f;
    m[0] = (dq < 0) ? cg : (own ? 1.f : 0.f);
    // sin / cos of 2^k x, k = 0..5: two libm-accurate anchors (k = 0 and k = 3; 2^k x is exact in fp32) and two angle doublings
    // from each (sin 2a = 2 s c, cos 2a = (c - s)(c + s)): at most 4x the anchors' rounding error (~2.5e-7) instead of six
    // sincosf calls - the encodings were 15 % of the SDF kernels' time ("skeleton").
    float sk[6], ck[6];
    sincosf(cg, &sk[0], &ck[0]);
    sincosf(cg * 8.0f, &sk[3], &ck[3]);
#pragma unroll
    for (int a = 0; a < 6; a += 3)
#pragma unroll
        for (int k = a + 1; k < a + 3; ++k) {
            sk[k] = 2.0f * sk[k - 1] * ck[k - 1];
            ck[k] = (ck[k - 1] - sk[k - 1]) * (ck[k - 1] + sk[k - 1]);
        }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const float f = (float)(1 << k);
        const float s = sk[k], c = ck[k];
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
        X[q].h = hi;
        X[q].l = lo;
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

// The 8 hidden layers of the SDF net; returns row0 . h7 summed over the wave's lane groups (and stores h7 if
// asked).  Layer bodies: layer 0 (2 encoding units), one body for layers 1..3, 5, 6, the skip layer 4 (7 hidden
// units + 2 encoding units; the 1/sqrt(2) of cat[h, enc]/sqrt(2) is folded into its weights), layer 7.
template <int MODE>
__device__ __forceinline__ float surface_chain(float px, float py, float pz, int g, int dq, Stream& s, const float* aux,
                                               float* h7_lane, bool is_val) {
    const EpiCtx ec{0.f, 0.f, is_val};
    Acc A, B;
    Unit x0, x0n, enc[2], none[1];
    encode_units(px, py, pz, g, dq, enc);
    none[0] = enc[0];
    x0 = enc[0];
    GradCtx gc;
    // AHEAD = 3 (a deeper fragment ring) was measured 3 % SLOWER on K2: the LDS is throughput-, not latency-limited
    constexpr int AH = 2;
    layer<Cfg<MODE, MODE, 0, 2, true, false, false, false, AH>>(B, A, x0, enc, x0n, s, aux, ec, gc);
    x0 = x0n;
#pragma nounroll
    for (int L = 1; L < 7; ++L) {
        if (L == 4) {
            // K2 (MODE 0) has the registers to keep layer 0's encoding units alive until the skip layer (16 VGPRs, round 6: 2 sincosf + 4 angle
            // doublings + 16 splits per lane per tile less, ~6 % of K2's VALU instructions, same bits); the tangent kernels recompute them
            if constexpr (MODE != 0) encode_units(px, py, pz, g, dq, enc);
            layer<Cfg<MODE, MODE, 7, 2, true, false, false, false, AH>>(A, B, x0, enc, x0n, s, aux + L * 256, ec, gc);
        } else {
            layer<Cfg<MODE, MODE, 8, 0, true, false, false, false, AH>>(A, B, x0, none, x0n, s, aux + L * 256, ec, gc);
        }
        A = B;
        x0 = x0n;
    }
    layer<Cfg<MODE, MODE, 8, 0, false, false, false, false, AH>>(A, B, x0, none, x0n, s, aux + 7 * 256, ec, gc);
    float dot[1] = {0.f};
    last_epilogue<MODE, 1>(B, aux + SURF_AUX_ROW, dot, h7_lane, ec);
    return sum_over_groups(dot[0]);        // the 4 lane groups of a column hold complementary feature sets
}

__device__ __forceinline__ Stream make_stream(const float* blob, const float* aux, float* smem, int nc) {
    Stream s;
    s.blob = blob; s.tab = reinterpret_cast<const int*>(aux + AUX_FLOATS_MAX); s.lds = smem; s.nc = nc;
    s.iss_src = blob; s.iss_dst = 0; s.voff_a = 0; s.voff_b = 0; s.nxt = -1; s.nxt_o0 = 0; s.pb = 0; s.wrap = false; s.late_st = 0; s.lo_exec = ~0ull;
    return s;
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

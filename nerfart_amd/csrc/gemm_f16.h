// gemm_f16.h - the fp16 MFMA GEMM shared by the CLIP image encoder (clip_vit.hip) and the VGG16 perceptual net (vgg_conv.hip):
//
//   C[M, N] = A[M, K] . W[N, K]^T      64 x 64 x 64 tiles, 256 threads = 4 waves x (32 x 32) on v_mfma_f32_32x32x16_f16, fp32
//   accumulate, LDS double buffer (row stride 72 halfs: the 16-byte fragment reads of 16 consecutive rows hit 16 disjoint bank
//   quads), global -> register -> LDS staging of tile k + 1 under the MFMAs of tile k, one barrier per k tile.
//   BT = true:  C[M, N] = A[M, K] . Wt[K, N] with Wt row-major (the reduction index is Wt's ROW): the backward GEMMs read the
//   forward weight matrices in place - the tile is staged as it lies in memory ([k][n]) and the B fragments come out of LDS through
//   ds_read_b64_tr_b16, the 4 x 4 transposing read (16 lanes fetch 4 rows x 16 columns, lane c receives the 4 rows of column c;
//   two reads = the 8 k-values of a 32x32x16 operand).  Round 2 kept a transposed copy of every matrix in the blob for this.
//
// A sources (template ASRC): fp16 matrix; fp32 matrix converted while staging; IMPLICIT 3 x 3 convolution: row m is pixel
// (b, y, x) of an NHWC fp16 image, column k = (ky, kx, c) is channel c of the neighbour (y + ky - 1, x + kx - 1) (zero outside):
// im2col never exists in memory, each 16-half chunk of a row is one contiguous 32-byte load (C is a multiple of 16).
// Epilogues (template EPI): bias, residual add, QuickGELU (+ kept pre-activation), QuickGELU' multiply, bias + ReLU, ReLU mask.
#pragma once
#include "nerfart_common.h"

namespace nerfart {
namespace gemm16 {

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LS = 72;                  // LDS row stride of a 64-column fp16 tile, in halfs

enum { EPI_F32 = 0, EPI_F16 = 1, EPI_BIAS_F16 = 2, EPI_BIAS_RESID_F32 = 3, EPI_BIAS_GELU_F16 = 4, EPI_GELUBWD_F16 = 5,
       EPI_BIAS_RELU_F16 = 6, EPI_RELUMASK_F16 = 7 };
enum { A_F16 = 0, A_F32 = 1, A_CONV3 = 2 };

struct Epi {
    const float* bias;      // [N]
    const float* resid;     // [M, ldo] fp32 (EPI_BIAS_RESID_F32)
    float* out_f32;
    half_t* out_f16;
    half_t* out2_f16;       // EPI_BIAS_GELU_F16: the activation (out_f16 holds the pre-activation)
    const half_t* aux_f16;  // EPI_GELUBWD_F16: pre-activation;  EPI_RELUMASK_F16: the forward activation (mask = aux > 0)
    int ldo;                // row stride of every output / aux / resid matrix
    int m_valid;            // rows >= m_valid are computed (padding) but never stored
    int cH, cW, cC;         // A_CONV3: image height, width, channels (rows m = (b cH + y) cW + x)
};

__device__ __forceinline__ float quick_gelu(float x) { return x / (1.0f + __expf(-1.702f * x)); }
__device__ __forceinline__ float quick_gelu_grad(float x) {
    const float s = 1.0f / (1.0f + __expf(-1.702f * x));
    return s * (1.0f + 1.702f * x * (1.0f - s));
}

// 8 halfs of one column out of a row-major [k][n] LDS tile: rows k0 .. k0 + 7 of this lane's column (addr = the lane's 8-byte piece
// of the first 4 x 16 block: row (lane & 15) >> 2, columns 4 (lane & 3) ..; the second block lies 4 rows = ROWS4 bytes further)
template <int OFF, int ROWS4>
__device__ __forceinline__ half8 frag_tr(unsigned addr) {
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    u32x2_ a, b;
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b) : "v"(addr), "i"(OFF), "i"(OFF + ROWS4) : "memory");
    const u32x4_ r = {a[0], a[1], b[0], b[1]};
    return __builtin_bit_cast(half8, r);
}

template <int EPI, int ASRC, bool BT = false>
__global__ __launch_bounds__(256) void k_gemm(const void* __restrict__ Av, int lda, const half_t* __restrict__ W, int K, int ldw, Epi e) {
    __shared__ __attribute__((aligned(16))) half_t As[2][64][LS];
    __shared__ __attribute__((aligned(16))) half_t Bs[2][64][LS];
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int bm = blockIdx.y * 64, bn = blockIdx.x * 64;
    const int lr = tid >> 2, lc = (tid & 3) * 16;           // this thread stages 16 halfs of row lr at column lc of each tile
    int py = 0, px = 0;
    size_t pbase = 0;
    if constexpr (ASRC == A_CONV3) {
        const int m = bm + lr;
        px = m % e.cW;
        py = (m / e.cW) % e.cH;
        pbase = (size_t)(m - py * e.cW - px) * e.cC;         // start of image b
    }
    half8 ra0, ra1, rb0, rb1;
    auto fetch = [&](int k0) {
        if constexpr (ASRC == A_F32) {
            const float* p = reinterpret_cast<const float*>(Av) + (size_t)(bm + lr) * lda + k0 + lc;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
            const f32x4 v2 = *reinterpret_cast<const f32x4*>(p + 8), v3 = *reinterpret_cast<const f32x4*>(p + 12);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra0[i] = (half_t)v0[i]; ra0[4 + i] = (half_t)v1[i];
                ra1[i] = (half_t)v2[i]; ra1[4 + i] = (half_t)v3[i];
            }
        } else if constexpr (ASRC == A_CONV3) {
            const int k = k0 + lc, tap = k / e.cC, c0 = k - tap * e.cC;
            const int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
#pragma unroll
            for (int i = 0; i < 8; ++i) { ra0[i] = (half_t)0.f; ra1[i] = (half_t)0.f; }
            if (yy >= 0 && yy < e.cH && xx >= 0 && xx < e.cW) {
                const half_t* p = reinterpret_cast<const half_t*>(Av) + pbase + ((size_t)yy * e.cW + xx) * e.cC + c0;
                ra0 = *reinterpret_cast<const half8*>(p);
                ra1 = *reinterpret_cast<const half8*>(p + 8);
            }
        } else {
            const half_t* p = reinterpret_cast<const half_t*>(Av) + (size_t)(bm + lr) * lda + k0 + lc;
            ra0 = *reinterpret_cast<const half8*>(p);
            ra1 = *reinterpret_cast<const half8*>(p + 8);
        }
        // W[N, K]: row bn + lr, columns k0 + lc ..;  Wt[K, N] (BT): row k0 + lr, columns bn + lc ..  (ldw = the row stride)
        const half_t* q = BT ? W + (size_t)(k0 + lr) * ldw + bn + lc : W + (size_t)(bn + lr) * ldw + k0 + lc;
        rb0 = *reinterpret_cast<const half8*>(q);
        rb1 = *reinterpret_cast<const half8*>(q + 8);
    };
    auto stage = [&](int buf) {
        *reinterpret_cast<half8*>(&As[buf][lr][lc]) = ra0;
        *reinterpret_cast<half8*>(&As[buf][lr][lc + 8]) = ra1;
        *reinterpret_cast<half8*>(&Bs[buf][lr][lc]) = rb0;
        *reinterpret_cast<half8*>(&Bs[buf][lr][lc + 8]) = rb1;
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int wm = (w >> 1) * 32, wn = (w & 1) * 32, r = l & 31, h8 = (l >> 5) * 8;
    const int nk = K / 64;
    // BT: this lane's piece of the first 4 x 16 block of k-step 0 (bytes from the start of a Bs buffer)
    const unsigned bt_lane = (unsigned)(((h8 + ((l & 15) >> 2)) * LS + wn + 16 * ((l >> 4) & 1) + 4 * (l & 3)) * 2);
    const unsigned bs0 = (unsigned)(size_t)&Bs[0][0][0];
    fetch(0);
    stage(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) fetch((kt + 1) * 64);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const half8 a = *reinterpret_cast<const half8*>(&As[buf][wm + r][16 * s + h8]);
            half8 b;
            if constexpr (BT) {
                const unsigned ba = bs0 + buf * (64 * LS * 2) + bt_lane;
                b = s == 0 ? frag_tr<0, 4 * LS * 2>(ba) : s == 1 ? frag_tr<16 * LS * 2, 4 * LS * 2>(ba)
                  : s == 2 ? frag_tr<32 * LS * 2, 4 * LS * 2>(ba) : frag_tr<48 * LS * 2, 4 * LS * 2>(ba);
            } else {
                b = *reinterpret_cast<const half8*>(&Bs[buf][wn + r][16 * s + h8]);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        }
        if (kt + 1 < nk) stage(buf ^ 1);
        __syncthreads();
    }
    // C layout: lane (col = l & 31, half = l >> 5), reg i -> row (i & 3) + 8 (i >> 2) + 4 half
    const int col = bn + wn + r;
    float bias = 0.f;
    if constexpr (EPI == EPI_BIAS_F16 || EPI == EPI_BIAS_RESID_F32 || EPI == EPI_BIAS_GELU_F16 || EPI == EPI_BIAS_RELU_F16) bias = e.bias[col];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = bm + wm + (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
        if (row >= e.m_valid) continue;
        const size_t o = (size_t)row * e.ldo + col;
        const float v = acc[i] + bias;
        if constexpr (EPI == EPI_F32) e.out_f32[o] = v;
        else if constexpr (EPI == EPI_F16 || EPI == EPI_BIAS_F16) e.out_f16[o] = (half_t)v;
        else if constexpr (EPI == EPI_BIAS_RESID_F32) e.out_f32[o] = e.resid[o] + v;
        else if constexpr (EPI == EPI_BIAS_GELU_F16) { e.out_f16[o] = (half_t)v; e.out2_f16[o] = (half_t)quick_gelu(v); }
        else if constexpr (EPI == EPI_GELUBWD_F16) e.out_f16[o] = (half_t)(v * quick_gelu_grad((float)e.aux_f16[o]));
        else if constexpr (EPI == EPI_BIAS_RELU_F16) e.out_f16[o] = (half_t)fmaxf(v, 0.f);
        else e.out_f16[o] = ((float)e.aux_f16[o] > 0.f) ? (half_t)v : (half_t)0.f;
    }
}

// A_F16 / A_F32: A is [Mp, lda];  A_CONV3: A is the NHWC image, lda unused, K = 9 C.  Mp, N, K multiples of 64.
// BT: W is Wt[K, N] row-major (a forward weight matrix [out = K, in = N] read in place by its backward GEMM).
template <int EPI, int ASRC, bool BT = false>
static int gemm(hipStream_t st, const void* A, int lda, const half_t* W, int Mp, int N, int K, const Epi& e) {
    hipLaunchKernelGGL((k_gemm<EPI, ASRC, BT>), dim3(N / 64, Mp / 64), dim3(256), 0, st, A, lda, W, K, BT ? N : K, e);
    return check_hip(hipGetLastError(), "k_gemm launch");
}

}  // namespace gemm16
}  // namespace nerfart

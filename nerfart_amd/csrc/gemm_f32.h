// gemm_f32.h - fp32 MFMA GEMM of the VGG16 perceptual net (vgg_conv.hip).  The reference runs torchvision's VGG in fp32
// (criteria/perp_loss.py) and the term's pixel gradient is piecewise constant in ~24 M ReLU / max-pool / sign decisions, so the
// operands stay fp32 (v_mfma_f32_32x32x2_f32: exact fp32 products, fp32 accumulation) - fp16 operands flip ~1e-3 of the
// decisions and move the gradient by 6e-2 (measured, round 2).
//
//   C[M, N] = A[M, K] . W[N, K]^T     64 x 64 x 32 tiles, 256 threads = 4 waves x (32 x 32), LDS double buffer (row stride 36
//   floats), global -> register -> LDS staging of tile k + 1 under the MFMAs of tile k, one barrier per k tile.
//   A sources: fp32 matrix, or IMPLICIT 3 x 3 convolution over an NHWC fp32 image (row m = pixel (b, y, x), column k =
//   (ky, kx, c) = channel c of neighbour (y + ky - 1, x + kx - 1), zero outside; C a multiple of 8).
//   Epilogues: plain, bias + ReLU, ReLU mask (aux > 0).
#pragma once
#include "nerfart_common.h"

namespace nerfart {
namespace gemm32 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LS = 36;                  // LDS row stride of a 32-column fp32 tile, in floats
enum { EPI_PLAIN = 0, EPI_BIAS_RELU = 1, EPI_RELUMASK = 2, EPI_BIAS = 3 };      // EPI_BIAS: + bias, no activation (geo_feature.hip)
enum { A_MAT = 0, A_CONV3 = 1 };

struct Epi {
    const float* bias;      // [N]
    float* out;
    const float* aux;       // EPI_RELUMASK: the forward activation of the layer below (mask = aux > 0)
    int ldo;                // row stride of out / aux
    int m_valid;            // rows >= m_valid are computed (padding) but never stored
    int cH, cW, cC;         // A_CONV3: image height, width, channels (rows m = (b cH + y) cW + x)
};

template <int EPI, int ASRC>
__global__ __launch_bounds__(256) void k_gemm(const float* __restrict__ A, int lda, const float* __restrict__ W, int K, Epi e) {
    __shared__ __attribute__((aligned(16))) float As[2][64][LS];
    __shared__ __attribute__((aligned(16))) float Bs[2][64][LS];
    const int tid = threadIdx.x, l = tid & 63, w = tid >> 6;
    const int bm = blockIdx.y * 64, bn = blockIdx.x * 64;
    const int lr = tid >> 2, lc = (tid & 3) * 8;            // this thread stages 8 floats of row lr at column lc of each tile
    int py = 0, px = 0;
    size_t pbase = 0;
    if constexpr (ASRC == A_CONV3) {
        const int m = bm + lr;
        px = m % e.cW;
        py = (m / e.cW) % e.cH;
        pbase = (size_t)(m - py * e.cW - px) * e.cC;         // start of image b
    }
    f32x4 ra0, ra1, rb0, rb1;
    auto fetch = [&](int k0) {
        if constexpr (ASRC == A_CONV3) {
            const int k = k0 + lc, tap = k / e.cC, c0 = k - tap * e.cC;
            const int yy = py + tap / 3 - 1, xx = px + tap % 3 - 1;
            ra0 = f32x4{0.f, 0.f, 0.f, 0.f};
            ra1 = ra0;
            if (yy >= 0 && yy < e.cH && xx >= 0 && xx < e.cW) {
                const float* p = A + pbase + ((size_t)yy * e.cW + xx) * e.cC + c0;
                ra0 = *reinterpret_cast<const f32x4*>(p);
                ra1 = *reinterpret_cast<const f32x4*>(p + 4);
            }
        } else {
            // rows past m_valid are padding (computed, never stored): re-read the last valid row instead of memory past a [m_valid, lda] matrix
            const int ar = bm + lr < e.m_valid ? bm + lr : e.m_valid - 1;
            const float* p = A + (size_t)ar * lda + k0 + lc;
            ra0 = *reinterpret_cast<const f32x4*>(p);
            ra1 = *reinterpret_cast<const f32x4*>(p + 4);
        }
        const float* q = W + (size_t)(bn + lr) * K + k0 + lc;
        rb0 = *reinterpret_cast<const f32x4*>(q);
        rb1 = *reinterpret_cast<const f32x4*>(q + 4);
    };
    auto stage = [&](int buf) {
        *reinterpret_cast<f32x4*>(&As[buf][lr][lc]) = ra0;
        *reinterpret_cast<f32x4*>(&As[buf][lr][lc + 4]) = ra1;
        *reinterpret_cast<f32x4*>(&Bs[buf][lr][lc]) = rb0;
        *reinterpret_cast<f32x4*>(&Bs[buf][lr][lc + 4]) = rb1;
    };
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    // 32x32x2: lane (row / col = l & 31, k = l >> 5).  Lane half h reads k = 8 s + 4 h .. + 3 of its row as one 16-byte LDS read
    // and feeds the four values to four MFMAs: A and B use the same k assignment, so every k meets its partner exactly once.
    const int wm = (w >> 1) * 32, wn = (w & 1) * 32, r = l & 31, h4 = (l >> 5) * 4;
    const int nk = K / 32;
    fetch(0);
    stage(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) fetch((kt + 1) * 32);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&As[buf][wm + r][8 * s + h4]);
            const f32x4 b = *reinterpret_cast<const f32x4*>(&Bs[buf][wn + r][8 * s + h4]);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q], b[q], acc, 0, 0, 0);
        }
        if (kt + 1 < nk) stage(buf ^ 1);
        __syncthreads();
    }
    // C layout: lane (col = l & 31, half = l >> 5), reg i -> row (i & 3) + 8 (i >> 2) + 4 half
    const int col = bn + wn + r;
    float bias = 0.f;
    if constexpr (EPI == EPI_BIAS_RELU || EPI == EPI_BIAS) bias = e.bias[col];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = bm + wm + (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
        if (row >= e.m_valid) continue;
        const size_t o = (size_t)row * e.ldo + col;
        const float v = acc[i] + bias;
        if constexpr (EPI == EPI_PLAIN || EPI == EPI_BIAS) e.out[o] = v;
        else if constexpr (EPI == EPI_BIAS_RELU) e.out[o] = fmaxf(v, 0.f);
        else e.out[o] = (e.aux[o] > 0.f) ? v : 0.f;
    }
}

// A_MAT: A is [Mp, lda];  A_CONV3: A is the NHWC image, lda unused, K = 9 C.  Mp, N multiples of 64, K of 32.
template <int EPI, int ASRC>
static int gemm(hipStream_t st, const float* A, int lda, const float* W, int Mp, int N, int K, const Epi& e) {
    hipLaunchKernelGGL((k_gemm<EPI, ASRC>), dim3(N / 64, Mp / 64), dim3(256), 0, st, A, lda, W, K, e);
    return check_hip(hipGetLastError(), "k_gemm32 launch");
}

}  // namespace gemm32
}  // namespace nerfart

// clip_vit.hip - CLIP ViT-B/32 image encoder, forward and backward (d features / d pixels), hand-written for gfx950.
//
// Rows a23 / B4 of SURVEY.md section 8: what `model.encode_image(img)` of `clip.load("ViT-B/32", device="cuda")` computes
// at the reference's call sites criteria/clip_loss.py:204-216, contrastive_loss.py:110-114, patchnce_loss.py:124-128, and
// what autograd does to it when the style loss is back-propagated to the rendered image (volsdf.py:912-915).  The CLIP
// weights are frozen there, so the backward pass is the input-gradient chain only (no weight gradients).
//
//   image [B,3,224,224] -> 49 patches x 3072 -> conv1 as a GEMM -> [cls; patches] + pos -> ln_pre
//     -> 12 x { x += out_proj(attn(in_proj(ln_1 x)));  x += c_proj(quick_gelu(c_fc(ln_2 x))) } -> ln_post(x[:,0]) @ proj
//
// Numerics: fp16 weights and fp16 GEMM operands with fp32 accumulation on v_mfma_f32_32x32x16_f16, LayerNorm / softmax /
// residual stream in fp32 (clip.load's CUDA model keeps the residual stream in fp16; this one is strictly more accurate).
// Backward GEMM operands are fp16 too: the incoming cotangent is multiplied by a power of two chosen on the device from
// its max-abs (so nothing under- or overflows in fp16; the chain is linear) and the pixel gradient is divided by it.
//
// Kernels
//   k_gemm<EPI, A_F32>   C[M,N] = A[M,K] . W[N,K]^T, 64x64x64 tiles, 4 waves x (32x32) on mfma 32x32x16 f16, LDS double
//                        buffer (row stride 72 halfs: the 16-byte fragment reads of 16 consecutive rows hit 16 disjoint
//                        bank quads), fused epilogues: bias, residual add, QuickGELU (+ pre-activation kept for backward),
//                        QuickGELU' multiply; backward uses the same kernel on the SAME matrices (BT: transposing LDS reads).
//   k_attn_fwd / k_attn_bwd   one workgroup per (image, head): Q, K, V (50 x 64, zero padded to 64) and their transposes in
//                        LDS, S = Q K^T, softmax rows in fp32, O = P V;  backward recomputes P, then dP = dO V^T,
//                        dS = P o (dP - rowsum(P o dP)) / 8, dQ = dS K, dK = dS^T Q, dV = P^T dO - five 64^3 products on MFMA.
//   k_ln_fwd / k_ln_bwd  one wave per token row (768 = 64 lanes x 3 float4), statistics in fp32; backward recomputes them.
//   k_patchify / k_unpatchify, k_embed_lnpre / k_lnpre_bwd, k_lnpost / k_lnpost_bwd, k_gscale: the ends of the chain.
#include "gemm_f16.h"
#include <stdio.h>
#include <cmath>

namespace nerfart {
namespace clip {

using namespace nerfart::gemm16;

constexpr int D = 768, L = 50, NH = 12, HD = 64, DM = 3072, NL = 12, DOUT = 512, NP = 49, PK = 3072, IMG = 224, PS = 32;
constexpr float LN_EPS = 1e-5f;

// ---------------------------------------------------------------------------------------------------------------
// Weight blob: sections in a fixed order, each 256-byte aligned.  fp16 matrices are row-major [rows, cols], stored ONCE: a
// backward GEMM reads its forward matrix in place (gemm_f16.h, BT: transposing LDS reads).  The odd sections below held the
// transposed copies in round 2 (352 MB blob); they are empty now (176 MB) and keep their numbers.
// nerfart_clip_vitb32_blob_layout() returns the offsets.
//   0 conv [768,3072]   1 (empty)
//   2 + 8 l + j, l < 12: j = 0 in_proj W [2304,768], 2 out_proj W [768,768], 4 c_fc W [3072,768], 6 c_proj W [768,3072];
//                        j = 1, 3, 5, 7 (empty)
//   98 (empty)  99 proj [768,512]
//   fp32: 100 class_embedding [768]  101 positional_embedding [50,768]  102 ln_pre.weight  103 ln_pre.bias
//   104 + 8 l + j: j = 0 ln_1.weight, 1 ln_1.bias, 2 in_proj_bias [2304], 3 out_proj.bias, 4 ln_2.weight, 5 ln_2.bias,
//                  6 c_fc.bias [3072], 7 c_proj.bias
//   200 ln_post.weight  201 ln_post.bias
// ---------------------------------------------------------------------------------------------------------------
constexpr int N_SECTIONS = 202;
static long long section_bytes(int i) {
    if (i == 0) return 2LL * D * PK;
    if (i == 1 || i == 98) return 0;
    if (i >= 2 && i < 98) {
        const int j = (i - 2) & 7;
        if (j & 1) return 0;
        return 2LL * D * (j < 2 ? 3 * D : (j < 4 ? D : DM));
    }
    if (i == 99) return 2LL * D * DOUT;
    if (i == 100) return 4LL * D;
    if (i == 101) return 4LL * L * D;
    if (i == 102 || i == 103 || i == 200 || i == 201) return 4LL * D;
    const int j = (i - 104) & 7;
    return 4LL * (j == 2 ? 3 * D : (j == 6 ? DM : D));
}
static long long blob_layout(long long* offs) {
    long long o = 0;
    for (int i = 0; i < N_SECTIONS; ++i) {
        if (offs) offs[i] = o;
        o += (section_bytes(i) + 255) / 256 * 256;
    }
    if (offs) offs[N_SECTIONS] = o;
    return o;
}
struct Blob {
    const char* base;
    long long off[N_SECTIONS + 1];
    const half_t* h(int i) const { return reinterpret_cast<const half_t*>(base + off[i]); }
    const float* f(int i) const { return reinterpret_cast<const float*>(base + off[i]); }
};

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row; lane holds float4 at columns 4 lane + 256 j, j < 3.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
struct Row { f32x4 v[3]; };
__device__ __forceinline__ Row row_load(const float* p, int lane) {
    Row r;
#pragma unroll
    for (int j = 0; j < 3; ++j) r.v[j] = *reinterpret_cast<const f32x4*>(p + 4 * lane + 256 * j);
    return r;
}
__device__ __forceinline__ void row_stats(const Row& x, float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j) s += x.v[j][0] + x.v[j][1] + x.v[j][2] + x.v[j][3];
    mean = wave_sum(s) * (1.0f / D);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float d = x.v[j][i] - mean; q += d * d; }
    rstd = rsqrtf(wave_sum(q) * (1.0f / D) + LN_EPS);
}
// y = (x - mean) rstd g + b
__device__ __forceinline__ Row row_ln(const Row& x, const float* g, const float* b, int lane) {
    float mean, rstd;
    row_stats(x, mean, rstd);
    const Row gg = row_load(g, lane), bb = row_load(b, lane);
    Row y;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) y.v[j][i] = (x.v[j][i] - mean) * rstd * gg.v[j][i] + bb.v[j][i];
    return y;
}
// dx = rstd (gy - mean(gy) - xhat mean(gy xhat)),  gy = dy g
__device__ __forceinline__ Row row_ln_bwd(const Row& dy, const Row& x, const float* g, int lane) {
    float mean, rstd;
    row_stats(x, mean, rstd);
    const Row gg = row_load(g, lane);
    Row gy, xh;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            gy.v[j][i] = dy.v[j][i] * gg.v[j][i];
            xh.v[j][i] = (x.v[j][i] - mean) * rstd;
            s1 += gy.v[j][i];
            s2 += gy.v[j][i] * xh.v[j][i];
        }
    const float m1 = wave_sum(s1) * (1.0f / D), m2 = wave_sum(s2) * (1.0f / D);
    Row dx;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) dx.v[j][i] = rstd * (gy.v[j][i] - m1 - xh.v[j][i] * m2);
    return dx;
}
__device__ __forceinline__ void row_store_f16(half_t* p, const Row& y, int lane) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        half4 o = {(half_t)y.v[j][0], (half_t)y.v[j][1], (half_t)y.v[j][2], (half_t)y.v[j][3]};
        *reinterpret_cast<half4*>(p + 4 * lane + 256 * j) = o;
    }
}
__device__ __forceinline__ void row_store_f32(float* p, const Row& y, int lane, bool accumulate) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        f32x4* q = reinterpret_cast<f32x4*>(p + 4 * lane + 256 * j);
        *q = accumulate ? (*q + y.v[j]) : y.v[j];
    }
}

// rows of x [M, 768] fp32 -> y fp16
__global__ __launch_bounds__(256) void k_ln_fwd(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ b,
                                               half_t* __restrict__ y, int M) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    row_store_f16(y + (size_t)row * D, row_ln(row_load(x + (size_t)row * D, lane), g, b, lane), lane);
}
// dx[row] += LN'(dy[row]; x[row])
__global__ __launch_bounds__(256) void k_ln_bwd(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ g,
                                               float* __restrict__ dx, int M) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const Row d = row_ln_bwd(row_load(dy + (size_t)row * D, lane), row_load(x + (size_t)row * D, lane), g, lane);
    row_store_f32(dx + (size_t)row * D, d, lane, true);
}

// ---------------------------------------------------------------------------------------------------------------
// Ends of the chain
// ---------------------------------------------------------------------------------------------------------------
// img [B,3,224,224] fp32 -> patches fp16 [B 49, 3072], column = c 1024 + kh 32 + kw (conv1.weight.reshape(768, -1) order)
__global__ __launch_bounds__(256) void k_patchify(const float* __restrict__ img, half_t* __restrict__ patches, int B) {
    const int m = blockIdx.x;                       // b * 49 + p
    const int b = m / NP, p = m % NP, py = p / 7, px = p % 7;
    for (int k = threadIdx.x * 4; k < PK; k += 1024) {
        const int c = k >> 10, kh = (k >> 5) & 31, kw = k & 31;
        const f32x4 v = *reinterpret_cast<const f32x4*>(img + (((size_t)b * 3 + c) * IMG + py * PS + kh) * IMG + px * PS + kw);
        half4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        *reinterpret_cast<half4*>(patches + (size_t)m * PK + k) = o;
    }
}
// dpatch fp32 [B 49, 3072] (scaled by s) -> g_img [B,3,224,224] = dpatch / s
__global__ __launch_bounds__(256) void k_unpatchify(const float* __restrict__ dpatch, const float* __restrict__ scale, float* __restrict__ g_img, int B) {
    const int m = blockIdx.x;
    const int b = m / NP, p = m % NP, py = p / 7, px = p % 7;
    const float inv = scale[1];
    for (int k = threadIdx.x * 4; k < PK; k += 1024) {
        const int c = k >> 10, kh = (k >> 5) & 31, kw = k & 31;
        const f32x4 v = *reinterpret_cast<const f32x4*>(dpatch + (size_t)m * PK + k) * inv;
        *reinterpret_cast<f32x4*>(g_img + (((size_t)b * 3 + c) * IMG + py * PS + kh) * IMG + px * PS + kw) = v;
    }
}
// token rows: xemb = (t == 0 ? cls : pe[b 49 + t - 1]) + pos[t];  x = ln_pre(xemb)
__global__ __launch_bounds__(256) void k_embed_lnpre(const float* __restrict__ pe, const float* __restrict__ cls, const float* __restrict__ pos,
                                                    const float* __restrict__ g, const float* __restrict__ bb, float* __restrict__ xemb,
                                                    float* __restrict__ x, int M) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const int b = row / L, t = row % L;
    Row v = (t == 0) ? row_load(cls, lane) : row_load(pe + (size_t)(b * NP + t - 1) * D, lane);
    const Row pp = row_load(pos + (size_t)t * D, lane);
#pragma unroll
    for (int j = 0; j < 3; ++j) v.v[j] += pp.v[j];
    row_store_f32(xemb + (size_t)row * D, v, lane, false);
    row_store_f32(x + (size_t)row * D, row_ln(v, g, bb, lane), lane, false);
}
// dx (w.r.t. ln_pre's output) -> d patch-embedding rows, fp16 [B 49, 768]
__global__ __launch_bounds__(256) void k_lnpre_bwd(const float* __restrict__ dx, const float* __restrict__ xemb, const float* __restrict__ g,
                                                  half_t* __restrict__ dpe, int M) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const int b = row / L, t = row % L;
    if (t == 0) return;                              // the class token has no pixels behind it
    const Row d = row_ln_bwd(row_load(dx + (size_t)row * D, lane), row_load(xemb + (size_t)row * D, lane), g, lane);
    row_store_f16(dpe + (size_t)(b * NP + t - 1) * D, d, lane);
}
// x0[b] = ln_post(x[b * 50]) fp16
__global__ __launch_bounds__(256) void k_lnpost(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ bb,
                                               half_t* __restrict__ x0, int B) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    row_store_f16(x0 + (size_t)b * D, row_ln(row_load(x + (size_t)b * L * D, lane), g, bb, lane), lane);
}
// dx[b * 50] = LN'(t0[b]; x[b * 50])   (dx was zeroed: only the class rows receive a cotangent here)
__global__ __launch_bounds__(256) void k_lnpost_bwd(const float* __restrict__ t0, const float* __restrict__ x, const float* __restrict__ g,
                                                   float* __restrict__ dx, int B) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const Row d = row_ln_bwd(row_load(t0 + (size_t)b * D, lane), row_load(x + (size_t)b * L * D, lane), g, lane);
    row_store_f32(dx + (size_t)b * L * D, d, lane, false);
}
// scale[0] = s = 2^floor(log2(16 / max|g|)) (1 if g == 0), scale[1] = 1 / s;  g16 = fp16(g s), [B, 512] (one workgroup)
__global__ __launch_bounds__(1024) void k_gscale(const float* __restrict__ g, int n, float* __restrict__ scale, half_t* __restrict__ g16) {
    __shared__ float red[16];
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf(g[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = red[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
    const float s = (m > 0.f && isfinite(m)) ? exp2f(floorf(log2f(16.0f / m))) : 1.0f;
    if (threadIdx.x == 0) { scale[0] = s; scale[1] = 1.0f / s; }
    for (int i = threadIdx.x; i < n; i += 1024) g16[i] = (half_t)(g[i] * s);
}

// ---------------------------------------------------------------------------------------------------------------
// Attention, one workgroup (4 waves) per (image, head).  All operands are 64 x 64 fp16 tiles in LDS (rows >= 50 zero);
// every product has the form C[m][n] = sum_k A[m][k] B[n][k]; wave w owns the 32 x 32 output tile (w >> 1, w & 1).
// ---------------------------------------------------------------------------------------------------------------
typedef half_t Tile[64][LS];
typedef float TileF[64][65];

__device__ __forceinline__ f32x16 mm_tile(const Tile& A, const Tile& B, int w, int l) {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int m0 = (w >> 1) * 32 + (l & 31), n0 = (w & 1) * 32 + (l & 31), h8 = (l >> 5) * 8;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const half8 a = *reinterpret_cast<const half8*>(&A[m0][16 * s + h8]);
        const half8 b = *reinterpret_cast<const half8*>(&B[n0][16 * s + h8]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    return acc;
}
// rows t < 50 of the [*, ld] fp16 matrix at column c0 -> X (and X^T if XT != nullptr); rows >= 50 zero.  256 threads.
__device__ __forceinline__ void load_tile(const half_t* __restrict__ src, int ld, Tile& X, Tile* XT) {
    const int r = threadIdx.x >> 2, c = (threadIdx.x & 3) * 16;
    half8 v0, v1;
#pragma unroll
    for (int i = 0; i < 8; ++i) { v0[i] = (half_t)0.f; v1[i] = (half_t)0.f; }
    if (r < L) {
        v0 = *reinterpret_cast<const half8*>(src + (size_t)r * ld + c);
        v1 = *reinterpret_cast<const half8*>(src + (size_t)r * ld + c + 8);
    }
    *reinterpret_cast<half8*>(&X[r][c]) = v0;
    *reinterpret_cast<half8*>(&X[r][c + 8]) = v1;
    if (XT != nullptr) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { (*XT)[c + i][r] = v0[i]; (*XT)[c + 8 + i][r] = v1[i]; }
    }
}
__device__ __forceinline__ void store_tile_f32(TileF& S, const f32x16& acc, int w, int l) {
    const int col = (w & 1) * 32 + (l & 31);
#pragma unroll
    for (int i = 0; i < 16; ++i) S[(w >> 1) * 32 + (i & 3) + 8 * (i >> 2) + 4 * (l >> 5)][col] = acc[i];
}
// rows t < 50 of a C tile -> dst[t * ld + col] fp16
__device__ __forceinline__ void store_tile_out(half_t* __restrict__ dst, int ld, const f32x16& acc, int w, int l) {
    const int col = (w & 1) * 32 + (l & 31);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int row = (w >> 1) * 32 + (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
        if (row < L) dst[(size_t)row * ld + col] = (half_t)acc[i];
    }
}
// softmax of row r of S / 8 over the 50 keys, in place (fp32 probabilities; columns >= 50 zero)
__device__ __forceinline__ void softmax_row(TileF& S, int r) {
    float m = -INFINITY;
    for (int c = 0; c < L; ++c) m = fmaxf(m, S[r][c]);
    float sum = 0.f;
    for (int c = 0; c < L; ++c) { const float e = __expf((S[r][c] - m) * 0.125f); S[r][c] = e; sum += e; }
    const float inv = 1.0f / sum;
    for (int c = 0; c < L; ++c) S[r][c] *= inv;
    for (int c = L; c < 64; ++c) S[r][c] = 0.f;
}

__global__ __launch_bounds__(256) void k_attn_fwd(const half_t* __restrict__ qkv, half_t* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) Tile Q, K, VT, P, V;
    __shared__ TileF S;
    const int b = blockIdx.x / NH, h = blockIdx.x % NH, w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const half_t* base = qkv + (size_t)b * L * (3 * D) + h * HD;
    load_tile(base, 3 * D, Q, nullptr);
    load_tile(base + D, 3 * D, K, nullptr);
    load_tile(base + 2 * D, 3 * D, V, &VT);
    __syncthreads();
    store_tile_f32(S, mm_tile(Q, K, w, l), w, l);
    __syncthreads();
    if (threadIdx.x < 64) {
        const int r = threadIdx.x;
        if (r < L) softmax_row(S, r);
        for (int c = 0; c < 64; ++c) P[r][c] = (half_t)(r < L ? S[r][c] : 0.f);
    }
    __syncthreads();
    store_tile_out(out + (size_t)b * L * D + h * HD, D, mm_tile(P, VT, w, l), w, l);
}

__global__ __launch_bounds__(256) void k_attn_bwd(const half_t* __restrict__ qkv, const half_t* __restrict__ dout, half_t* __restrict__ dqkv) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Tile* T = reinterpret_cast<Tile*>(smem);
    Tile &Q = T[0], &K = T[1], &V = T[2], &dO = T[3], &QT = T[4], &KT = T[5], &dOT = T[6], &P = T[7], &PT = T[8], &dS = T[9], &dST = T[10];
    TileF& S = *reinterpret_cast<TileF*>(smem + 11 * sizeof(Tile));
    TileF& dP = *reinterpret_cast<TileF*>(smem + 11 * sizeof(Tile) + sizeof(TileF));
    const int b = blockIdx.x / NH, h = blockIdx.x % NH, w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const half_t* base = qkv + (size_t)b * L * (3 * D) + h * HD;
    load_tile(base, 3 * D, Q, &QT);
    load_tile(base + D, 3 * D, K, &KT);
    load_tile(base + 2 * D, 3 * D, V, nullptr);
    load_tile(dout + (size_t)b * L * D + h * HD, D, dO, &dOT);
    __syncthreads();
    store_tile_f32(S, mm_tile(Q, K, w, l), w, l);
    store_tile_f32(dP, mm_tile(dO, V, w, l), w, l);
    __syncthreads();
    if (threadIdx.x < 64) {
        const int r = threadIdx.x;
        if (r < L) {
            softmax_row(S, r);
            float rs = 0.f;
            for (int c = 0; c < L; ++c) rs += S[r][c] * dP[r][c];
            for (int c = 0; c < 64; ++c) {
                const float p = S[r][c];
                const half_t ph = (half_t)p, dh = (half_t)(c < L ? p * (dP[r][c] - rs) * 0.125f : 0.f);
                P[r][c] = ph; PT[c][r] = ph; dS[r][c] = dh; dST[c][r] = dh;
            }
        } else {
            for (int c = 0; c < 64; ++c) { P[r][c] = (half_t)0.f; PT[c][r] = (half_t)0.f; dS[r][c] = (half_t)0.f; dST[c][r] = (half_t)0.f; }
        }
    }
    __syncthreads();
    half_t* dbase = dqkv + (size_t)b * L * (3 * D) + h * HD;
    store_tile_out(dbase, 3 * D, mm_tile(dS, KT, w, l), w, l);              // dQ = dS K
    store_tile_out(dbase + D, 3 * D, mm_tile(dST, QT, w, l), w, l);         // dK = dS^T Q
    store_tile_out(dbase + 2 * D, 3 * D, mm_tile(PT, dOT, w, l), w, l);     // dV = P^T dO
}
constexpr size_t ATTN_BWD_LDS = 11 * sizeof(Tile) + 2 * sizeof(TileF);

// ---------------------------------------------------------------------------------------------------------------
// Workspace: [scale | scratch | saved].  keep = 0: one layer's worth of "saved" is reused by every layer (forward only).
// ---------------------------------------------------------------------------------------------------------------
static inline long long up(long long v, long long a) { return (v + a - 1) / a * a; }
struct Work {
    int B, M, Mp, Pp, Bp, keep;
    long long o_scale, o_patches, o_pe, o_xemb, o_xfin, o_y, o_att, o_act, o_t, o_dx, o_dqkv, o_x0, o_g16, o_t0, o_saved, layer_bytes, total;
    long long o_xin, o_xmid, o_qkv, o_pre;       // within one layer's saved block
};
static Work work_layout(int B, int keep) {
    Work w;
    w.B = B; w.M = B * L; w.Mp = (int)up(w.M, 64); w.Pp = (int)up((long long)B * NP, 64); w.Bp = (int)up(B, 64); w.keep = keep;
    long long o = 0;
    auto take = [&](long long bytes) { const long long at = o; o += up(bytes, 256); return at; };
    w.o_scale = take(256);
    w.o_patches = take(2LL * w.Pp * PK);          // fp16 patches; backward: fp32 d patches needs 4 B -> separate below
    w.o_pe = take(4LL * w.Pp * PK);               // fp32 [Pp, 768] forward patch embedding; backward: fp32 [Pp, 3072] d patches
    w.o_xemb = take(4LL * w.Mp * D);
    w.o_xfin = take(4LL * w.Mp * D);
    w.o_y = take(2LL * w.Mp * D);
    w.o_att = take(2LL * w.Mp * D);
    w.o_act = take(2LL * w.Mp * DM);
    w.o_t = take(4LL * w.Mp * D);
    w.o_dx = take(4LL * w.Mp * D);
    w.o_dqkv = take(2LL * w.Mp * 3 * D);
    w.o_x0 = take(2LL * w.Bp * D);
    w.o_g16 = take(2LL * w.Bp * DOUT);
    w.o_t0 = take(4LL * w.Bp * D);
    w.o_saved = o;
    long long lo = 0;
    auto ltake = [&](long long bytes) { const long long at = lo; lo += up(bytes, 256); return at; };
    w.o_xin = ltake(4LL * w.Mp * D);
    w.o_xmid = ltake(4LL * w.Mp * D);
    w.o_qkv = ltake(2LL * w.Mp * 3 * D);
    w.o_pre = ltake(2LL * w.Mp * DM);
    w.layer_bytes = lo;
    w.total = o + lo * (keep ? NL : 1) + (keep ? 0 : up(4LL * w.Mp * D, 256));   // !keep: a second x buffer to ping-pong
    return w;
}

static int check_args(const void* blob, const void* ws, int B, long long ws_bytes, int keep) {
    if (blob == nullptr || ws == nullptr) { set_last_error("clip_vitb32: null blob / workspace"); return 1; }
    if (B < 1 || B > 1024) { set_last_error("clip_vitb32: batch must be in 1..1024"); return 1; }
    if (ws_bytes < work_layout(B, keep).total) { set_last_error("clip_vitb32: workspace too small (nerfart_clip_vitb32_workspace_bytes)"); return 1; }
    return 0;
}
// a blob packed to another layout (the round-2 blob carried transposed copies: 352 MB) must be rejected, not read
static int check_blob_bytes(long long blob_bytes) {
    long long off[N_SECTIONS + 1];
    blob_layout(off);
    if (blob_bytes != off[N_SECTIONS]) { set_last_error("clip_vitb32: blob_bytes differs from nerfart_clip_vitb32_blob_layout(): blob packed to another layout / ABI version"); return 1; }
    return 0;
}

// ---- packer (ABI 3): `visual.*` state-dict tensors -> blob sections ---------------------------------------------------------------
// The tensors in the order nerfart_clip_vitb32_pack takes them (OpenAI CLIP parameter names below `visual.`): 5 stem tensors, 12 per
// residual block, 3 tail tensors = 152.  sec: the blob section; half: stored as fp16 (the GEMM operands) or fp32 (vectors).
struct PackEntry { const char* name; int sec; int half; long long n; };
static const char* BLOCK_NAMES[12] = {"ln_1.weight", "ln_1.bias", "attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight", "attn.out_proj.bias",
                                      "ln_2.weight", "ln_2.bias", "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias"};
constexpr int N_PACK_TENSORS = 5 + 12 * 12 + 3;
static PackEntry pack_entry(int i, char* name_buf, int name_len) {
    auto set = [&](const char* s) { if (name_buf) snprintf(name_buf, name_len, "%s", s); };
    if (i == 0) { set("conv1.weight"); return {nullptr, 0, 1, (long long)D * PK}; }
    if (i == 1) { set("class_embedding"); return {nullptr, 100, 0, D}; }
    if (i == 2) { set("positional_embedding"); return {nullptr, 101, 0, (long long)L * D}; }
    if (i == 3) { set("ln_pre.weight"); return {nullptr, 102, 0, D}; }
    if (i == 4) { set("ln_pre.bias"); return {nullptr, 103, 0, D}; }
    if (i < 5 + 144) {
        const int l = (i - 5) / 12, j = (i - 5) % 12;
        if (name_buf) snprintf(name_buf, name_len, "transformer.resblocks.%d.%s", l, BLOCK_NAMES[j]);
        const int s = 2 + 8 * l, f = 104 + 8 * l;
        switch (j) {
            case 0: return {nullptr, f + 0, 0, D};
            case 1: return {nullptr, f + 1, 0, D};
            case 2: return {nullptr, s + 0, 1, 3LL * D * D};
            case 3: return {nullptr, f + 2, 0, 3LL * D};
            case 4: return {nullptr, s + 2, 1, (long long)D * D};
            case 5: return {nullptr, f + 3, 0, D};
            case 6: return {nullptr, f + 4, 0, D};
            case 7: return {nullptr, f + 5, 0, D};
            case 8: return {nullptr, s + 4, 1, (long long)DM * D};
            case 9: return {nullptr, f + 6, 0, DM};
            case 10: return {nullptr, s + 6, 1, (long long)D * DM};
            default: return {nullptr, f + 7, 0, D};
        }
    }
    if (i == 149) { set("ln_post.weight"); return {nullptr, 200, 0, D}; }
    if (i == 150) { set("ln_post.bias"); return {nullptr, 201, 0, D}; }
    set("proj"); return {nullptr, 99, 1, (long long)D * DOUT};
}
__global__ void __launch_bounds__(256) k_pack_section(const float* __restrict__ src, void* __restrict__ dst, long long n, int to_half) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (to_half) reinterpret_cast<half_t*>(dst)[i] = (half_t)src[i];           // round to nearest even, like torch's .to(float16)
        else reinterpret_cast<float*>(dst)[i] = src[i];
    }
}

}  // namespace clip
}  // namespace nerfart

using namespace nerfart;
using namespace nerfart::clip;

extern "C" {

long long nerfart_clip_vitb32_blob_layout(long long* offsets) { return blob_layout(offsets); }

// The packer of the blob above.  tensors: HOST array of nerfart_clip_vitb32_n_tensors() DEVICE pointers to fp32 copies of the `visual.*`
// entries of a CLIP state dict, in the order nerfart_clip_vitb32_tensor_name(i) names them (name_out: a caller buffer; the return value is
// the tensor's element count, 0 for a bad index).  The blob (blob_bytes from nerfart_clip_vitb32_blob_layout) is written entirely: fp16
// matrices, fp32 vectors, zero padding.
int nerfart_clip_vitb32_n_tensors(void) { return N_PACK_TENSORS; }
long long nerfart_clip_vitb32_tensor_name(int i, char* name_out, int name_len) {
    if (i < 0 || i >= N_PACK_TENSORS) return 0;
    return pack_entry(i, name_out, name_len).n;
}
int nerfart_clip_vitb32_pack(const float* const* tensors, void* blob, long long blob_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (!tensors || !blob) { set_last_error("clip_vitb32_pack: NULL argument"); return 2; }
    if (check_blob_bytes(blob_bytes)) return 1;
    long long off[N_SECTIONS + 1];
    blob_layout(off);
    NERFART_HIP(hipMemsetAsync(blob, 0, (size_t)blob_bytes, st));
    for (int i = 0; i < N_PACK_TENSORS; ++i) {
        const PackEntry e = pack_entry(i, nullptr, 0);
        if (!tensors[i]) { set_last_error("clip_vitb32_pack: NULL tensor pointer"); return 2; }
        if (e.n * (e.half ? 2 : 4) != section_bytes(e.sec)) { set_last_error("clip_vitb32_pack: internal section table mismatch"); return 3; }
        const long long blocks = (e.n + 255) / 256;
        hipLaunchKernelGGL(k_pack_section, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, tensors[i], (char*)blob + off[e.sec], e.n, e.half);
    }
    NERFART_HIP(hipGetLastError());
    return 0;
}
long long nerfart_clip_vitb32_workspace_bytes(int B, int keep_for_bwd) { return B >= 1 ? work_layout(B, keep_for_bwd).total : 0; }

// C[M, N] fp32 = A[M, K] fp16 . W[N, K]^T fp16 (fp32 accumulate): the GEMM kernel on its own (tests).  M, N, K multiples of 64.
int nerfart_gemm_f16_nt(const void* A, const void* W, int M, int N, int K, float* C, void* stream) {
    if ((M | N | K) & 63) { set_last_error("nerfart_gemm_f16_nt: M, N, K must be multiples of 64"); return 1; }
    Epi e{};
    e.out_f32 = C; e.ldo = N; e.m_valid = M;
    return gemm<EPI_F32, A_F16>((hipStream_t)stream, A, K, (const half_t*)W, M, N, K, e);
}

// C[M, N] fp32 = A[M, K] fp16 . Wt[K, N] fp16: the same kernel reading the second operand with its reduction index as the ROW
// (what the backward GEMMs do with the forward weight matrices).
int nerfart_gemm_f16_nn(const void* A, const void* Wt, int M, int N, int K, float* C, void* stream) {
    if ((M | N | K) & 63) { set_last_error("nerfart_gemm_f16_nn: M, N, K must be multiples of 64"); return 1; }
    Epi e{};
    e.out_f32 = C; e.ldo = N; e.m_valid = M;
    return gemm<EPI_F32, A_F16, true>((hipStream_t)stream, A, K, (const half_t*)Wt, M, N, K, e);
}

int nerfart_clip_vitb32_image_fwd(const void* blob, long long blob_bytes, const float* img, int B, float* feat_out, int keep_for_bwd, void* workspace,
                                  long long workspace_bytes, void* stream) {
    if (check_args(blob, workspace, B, workspace_bytes, keep_for_bwd) || check_blob_bytes(blob_bytes)) return 1;
    hipStream_t st = (hipStream_t)stream;
    Blob bl;
    bl.base = (const char*)blob;
    blob_layout(bl.off);
    const Work w = work_layout(B, keep_for_bwd);
    char* ws = (char*)workspace;
    // padding rows are never stored by the kernels: zero once so they are finite operands
    NERFART_HIP(hipMemsetAsync(ws, 0, (size_t)(keep_for_bwd ? w.o_saved : w.total), st));
    half_t* patches = (half_t*)(ws + w.o_patches);
    float* pe = (float*)(ws + w.o_pe);
    float* xemb = (float*)(ws + w.o_xemb);
    float* xfin = (float*)(ws + w.o_xfin);
    half_t* y = (half_t*)(ws + w.o_y);
    half_t* att = (half_t*)(ws + w.o_att);
    half_t* act = (half_t*)(ws + w.o_act);
    half_t* x0 = (half_t*)(ws + w.o_x0);
    const int M = w.M, Mp = w.Mp, rows4 = (M + 3) / 4;

    hipLaunchKernelGGL(k_patchify, dim3(B * NP), dim3(256), 0, st, img, patches, B);
    { Epi e{}; e.out_f32 = pe; e.ldo = D; e.m_valid = B * NP;
      if (gemm<EPI_F32, A_F16>(st, patches, PK, bl.h(0), w.Pp, D, PK, e)) return 1; }
    auto layer_base = [&](int l) { return ws + w.o_saved + (keep_for_bwd ? (long long)l * w.layer_bytes : 0); };
    float* x_alt = (float*)(ws + w.o_saved + w.layer_bytes);     // !keep only: second residual buffer
    float* xin = (float*)(layer_base(0) + w.o_xin);
    hipLaunchKernelGGL(k_embed_lnpre, dim3(rows4), dim3(256), 0, st, pe, bl.f(100), bl.f(101), bl.f(102), bl.f(103), xemb, xin, M);
    for (int l = 0; l < NL; ++l) {
        char* lb = layer_base(l);
        float* xmid = (float*)(lb + w.o_xmid);
        half_t* qkv = (half_t*)(lb + w.o_qkv);
        half_t* pre = (half_t*)(lb + w.o_pre);
        float* xnext = (l + 1 == NL) ? xfin : (keep_for_bwd ? (float*)(layer_base(l + 1) + w.o_xin) : ((l & 1) ? (float*)(lb + w.o_xin) : x_alt));
        const int s = 2 + 8 * l, f = 104 + 8 * l;
        hipLaunchKernelGGL(k_ln_fwd, dim3(rows4), dim3(256), 0, st, xin, bl.f(f + 0), bl.f(f + 1), y, M);
        { Epi e{}; e.bias = bl.f(f + 2); e.out_f16 = qkv; e.ldo = 3 * D; e.m_valid = M;
          if (gemm<EPI_BIAS_F16, A_F16>(st, y, D, bl.h(s + 0), Mp, 3 * D, D, e)) return 1; }
        hipLaunchKernelGGL(k_attn_fwd, dim3(B * NH), dim3(256), 0, st, qkv, att);
        { Epi e{}; e.bias = bl.f(f + 3); e.resid = xin; e.out_f32 = xmid; e.ldo = D; e.m_valid = M;
          if (gemm<EPI_BIAS_RESID_F32, A_F16>(st, att, D, bl.h(s + 2), Mp, D, D, e)) return 1; }
        hipLaunchKernelGGL(k_ln_fwd, dim3(rows4), dim3(256), 0, st, xmid, bl.f(f + 4), bl.f(f + 5), y, M);
        { Epi e{}; e.bias = bl.f(f + 6); e.out_f16 = pre; e.out2_f16 = act; e.ldo = DM; e.m_valid = M;
          if (gemm<EPI_BIAS_GELU_F16, A_F16>(st, y, D, bl.h(s + 4), Mp, DM, D, e)) return 1; }
        { Epi e{}; e.bias = bl.f(f + 7); e.resid = xmid; e.out_f32 = xnext; e.ldo = D; e.m_valid = M;
          if (gemm<EPI_BIAS_RESID_F32, A_F16>(st, act, DM, bl.h(s + 6), Mp, D, DM, e)) return 1; }
        xin = xnext;
    }
    hipLaunchKernelGGL(k_lnpost, dim3((B + 3) / 4), dim3(256), 0, st, xfin, bl.f(200), bl.f(201), x0, B);
    { Epi e{}; e.out_f32 = feat_out; e.ldo = DOUT; e.m_valid = B;
      if (gemm<EPI_F32, A_F16, true>(st, x0, D, bl.h(99), w.Bp, DOUT, D, e)) return 1; }
    NERFART_HIP(hipGetLastError());
    return 0;
}

int nerfart_clip_vitb32_image_bwd(const void* blob, long long blob_bytes, int B, const float* g_feat, float* g_img, void* workspace, long long workspace_bytes,
                                  void* stream) {
    if (check_args(blob, workspace, B, workspace_bytes, 1) || check_blob_bytes(blob_bytes)) return 1;
    hipStream_t st = (hipStream_t)stream;
    Blob bl;
    bl.base = (const char*)blob;
    blob_layout(bl.off);
    const Work w = work_layout(B, 1);
    char* ws = (char*)workspace;
    float* scale = (float*)(ws + w.o_scale);
    float* dpatch = (float*)(ws + w.o_pe);
    float* xemb = (float*)(ws + w.o_xemb);
    float* xfin = (float*)(ws + w.o_xfin);
    half_t* y = (half_t*)(ws + w.o_y);
    half_t* att = (half_t*)(ws + w.o_att);
    half_t* act = (half_t*)(ws + w.o_act);
    float* t = (float*)(ws + w.o_t);
    float* dx = (float*)(ws + w.o_dx);
    half_t* dqkv = (half_t*)(ws + w.o_dqkv);
    half_t* g16 = (half_t*)(ws + w.o_g16);
    float* t0 = (float*)(ws + w.o_t0);
    const int M = w.M, Mp = w.Mp, rows4 = (M + 3) / 4;
    static bool attr_set = false;
    if (!attr_set) {
        NERFART_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_attn_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATTN_BWD_LDS));
        attr_set = true;
    }
    NERFART_HIP(hipMemsetAsync(g16, 0, (size_t)2 * w.Bp * DOUT, st));
    NERFART_HIP(hipMemsetAsync(dx, 0, (size_t)4 * Mp * D, st));
    NERFART_HIP(hipMemsetAsync(dqkv, 0, (size_t)2 * Mp * 3 * D, st));
    hipLaunchKernelGGL(k_gscale, dim3(1), dim3(1024), 0, st, g_feat, B * DOUT, scale, g16);
    { Epi e{}; e.out_f32 = t0; e.ldo = D; e.m_valid = B;
      if (gemm<EPI_F32, A_F16>(st, g16, DOUT, bl.h(99), w.Bp, D, DOUT, e)) return 1; }
    hipLaunchKernelGGL(k_lnpost_bwd, dim3((B + 3) / 4), dim3(256), 0, st, t0, xfin, bl.f(200), dx, B);
    for (int l = NL - 1; l >= 0; --l) {
        char* lb = ws + w.o_saved + (long long)l * w.layer_bytes;
        const float* xin = (const float*)(lb + w.o_xin);
        const float* xmid = (const float*)(lb + w.o_xmid);
        const half_t* qkv = (const half_t*)(lb + w.o_qkv);
        const half_t* pre = (const half_t*)(lb + w.o_pre);
        const int s = 2 + 8 * l, f = 104 + 8 * l;
        // MLP branch: d act = dx W2 -> x gelu'(pre) -> d ln_2 out = . W1 -> dx += LN'
        { Epi e{}; e.out_f16 = act; e.aux_f16 = pre; e.ldo = DM; e.m_valid = M;
          if (gemm<EPI_GELUBWD_F16, A_F32, true>(st, dx, D, bl.h(s + 6), Mp, DM, D, e)) return 1; }
        { Epi e{}; e.out_f32 = t; e.ldo = D; e.m_valid = M;
          if (gemm<EPI_F32, A_F16, true>(st, act, DM, bl.h(s + 4), Mp, D, DM, e)) return 1; }
        hipLaunchKernelGGL(k_ln_bwd, dim3(rows4), dim3(256), 0, st, t, xmid, bl.f(f + 4), dx, M);
        // attention branch: d attn = dx Wo -> attention backward -> d ln_1 out = dqkv Wqkv -> dx += LN'
        { Epi e{}; e.out_f16 = att; e.ldo = D; e.m_valid = M;
          if (gemm<EPI_F16, A_F32, true>(st, dx, D, bl.h(s + 2), Mp, D, D, e)) return 1; }
        hipLaunchKernelGGL(k_attn_bwd, dim3(B * NH), dim3(256), ATTN_BWD_LDS, st, qkv, att, dqkv);
        { Epi e{}; e.out_f32 = t; e.ldo = D; e.m_valid = M;
          if (gemm<EPI_F32, A_F16, true>(st, dqkv, 3 * D, bl.h(s + 0), Mp, D, 3 * D, e)) return 1; }
        hipLaunchKernelGGL(k_ln_bwd, dim3(rows4), dim3(256), 0, st, t, xin, bl.f(f + 0), dx, M);
    }
    NERFART_HIP(hipMemsetAsync(y, 0, (size_t)2 * Mp * D, st));
    hipLaunchKernelGGL(k_lnpre_bwd, dim3(rows4), dim3(256), 0, st, dx, xemb, bl.f(102), y, M);
    { Epi e{}; e.out_f32 = dpatch; e.ldo = PK; e.m_valid = B * NP;
      if (gemm<EPI_F32, A_F16, true>(st, y, D, bl.h(0), w.Pp, PK, D, e)) return 1; }
    hipLaunchKernelGGL(k_unpatchify, dim3(B * NP), dim3(256), 0, st, dpatch, scale, g_img, B);
    NERFART_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"

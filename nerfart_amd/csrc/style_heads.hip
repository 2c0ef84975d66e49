// style_heads.hip - the image side of the style losses around the CLIP encoder (rows a20-a22):
//
//   k_resample_fwd / _bwd   what torchvision's Resize / CenterCrop / crops / ZeroPad2d / Normalize chains of the reference's
//                           `preprocess` pipelines do (criteria/clip_loss.py:166-168, contrastive_loss.py:98-101,
//                           patchnce_loss.py:98-117, :184-215), as ONE gather per stage: every output pixel of every output
//                           image is a bicubic (A = -0.75, align_corners = False, no antialias - torchvision 0.9's tensor
//                           Resize) or bilinear interpolation of a (virtually zero-padded) source at the position its crop
//                           window selects, followed by a per-channel affine ((x + 1) / 2 and the CLIP mean / std folded).
//                           Backward scatters the cotangent with fp32 atomics.
//   k_clip_style_heads      directional CLIP loss (clip_loss.py:244-254), global contrastive loss (contrastive_loss.py:146-153)
//                           and the PatchNCE terms (patchnce_loss.py:153-173) from the 4 + P image features and the cached
//                           text features, value AND gradient w.r.t. the image features in one launch (fp32), one workgroup
//                           per head / per crop.
#include "nerfart_common.h"
#include <cmath>

namespace nerfart {
namespace style {

// ---- interpolation taps (ATen UpSample.h conventions) ------------------------------------------------------------
struct Taps { int idx[4]; float w[4]; int n; };

__device__ __forceinline__ float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
__device__ __forceinline__ float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }

// destination index d of a size-`out` axis resampled from a size-`in` axis
__device__ __forceinline__ Taps taps_of(int d, int in, int out, int bicubic) {
    Taps t;
    const float scale = (float)in / (float)out;
    float real = scale * ((float)d + 0.5f) - 0.5f;
    if (bicubic) {
        const float fl = floorf(real);
        const int i = (int)fl;
        const float x = real - fl;
        const float A = -0.75f;
        t.w[0] = cubic2(x + 1.f, A); t.w[1] = cubic1(x, A); t.w[2] = cubic1(1.f - x, A); t.w[3] = cubic2(2.f - x, A);
#pragma unroll
        for (int k = 0; k < 4; ++k) t.idx[k] = min(max(i - 1 + k, 0), in - 1);
        t.n = 4;
    } else {
        real = fmaxf(real, 0.f);
        const int i0 = min((int)real, in - 1);
        const int i1 = i0 + (i0 < in - 1 ? 1 : 0);
        const float l1 = fminf(fmaxf(real - (float)i0, 0.f), 1.f);
        t.idx[0] = i0; t.idx[1] = i1; t.idx[2] = i0; t.idx[3] = i0;
        t.w[0] = 1.f - l1; t.w[1] = l1; t.w[2] = 0.f; t.w[3] = 0.f;
        t.n = 2;
    }
    return t;
}

struct Resample {
    int n_src;            // 1: every output image reads source image 0; otherwise output n reads source n
    int C, Hs, Ws;        // source image size
    int pad_t, pad_l;     // the source sits at (pad_t, pad_l) inside a zero canvas of Hp x Wp (ZeroPad2d)
    int Hp, Wp;
    int Hr, Wr;           // the canvas is resampled to Hr x Wr ...
    int bicubic;
    int N, Ho, Wo;        // ... and output image n is the Ho x Wo window of it at crop[n] = (y0, x0) ((0, 0) if crop == NULL)
};
// win != NULL: output n reads the sub-window win[n] = (y, x, h, w) of its source image instead of all of it (crop-then-resize:
// taps clamp at the window's edges, as interpolating the cropped tensor does); the canvas is then the window itself (no padding).
struct View { size_t off; int Hs, Ws, Hp, Wp, pad_t, pad_l; };
__device__ __forceinline__ View view_of(const Resample& p, const int* __restrict__ win, int n, int c) {
    View v;
    const size_t plane = ((size_t)(p.n_src == 1 ? 0 : n) * p.C + c) * p.Hs * p.Ws;
    if (win) {
        const int y = win[4 * n], x = win[4 * n + 1];
        v.Hs = win[4 * n + 2]; v.Ws = win[4 * n + 3];
        v.Hp = v.Hs; v.Wp = v.Ws; v.pad_t = 0; v.pad_l = 0;
        v.off = plane + (size_t)y * p.Ws + x;
    } else {
        v.Hs = p.Hs; v.Ws = p.Ws; v.Hp = p.Hp; v.Wp = p.Wp; v.pad_t = p.pad_t; v.pad_l = p.pad_l;
        v.off = plane;
    }
    return v;
}

__global__ __launch_bounds__(256) void k_resample_fwd(Resample p, const float* __restrict__ src, const int* __restrict__ crop,
                                                      const int* __restrict__ win, const float* __restrict__ affine, float* __restrict__ dst) {
    const long long total = (long long)p.N * p.C * p.Ho * p.Wo;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % p.Wo), oy = (int)((i / p.Wo) % p.Ho), c = (int)((i / ((long long)p.Wo * p.Ho)) % p.C);
        const int n = (int)(i / ((long long)p.Wo * p.Ho * p.C));
        const int y0 = crop ? crop[2 * n] : 0, x0 = crop ? crop[2 * n + 1] : 0;
        const View vw = view_of(p, win, n, c);
        const Taps ty = taps_of(oy + y0, vw.Hp, p.Hr, p.bicubic), tx = taps_of(ox + x0, vw.Wp, p.Wr, p.bicubic);
        const float* s = src + vw.off;
        float v = 0.f;
        for (int a = 0; a < ty.n; ++a) {
            const int sy = ty.idx[a] - vw.pad_t;
            if (sy < 0 || sy >= vw.Hs) continue;
            float row = 0.f;
            for (int b = 0; b < tx.n; ++b) {
                const int sx = tx.idx[b] - vw.pad_l;
                if (sx >= 0 && sx < vw.Ws) row += tx.w[b] * s[(size_t)sy * p.Ws + sx];
            }
            v += ty.w[a] * row;
        }
        if (affine) v = v * affine[c] + affine[p.C + c];
        dst[i] = v;
    }
}

// g_src += J^T g_dst (g_src zeroed by the caller; several stages may accumulate into it)
__global__ __launch_bounds__(256) void k_resample_bwd(Resample p, const float* __restrict__ g_dst, const int* __restrict__ crop,
                                                      const int* __restrict__ win, const float* __restrict__ affine, float* __restrict__ g_src) {
    const long long total = (long long)p.N * p.C * p.Ho * p.Wo;
    for (long long i = blockIdx.x * 256LL + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ox = (int)(i % p.Wo), oy = (int)((i / p.Wo) % p.Ho), c = (int)((i / ((long long)p.Wo * p.Ho)) % p.C);
        const int n = (int)(i / ((long long)p.Wo * p.Ho * p.C));
        float g = g_dst[i];
        if (affine) g *= affine[c];
        if (g == 0.f) continue;
        const int y0 = crop ? crop[2 * n] : 0, x0 = crop ? crop[2 * n + 1] : 0;
        const View vw = view_of(p, win, n, c);
        const Taps ty = taps_of(oy + y0, vw.Hp, p.Hr, p.bicubic), tx = taps_of(ox + x0, vw.Wp, p.Wr, p.bicubic);
        float* s = g_src + vw.off;
        for (int a = 0; a < ty.n; ++a) {
            const int sy = ty.idx[a] - vw.pad_t;
            if (sy < 0 || sy >= vw.Hs) continue;
            for (int b = 0; b < tx.n; ++b) {
                const int sx = tx.idx[b] - vw.pad_l;
                if (sx >= 0 && sx < vw.Ws) atomicAdd(&s[(size_t)sy * p.Ws + sx], g * ty.w[a] * tx.w[b]);
            }
        }
    }
}

// ---- loss heads --------------------------------------------------------------------------------------------------
constexpr int FD = 512;          // CLIP embedding width
constexpr int HT = 512;          // threads of the single workgroup: thread j owns feature component j
constexpr int MAXP = 16, MAXS = 16;

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < HT / 64; ++i) s += red[i];
    return s;
}

// n <= NMAX values per thread summed over the workgroup at once (one barrier pair for all of them); v[k] <- total.  The loops run
// to the compile-time NMAX with a predicate so that v[] stays in registers.
template <int NMAX>
__device__ __forceinline__ void block_sum_vec(float (&v)[NMAX], int n, float* redv) {
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
        float x = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
        v[k] = x;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < NMAX; ++k) if (k < n) redv[k * (HT / 64) + (threadIdx.x >> 6)] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NMAX; ++k) {
        float sum = 0.f;
        if (k < n) {
#pragma unroll
            for (int i = 0; i < HT / 64; ++i) sum += redv[k * (HT / 64) + i];
        }
        v[k] = sum;
    }
}

// feats [4 + P, 512]: 0 directional prediction, 1 directional source (no gradient), 2 contrastive prediction, 3 contrastive
// source (no gradient), 4.. the P PatchNCE crops.  text_dir [512] (unit), t_tgt / t_con [T, 512] (unit rows: templates of the
// target prompt / of the contrastive head's negative prompt), t_neg [S, T, 512] (the PatchNCE negative prompts).
// out[0..3] += total, directional, contrastive, patchnce (unweighted parts; zeroed by the caller); g_feats [4 + P, 512] =
// d total / d feats.  One workgroup per head / per PatchNCE crop: block 0 directional, 1 contrastive, 2 + p crop p.
__global__ __launch_bounds__(HT) void k_clip_style_heads(const float* __restrict__ feats, int P, const float* __restrict__ text_dir,
                                                        const float* __restrict__ t_tgt, const float* __restrict__ t_con,
                                                        const float* __restrict__ t_neg, int S, int T, float w_dir, float w_con,
                                                        float w_nce, float margin, float tau, float* __restrict__ out,
                                                        float* __restrict__ g_feats) {
    __shared__ float red[HT / 64];
    __shared__ float redv[2 * (1 + MAXS) * (HT / 64)];
    const int j = threadIdx.x;
    const float eps_cos = 1e-8f, eps_pd = 1e-6f;
    auto unit = [&](int row, float& fh, float& nrm) {          // f / |f| (criteria: f / f.norm(dim=-1, keepdim=True))
        const float f = feats[(size_t)row * FD + j];
        nrm = sqrtf(block_sum(f * f, red));
        fh = f / nrm;
    };
    // d L / d f from d L / d fhat:  (g - fhat (fhat . g)) / |f|
    auto unit_bwd = [&](float g, float fh, float nrm) { return (g - fh * block_sum(g * fh, red)) / nrm; };

    if (blockIdx.x == 0) {
        // ---- directional: 1 - cos(normalize(fhat0 - fhat1), dir)
        float f0, n0, f1, n1;
        unit(0, f0, n0);
        unit(1, f1, n1);
        const float e = f0 - f1;
        const float en = sqrtf(block_sum(e * e, red));
        const float eh = e / en;
        const float d = text_dir[j];
        const float dn = sqrtf(block_sum(d * d, red));
        const float ehn = sqrtf(block_sum(eh * eh, red));
        const float dot = block_sum(eh * d, red);
        const float den = fmaxf(ehn, eps_cos) * fmaxf(dn, eps_cos);
        const float cosv = dot / den;
        const float g_eh = -(d / den - cosv * eh / fmaxf(ehn * ehn, eps_cos * eps_cos));      // d (1 - cos) / d eh
        const float g_e = (g_eh - eh * block_sum(g_eh * eh, red)) / en;
        g_feats[0 * FD + j] = w_dir * unit_bwd(g_e, f0, n0);
        g_feats[1 * FD + j] = 0.f;
        if (j == 0) { atomicAdd(&out[0], w_dir * (1.f - cosv)); atomicAdd(&out[1], 1.f - cosv); }
    } else if (blockIdx.x == 1) {
        // ---- global contrastive: mean_t(near_t^2 + relu(m - far_text_t)^2 + relu(m - far_img)^2), pairwise_distance(x, y) = |x - y + 1e-6|
        float f2, n2, f3, n3;
        unit(2, f2, n2);
        unit(3, f3, n3);
        float L_con = 0.f, g2 = 0.f;
        const float di = f2 - f3 + eps_pd;
        const float far_img = sqrtf(block_sum(di * di, red));
        const float hi = fmaxf(margin - far_img, 0.f);
        L_con += hi * hi;                                                  // the same value for every template
        if (hi > 0.f) g2 += -2.f * hi * di / far_img;
        float acc_l = 0.f;
        for (int t = 0; t < T; ++t) {
            float v[2];
            const float a = f2 - t_tgt[(size_t)t * FD + j] + eps_pd, b = f2 - t_con[(size_t)t * FD + j] + eps_pd;
            v[0] = a * a; v[1] = b * b;
            block_sum_vec(v, 2, redv);
            const float far_t = sqrtf(v[1]);
            const float ht = fmaxf(margin - far_t, 0.f);
            acc_l += v[0] + ht * ht;
            g2 += (2.f * a + (ht > 0.f ? -2.f * ht * b / far_t : 0.f)) / (float)T;
        }
        L_con += acc_l / (float)T;
        g_feats[2 * FD + j] = w_con * unit_bwd(g2, f2, n2);
        g_feats[3 * FD + j] = 0.f;
        if (j == 0) { atomicAdd(&out[0], w_con * L_con); atomicAdd(&out[2], L_con); }
    } else {
        // ---- PatchNCE crop p: mean_t( -log( pos / (pos + sum_s neg_s) ) ), pos = exp(cos(f, T_tgt[t]) / tau)
        const int p = blockIdx.x - 2;
        float fp, npn;
        unit(4 + p, fp, npn);
        const float fn = fmaxf(sqrtf(block_sum(fp * fp, red)), eps_cos);       // |fhat| (= 1 up to rounding), as cosine_similarity divides
        float gp = 0.f, L = 0.f;
        for (int t = 0; t < T; ++t) {
            float tv[1 + MAXS], v[2 * (1 + MAXS)];
            tv[0] = t_tgt[(size_t)t * FD + j];
#pragma unroll
            for (int s = 0; s < MAXS; ++s) tv[1 + s] = (s < S) ? t_neg[((size_t)s * T + t) * FD + j] : 0.f;
#pragma unroll
            for (int k = 0; k <= MAXS; ++k) { v[2 * k] = tv[k] * tv[k]; v[2 * k + 1] = fp * tv[k]; }
            block_sum_vec(v, 2 * (1 + S), redv);                                  // norms and dot products of the 1 + S text rows at once
            float c[1 + MAXS], tn[1 + MAXS];
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k <= MAXS; ++k)
                if (k <= S) { tn[k] = fmaxf(sqrtf(v[2 * k]), eps_cos); c[k] = v[2 * k + 1] / (fn * tn[k]); mx = fmaxf(mx, c[k] / tau); }
            float z = 0.f;
#pragma unroll
            for (int k = 0; k <= MAXS; ++k) if (k <= S) z += __expf(c[k] / tau - mx);
            const float lse = mx + __logf(z);
            L += (lse - c[0] / tau) / (float)T;
            // d/d fhat of (lse - cos_0 / tau) = sum_k (softmax_k - [k == 0]) dcos_k / tau;  dcos_k = t_k / (fn tn_k) - cos_k fhat / fn^2
            float gt = 0.f;
#pragma unroll
            for (int k = 0; k <= MAXS; ++k)
                if (k <= S) {
                    const float sk = __expf(c[k] / tau - lse) - (k == 0 ? 1.f : 0.f);
                    gt += sk * (tv[k] / (fn * tn[k]) - c[k] * fp / (fn * fn));
                }
            gp += gt / (tau * (float)T);
        }
        g_feats[(size_t)(4 + p) * FD + j] = w_nce * unit_bwd(gp, fp, npn);
        if (j == 0) { atomicAdd(&out[0], w_nce * L); atomicAdd(&out[3], L); }
    }
}

}  // namespace style
}  // namespace nerfart

using namespace nerfart;
using namespace nerfart::style;

extern "C" {

static int make_resample(Resample& p, int n_src, int C, int Hs, int Ws, int pad_t, int pad_l, int Hp, int Wp, int Hr, int Wr, int mode,
                         int N, int Ho, int Wo) {
    if (n_src != 1 && n_src != N) { set_last_error("resample: n_src must be 1 or N"); return 1; }
    if (C < 1 || Hs < 1 || Ws < 1 || Hr < 1 || Wr < 1 || N < 1 || Ho < 1 || Wo < 1 || pad_t < 0 || pad_l < 0 || Hp < pad_t + Hs || Wp < pad_l + Ws ||
        (mode != 0 && mode != 1)) {
        set_last_error("resample: bad geometry / mode (0 = bilinear, 1 = bicubic)");
        return 1;
    }
    p = Resample{n_src, C, Hs, Ws, pad_t, pad_l, Hp, Wp, Hr, Wr, mode, N, Ho, Wo};
    return 0;
}
static unsigned resample_grid(const Resample& p) {
    const long long total = (long long)p.N * p.C * p.Ho * p.Wo;
    return (unsigned)((total + 255) / 256 < 65535 * 16 ? (total + 255) / 256 : 65535 * 16);
}

int nerfart_resample_fwd(const float* src, int n_src, int C, int Hs, int Ws, int pad_t, int pad_l, int Hp, int Wp, int Hr, int Wr, int mode,
                         const int* crop_yx, const int* src_win, const float* affine, float* dst, int N, int Ho, int Wo, void* stream) {
    Resample p;
    if (make_resample(p, n_src, C, Hs, Ws, pad_t, pad_l, Hp, Wp, Hr, Wr, mode, N, Ho, Wo)) return 1;
    hipLaunchKernelGGL(k_resample_fwd, dim3(resample_grid(p)), dim3(256), 0, (hipStream_t)stream, p, src, crop_yx, src_win, affine, dst);
    NERFART_HIP(hipGetLastError());
    return 0;
}
int nerfart_resample_bwd(const float* g_dst, int n_src, int C, int Hs, int Ws, int pad_t, int pad_l, int Hp, int Wp, int Hr, int Wr, int mode,
                         const int* crop_yx, const int* src_win, const float* affine, float* g_src, int N, int Ho, int Wo, void* stream) {
    Resample p;
    if (make_resample(p, n_src, C, Hs, Ws, pad_t, pad_l, Hp, Wp, Hr, Wr, mode, N, Ho, Wo)) return 1;
    hipLaunchKernelGGL(k_resample_bwd, dim3(resample_grid(p)), dim3(256), 0, (hipStream_t)stream, p, g_dst, crop_yx, src_win, affine, g_src);
    NERFART_HIP(hipGetLastError());
    return 0;
}
int nerfart_clip_style_heads(const float* feats, int n_patches, const float* text_dir, const float* t_tgt, const float* t_con, const float* t_neg,
                             int n_neg, int n_templates, float w_dir, float w_con, float w_nce, float margin, float tau, float* out4,
                             float* g_feats, void* stream) {
    if (n_patches < 0 || n_patches > MAXP || n_neg < 0 || n_neg > MAXS || n_templates < 1) {
        set_last_error("clip_style_heads: need n_patches <= 16, n_neg <= 16, n_templates >= 1");
        return 1;
    }
    NERFART_HIP(hipMemsetAsync(out4, 0, 4 * sizeof(float), (hipStream_t)stream));
    hipLaunchKernelGGL(k_clip_style_heads, dim3(2 + n_patches), dim3(HT), 0, (hipStream_t)stream, feats, n_patches, text_dir, t_tgt, t_con, t_neg, n_neg,
                       n_templates, w_dir, w_con, w_nce, margin, tau, out4, g_feats);
    NERFART_HIP(hipGetLastError());
    return 0;
}

}  // extern "C"
